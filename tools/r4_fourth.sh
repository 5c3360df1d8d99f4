#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/w1_probe.hip -o /tmp/w1_probe && /tmp/w1_probe > gpurun_out/r04_w1_probe_b.txt 2>&1
cat gpurun_out/r04_w1_probe_b.txt
timeout 1200 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_dataset.py tests/test_gpu_bits_input.py tests/test_gpu_postprocess.py tests/test_gpu_train_kernels.py tests/test_gpu_prefetch.py tests/test_gpu_stream.py -m gpu -q > gpurun_out/r04_gpu_suite_c.txt 2>&1
tail -8 gpurun_out/r04_gpu_suite_c.txt
