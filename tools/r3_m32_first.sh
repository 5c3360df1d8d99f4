#!/bin/bash
# Round 3, first GPU pass of the 32x32x16 form of stream8g: correctness, then the paired A/B against the 16x16x32 form (same library, the dispatch switch).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3a/stream_tests.txt
cat gpurun_out/r3a/stream_tests.txt
SO=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
timeout 900 python3 tools/ab_inproc.py $SO $SO A:V2X_STREAM_M32=0 B:V2X_STREAM_M32=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3a/ab_m32.txt | head -8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o /tmp/mfma_probe && timeout 300 /tmp/mfma_probe > gpurun_out/r3a/mfma_probe.txt 2>&1
grep "32x32x16\|block acc\[4\]" gpurun_out/r3a/mfma_probe.txt
