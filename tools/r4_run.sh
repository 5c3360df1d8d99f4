#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
for b in 1 8 64; do timeout 300 python tools/layer_profile.py $b 2>/dev/null | grep -E "total kernel|warp_fuse"; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_latency_tweaks.txt
cat gpurun_out/r04_latency_tweaks.txt
