#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run34; mkdir -p $O
timeout 600 python tools/train_host_profile.py faf 2>&1 | grep -v amdgpu.ids > $O/host_profile.txt
head -12 $O/host_profile.txt | cut -c1-170
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_c_abi.py tests/test_gpu_tail.py -m gpu -q -x 2>&1 | tail -3
