#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -4
python - <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "."); sys.path.insert(0, "v2x-sim_amd")
import torch
import bench_configs as bc
r = bc.run_training(torch.device("cuda:0"))
print({k: {kk[:22]: round(vv, 3) for kk, vv in v.items() if "MIOpen" not in kk} for k, v in r.items() if isinstance(v, dict)}, flush=True)
PY
timeout 300 python tools/train_small_ops.py FaFNet 2>&1 | grep -v "amdgpu.ids\|Warn\|warn"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_train_tests.txt
cat gpurun_out/r04_train_tests.txt
