#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_stages.py -m gpu -q -x -k "small_batch or voxel" 2>&1 | tail -8 | tee $O/tests.txt
for sb in 1; do for f in 1 8; do V2X_SMALL_BATCH=$sb timeout 600 python tools/layer_profile.py $f 2>&1 | grep -v amdgpu.ids > $O/layer_profile_b${f}_sb$sb.txt; head -1 $O/layer_profile_b${f}_sb$sb.txt; done; done
cat $O/layer_profile_b1_sb1.txt
for v in 1 0; do V2X_VOXELIZE_LDS=$v timeout 300 python tools/layer_profile.py 1 2>&1 | grep voxelize; V2X_VOXELIZE_LDS=$v timeout 300 python tools/layer_profile.py 8 2>&1 | grep voxelize; done
python3 - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "v2x-sim_amd")]
import torch, bench
from v2x_sim_amd.configs import Config
from v2x_sim_amd.models.det import V2VNet
from v2x_sim_amd.utils.synthetic import init_synthetic_weights
dev = torch.device("cuda:0")
m = init_synthetic_weights(V2VNet(Config("test"), num_agent=5), seed=0).to(dev)
print("latency (SMALL_BATCH=1):", {k: round(v, 3) for k, v in bench.measure_latency(m, dev).items() if k != "mode"})
print("latency (default)      :", {k: round(v, 3) for k, v in bench.measure_latency(m, dev, small_batch=False).items() if k != "mode"})
PY
