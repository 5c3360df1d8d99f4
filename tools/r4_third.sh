#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/w1_probe.hip -o /tmp/w1_probe && /tmp/w1_probe > gpurun_out/r04_w1_probe.txt 2>&1
cat gpurun_out/r04_w1_probe.txt
timeout 1800 python3 -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite_b.txt 2>&1
tail -25 gpurun_out/r04_gpu_suite_b.txt
python3 - <<'PY' > gpurun_out/r04_latency_a.txt 2>&1
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'v2x-sim_amd')
import torch, bench
from v2x_sim_amd.configs import Config
from v2x_sim_amd.models.det import V2VNet
from v2x_sim_amd.utils.synthetic import init_synthetic_weights
dev = torch.device('cuda:0')
model = init_synthetic_weights(V2VNet(Config('test')), seed=0).to(dev)
print(json.dumps(bench.measure_latency(model, dev)))
print(json.dumps(bench.measure_latency(model, dev, small_batch=False)))
PY
cat gpurun_out/r04_latency_a.txt | tail -3
