#!/bin/bash
# Round 3: fused detection heads -- tests, then the configs table (points -> detections both ways) and the per-layer profile at 1 / 8 frames (latency work)
cd "$(dirname "$0")/.."
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_postprocess.py -m gpu -q -x 2>&1 | tail -5 | tee $O/tests.txt
timeout 900 python tools/bench_configs.py --frames 64 > $O/configs.txt 2>&1; tail -12 $O/configs.txt | cut -c1-400
for f in 1 8; do timeout 600 python tools/layer_profile.py $f > $O/layer_profile_b$f.txt 2>&1; done
head -50 $O/layer_profile_b1.txt
