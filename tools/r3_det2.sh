#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_map.py -m gpu -q -x 2>&1 | tail -5 | tee $O/tests.txt
python3 tools/det_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/det_profile.txt
timeout 900 python tools/bench_configs.py --frames 64 2>&1 | grep -v amdgpu.ids > $O/configs.txt; grep "^2" $O/configs.txt; tail -1 $O/configs.txt | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['points_to_detections'])"
