#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 bash tools/profile_round.sh r04h > gpurun_out/r04h_profile.log 2>&1
for c in v2vnet seg upperbound; do timeout 600 bash tools/profile_round.sh --config $c r04h > gpurun_out/r04h_$c.log 2>&1; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_gpu_suite_e.txt
tail -3 gpurun_out/r04_gpu_suite_e.txt; cut -c1-300 gpurun_out/prof_r04h/bench.json | tail -1
