#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run30; mkdir -p $O
timeout 600 python tools/train_op_census.py faf 2 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/census_faf10.txt
timeout 600 python tools/train_op_census.py v2v 2 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/census_v2v10.txt
head -3 $O/census_faf10.txt | cut -c1-200; head -3 $O/census_v2v10.txt | cut -c1-200
bash tools/train_step_profile.sh > $O/train_step_profile_stdout.txt 2>&1
cp gpurun_out/train_prof/*_kernels.txt gpurun_out/train_prof/*_traffic.txt gpurun_out/train_prof/*_traffic.json $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 700 $O/bench.json
timeout 2400 python -m pytest tests -m gpu -q -x > $O/suite.txt 2>&1
echo "rc=$?" >> $O/suite.txt; tail -4 $O/suite.txt
