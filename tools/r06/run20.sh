#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run20; mkdir -p $O
timeout 600 python tools/train_op_census.py faf 2 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/census_faf10.txt
head -12 $O/census_faf10.txt | cut -c1-200
for fr in 2 4 8; do
  timeout 900 python tools/train_switch_ab.py FaFNet $fr "TRAIN_SPLITK=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
timeout 900 python tools/train_switch_ab.py V2VNet 2 "TRAIN_SPLITK=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
timeout 2400 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -m gpu -q -x > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; tail -5 $O/t1.txt
