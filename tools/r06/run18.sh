#!/bin/bash
# (historical: the BN_FOLD switch this script timed was removed again -- profiles/r06_bn_fold_rejected.txt)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x -s -k "bn_ or training_step or hip_graph" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; grep -E "differing between|passed|failed|rc=" $O/t1.txt | tail -12
for fr in 2 4 8; do
  timeout 900 python tools/train_switch_ab.py FaFNet $fr "BN_FOLD=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
timeout 900 python tools/train_switch_ab.py V2VNet 2 "BN_FOLD=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
