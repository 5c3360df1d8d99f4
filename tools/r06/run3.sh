#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x -s -k "upcat_conv8 or bias_gradient_in_front or pack_conv or repack or bias_gradients_come" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; tail -15 $O/t1.txt
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_decisions.py -m gpu -q > $O/t2.txt 2>&1
echo "rc=$?" >> $O/t2.txt; tail -5 $O/t2.txt
for fr in 2 8; do
  timeout 900 python tools/train_switch_ab.py FaFNet $fr "TRAIN_BN_BIAS_ZERO=0,TRAIN_UPCAT_CONV=0" "TRAIN_BN_BIAS_ZERO=1,TRAIN_UPCAT_CONV=0" "TRAIN_BN_BIAS_ZERO=0,TRAIN_UPCAT_CONV=1" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
timeout 900 python tools/train_switch_ab.py V2VNet 2 "TRAIN_BN_BIAS_ZERO=0,TRAIN_UPCAT_CONV=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
