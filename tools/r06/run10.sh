#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run10; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x -k "wgrad or bias_gradient_in_front" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; tail -4 $O/t1.txt
for fr in 2 8; do
  timeout 900 python tools/train_switch_ab.py FaFNet $fr "WGRAD_REDUCE4=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
timeout 900 python tools/train_switch_ab.py V2VNet 2 "WGRAD_REDUCE4=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
