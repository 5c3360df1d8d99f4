#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run35; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x -k "bn_ or batchnorm or training_step or bias_grad" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; tail -3 $O/t1.txt
for f in 2 8; do
V2X_CENSUS_SHOW=bn_finish timeout 600 python tools/train_op_census.py faf $f 2>&1 | grep "^# " | grep bn_finish | awk '{n[$2]++; t[$2]+=$(NF-1)} END {for (k in n) printf "%s x%d  %.1f us total  %.2f us each\n", k, n[k], t[k], t[k]/n[k]}'
V2X_CENSUS_SHOW=bn_partial timeout 600 python tools/train_op_census.py faf $f 2>&1 | grep "^# " | grep bn_partial | awk '{n[$3]++; t[$3]+=$(NF-1)} END {for (k in n) printf "%s x%d  %.1f us total  %.2f us each\n", k, n[k], t[k], t[k]/n[k]}'
done
timeout 900 python tools/train_switch_ab.py FaFNet 2 default 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python tools/train_switch_ab.py FaFNet 8 default 2>&1 | grep -v amdgpu.ids | tail -2
