#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run29; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -m gpu -q -x -k "adam or Adam or optimizer or graphed or fused" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; grep -v "amdgpu.ids" $O/t1.txt | tail -6
