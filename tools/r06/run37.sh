#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run37; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -m gpu -q -x > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; tail -3 $O/t1.txt
