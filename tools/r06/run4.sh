#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_decisions.py tests/test_gpu_train.py -m gpu -q -s > $O/t2.txt 2>&1
echo "rc=$?" >> $O/t2.txt; grep -E "passed|failed|^FAILED|residue|seed [0-9]" $O/t2.txt | tail -20
