#!/bin/bash
# round-6 GPU call 2: the parity tests of call 1 again (bounds fixed), the new training kernels' tests, and the paired A/B of the two training switches
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_direct_oracle.py tests/test_gpu_decisions.py tests/test_gpu_bits_input.py "tests/test_gpu_stages.py::test_voxelize_ring_sweep_bit_exact" -m gpu -q -s > $O/new_tests.txt 2>&1
echo "new tests rc=$?" | tee -a $O/new_tests.txt
timeout 1500 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x > $O/train_kernel_tests.txt 2>&1
echo "train kernel tests rc=$?" | tee -a $O/train_kernel_tests.txt
tail -3 $O/train_kernel_tests.txt
for fr in 2 8; do
  timeout 900 python tools/train_switch_ab.py FaFNet $fr "CONV1X1=0,TRAIN_HEAD_PACK=0" "CONV1X1=1,TRAIN_HEAD_PACK=0" "CONV1X1=0,TRAIN_HEAD_PACK=1" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
timeout 900 python tools/train_switch_ab.py V2VNet 2 "CONV1X1=0,TRAIN_HEAD_PACK=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
grep -E "tail vs oracle|pair_bits vs oracle|seed [0-9]:|passed|failed" $O/new_tests.txt | head -40
