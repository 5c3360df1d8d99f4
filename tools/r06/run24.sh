#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run24; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_kernels.py -m gpu -q -x -s -k "v2v_message or gates_nhwc or nhwc_fusion or warp_on_kernels" > $O/t1.txt 2>&1
echo "rc=$?" >> $O/t1.txt; grep -v "amdgpu.ids" $O/t1.txt | tail -25
timeout 900 python tools/train_switch_ab.py V2VNet 2 "TRAIN_V2V_NHWC=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
timeout 900 python tools/train_switch_ab.py V2VNet 8 "TRAIN_V2V_NHWC=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
timeout 600 python tools/train_op_census.py v2v 2 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/census_v2v10.txt
head -24 $O/census_v2v10.txt | cut -c1-170
