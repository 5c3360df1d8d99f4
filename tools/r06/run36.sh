#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run36; mkdir -p $O
for f in "FaFNet 2" "FaFNet 8" "V2VNet 2"; do
  timeout 900 python tools/train_switch_ab.py $f "BN_PARTIAL_T=0" default 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
done
cat $O/train_ab.txt
