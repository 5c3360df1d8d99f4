#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
{ timeout 600 python tools/train_switch_ab.py FaFNet 2; timeout 600 python tools/train_switch_ab.py V2VNet 2; timeout 600 python tools/train_switch_ab.py FaFNet 8; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_train_switch_ab.txt
cat gpurun_out/r04_train_switch_ab.txt
