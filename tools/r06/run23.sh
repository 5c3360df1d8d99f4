#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run23; mkdir -p $O
timeout 900 python tools/train_switch_ab.py FaFNet 2 "TRAIN_SPLITK=0" default "TRAIN_SPLITK=480" "TRAIN_SPLITK=640" "TRAIN_SPLITK=800" 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
timeout 900 python tools/train_switch_ab.py FaFNet 4 "TRAIN_SPLITK=0" default "TRAIN_SPLITK=480" "TRAIN_SPLITK=640" "TRAIN_SPLITK=800" 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
timeout 900 python tools/train_switch_ab.py V2VNet 2 "TRAIN_SPLITK=0" default "TRAIN_SPLITK=480" "TRAIN_SPLITK=640" "TRAIN_SPLITK=800" 2>&1 | grep -v amdgpu.ids >> $O/train_ab.txt
cat $O/train_ab.txt
