#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run33; mkdir -p $O
timeout 600 python tools/train_host_profile.py faf 2>&1 | grep -v amdgpu.ids > $O/host_profile.txt
head -60 $O/host_profile.txt | cut -c1-170
