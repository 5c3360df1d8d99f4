#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run19; mkdir -p $O
timeout 600 python tools/train_op_census.py faf 2 2>&1 | grep -v amdgpu.ids > $O/census_faf10.txt
timeout 600 python tools/train_op_census.py v2v 2 2>&1 | grep -v amdgpu.ids > $O/census_v2v10.txt
head -60 $O/census_faf10.txt | cut -c1-250
