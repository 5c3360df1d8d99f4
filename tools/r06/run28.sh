#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run28; mkdir -p $O
timeout 900 python tools/infer_op_census.py 64 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/infer_census.txt
cut -c1-200 $O/infer_census.txt | head -60
