#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run27; mkdir -p $O
timeout 1200 python tools/latency_target_sweep.py 2>&1 | grep -v amdgpu.ids > $O/latency_sweep.txt
cat $O/latency_sweep.txt
