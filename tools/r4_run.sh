#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/tail_chunk_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_tail_chunk_probe.txt
cat gpurun_out/r04_tail_chunk_probe.txt
