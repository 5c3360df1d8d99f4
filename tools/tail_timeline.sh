#!/bin/bash
# Runs on the GPU box: build conv_tail.o from the generated probe copy with time stamps, run tools/tail_timeline.py, restore the product build (also when interrupted).
cd "$(dirname "$0")/.."
. tools/probe_env.sh
mkdir -p gpurun_out
probe_build conv_tail "-DV2X_TAIL_DBG_BUILD=32"
python3 tools/tail_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tail_timeline.txt
