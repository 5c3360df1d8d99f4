#!/bin/bash
# Runs on the GPU box: build conv_tail.o from the probe copy with time stamps, run tools/tail_timeline.py, restore the product build.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f v2x-sim_amd/csrc/build/conv_tail.o
make -s -C v2x-sim_amd/csrc PROBE=conv_tail FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DV2X_TAIL_DBG_BUILD=32" > /dev/null 2>&1
python3 tools/tail_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tail_timeline.txt
rm -f v2x-sim_amd/csrc/build/conv_tail.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
