#!/bin/bash
# Runs on the GPU box: build libv2x_amd.so with the timestamp instrumentation, run tools/stream8g_timeline.py, restore the default build.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f v2x-sim_amd/csrc/build/conv_stream.o
make -s -C v2x-sim_amd/csrc PROBE=conv_stream FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DV2X_STREAM_DBG_BUILD=16" > /dev/null 2>&1
python3 tools/stream8g_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream8g_timeline.txt
rm -f v2x-sim_amd/csrc/build/conv_stream.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
