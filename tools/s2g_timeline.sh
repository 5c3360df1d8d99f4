#!/bin/bash
# Runs on the GPU box: build libv2x_amd.so with the s2g timestamp instrumentation, run tools/s2g_timeline.py, restore the default build.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f v2x-sim_amd/csrc/build/conv_stream_s2.o
make -s -C v2x-sim_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DV2X_S2G_DBG_BUILD=64" > /dev/null 2>&1
V2X_S2_G=2 python3 tools/s2g_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2g_timeline.txt
rm -f v2x-sim_amd/csrc/build/conv_stream_s2.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
