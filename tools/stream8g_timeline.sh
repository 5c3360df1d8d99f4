#!/bin/bash
# Runs on the GPU box: build libv2x_amd.so with the timestamp instrumentation, run tools/stream8g_timeline.py, restore the default build (also when interrupted).
cd "$(dirname "$0")/.."
. tools/probe_env.sh
mkdir -p gpurun_out
probe_build conv_stream "-DV2X_STREAM_DBG_BUILD=16"
python3 tools/stream8g_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream8g_timeline.txt
