#!/bin/bash
# paired in-process A/B of the one-wave-per-SIMD form (STREAM_W1) against stream8g on the bench layers, 320 maps
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
for o in conv6_1 conv5_1 conv3_2 conv6_2 ConvGRU; do
    timeout 300 python3 tools/ab_inproc.py $L $L A:STREAM_W1=0 B:STREAM_W1=2 only=$o 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_w1_ab.txt
