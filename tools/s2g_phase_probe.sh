#!/bin/bash
# Which phase bounds conv3x3_s2g_kernel's step?  Rebuilds libv2x_amd.so on the GPU box with one phase compiled out at a time
# (conv_stream_s2.hip: V2X_S2G_DBG_BUILD; results are garbage in those builds) and times the three stride-2 layers at 320 maps.
#   bash tools/s2g_phase_probe.sh [values...]  ->  gpurun_out/s2g_probe.txt   (the default build is restored at the end)
cd "$(dirname "$0")/.."
OUT=gpurun_out/s2g_probe.txt
mkdir -p gpurun_out; : > $OUT
VALS=${@:-0 1 2 3 4 8 16 20 32 63}
L=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
for d in $VALS; do
    rm -f v2x-sim_amd/csrc/build/conv_stream_s2.o
    make -s -C v2x-sim_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DV2X_S2G_DBG_BUILD=$d" > /dev/null 2>&1
    echo "dbg=$d" >> $OUT
    python3 tools/ab_inproc.py $L $L A:S2_G=0 B:S2_G=1 only=s2 2>&1 | grep "^s2" | sed -e 's/outputs.*//' >> $OUT
done
rm -f v2x-sim_amd/csrc/build/conv_stream_s2.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
cat $OUT
