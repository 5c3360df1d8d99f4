#!/bin/bash
# which companion work of the w1 step costs how much?  phase-removal builds against the full build (results garbage, timing only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r04_w1_phase.txt
: > $OUT
for d in 1 2 3 4 8 15; do
    echo "== B built with -DV2X_W1_DBG_BUILD=$d (1 no patch pieces, 2 no weight DMAs, 4 no wait/barrier, 8 no pixel-fragment reads); A = the full w1 kernel" >> $OUT
    timeout 600 bash tools/ab_inproc.sh "" "-DV2X_W1_DBG_BUILD=$d" A:STREAM_W1=2 B:STREAM_W1=2 only=conv5_1 2>&1 | grep "conv" | sed -e 's/outputs.*//' >> $OUT
    timeout 600 python3 tools/ab_inproc.py /tmp/ab_A/libv2x_amd_A.so /tmp/ab_B/libv2x_amd_B.so A:STREAM_W1=2 B:STREAM_W1=2 only=conv6_2 2>&1 | grep "conv" | sed -e 's/outputs.*//' >> $OUT
done
cat $OUT
