#!/bin/bash
# Re-runs the paired in-process A/B comparisons DESIGN.md section 6 quotes and writes their raw output to gpurun_out/r02_paired_ab.txt
# (copied to profiles/ afterwards).  Runs on the GPU box: bash tools/ab_round2_evidence.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02_paired_ab.txt
mkdir -p gpurun_out; : > $OUT
run() { echo "=== $1" >> $OUT; shift; bash tools/ab_inproc.sh "$@" >> $OUT 2>&1; }
run "null comparison: A = B = default build" "" ""
run "stream8g wave tiling: A = V2X_STREAM_WT=0 (all channels x 64 pixels per wave), B = default (128-row layers tiled)" "" "" A:V2X_STREAM_WT=0 B:V2X_STREAM_WT=1
run "ConvGRU wave tiling: A = default (not tiled), B = V2X_STREAM_WT=2 (tiled)" "" "" A:V2X_STREAM_WT=1 B:V2X_STREAM_WT=2
run "wide3: A = V2X_WIDE3=0 (one tap per synchronisation), B = default (three)" "" "" A:V2X_WIDE3=0 B:V2X_WIDE3=1
run "stream8g vs stream8: A = V2X_STREAM_G=0, B = default" "" "" A:V2X_STREAM_G=0 B:V2X_STREAM_G=1
run "epilogue parameters in LDS: A = -DV2X_STREAM_LSS_BUILD=0, B = default" "-DV2X_STREAM_LSS_BUILD=0" ""
run "patch swizzle: A = default (pc>>1)&3, B = -DV2X_STREAM_PSWZ_BUILD=2 -DV2X_HALO_PSWZ_BUILD=2 -DV2X_S2_PSWZ_BUILD=2" "" "-DV2X_STREAM_PSWZ_BUILD=2 -DV2X_HALO_PSWZ_BUILD=2 -DV2X_S2_PSWZ_BUILD=2"
run "weight fragments two blocks ahead: A = default, B = -DV2X_STREAM_PF_BUILD=2" "" "-DV2X_STREAM_PF_BUILD=2"
run "load-phase priority: A = default, B = -DV2X_STREAM_LPRIO_BUILD=3" "" "-DV2X_STREAM_LPRIO_BUILD=3"
run "one channel tile per XCD: A = default, B = -DV2X_STREAM_XCDCO_BUILD=1" "" "-DV2X_STREAM_XCDCO_BUILD=1"
run "load phase reads fragments before issuing DMAs: A = default, B = -DV2X_STREAM_LORDER_BUILD=1" "" "-DV2X_STREAM_LORDER_BUILD=1"
cat $OUT
