#!/bin/bash
# ConvGRU (stream8g<96, 2>) with the persistent grid walking 8 pixel x 4 channel tiles per XCD and round (GRU_XCD_WALK=1) against the default
# 4 x 8: paired time (tools/ab_inproc.py) and fabric read bytes (rocprofv3 --pmc FETCH_SIZE on tools/conv_layer_run.py gru, 160 maps).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
python3 tools/ab_inproc.py $L $L A:GRU_XCD_WALK=0 B:GRU_XCD_WALK=1 only=ConvGRU 2>&1 | grep -v amdgpu.ids
for w in 0 1; do
  rm -rf /tmp/gxw$w
  V2X_GRU_XCD_WALK=$w rocprofv3 --pmc FETCH_SIZE -d /tmp/gxw$w -o f --output-format csv -- python3 tools/conv_layer_run.py gru > /dev/null 2>&1
  python3 - $(find /tmp/gxw$w -name "*counter_collection.csv") $w <<'PY'
import csv, sys
n = tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "stream8g" in r["Kernel_Name"]:
        n += 1
        tot += float(r["Counter_Value"])
print("GRU_XCD_WALK=%s: FETCH_SIZE %.1f MB per launch over %d launches (KiB x 2: the gfx950 correction of tools/pmc_traffic.py; 160 maps)" % (sys.argv[2], 2.0 * 1024.0 * tot / max(n, 1) / 1e6, n))
PY
done
