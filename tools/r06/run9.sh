#!/bin/bash
# ConvGRU fabric reads split into their two streams (round 6): FETCH_SIZE of the production kernel, of a build without the WEIGHT DMAs (patches only) and of a build without
# the PATCH DMAs (weights only) -- the weight pieces are wide 1-KiB reads (FETCH_SIZE x 2), the patch pieces 64-byte segments (x ~1.08: tools/fetch_calib_probe.hip)
cd "$(dirname "$0")/../.."
. tools/probe_env.sh
O=gpurun_out/r06_run9; mkdir -p $O
export TMPDIR=/tmp
run() {
  rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/conv_layer_run.py gru > /dev/null 2>&1
  python3 - $(find $O/pf -name "*counter_collection.csv" | head -1) "$1" >> $O/gru_fetch_split.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "stream8g" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:70]][0] += 1; d[r["Kernel_Name"][:70]][1] += float(r["Counter_Value"])
for k, (n, v) in d.items():
    print("%-28s %s: FETCH_SIZE raw %.1f MB per launch (%d launches, 160 maps)" % (sys.argv[2], k, 1024.0 * v / n / 1e6, n))
PY
  rm -rf $O/pf
}
run "production"
probe_build conv_stream "-DV2X_STREAM_DBG_BUILD=1"
run "no weight DMAs (patches)"
probe_build conv_stream "-DV2X_STREAM_DBG_BUILD=2"
run "no patch DMAs (weights)"
cat $O/gru_fetch_split.txt
