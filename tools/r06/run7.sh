#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run7; mkdir -p $O
export TMPDIR=/tmp
for L in conv2_1 conv3_1 conv4_1; do
  python3 tools/s2g_traffic_probe.py $L 2>/dev/null | tail -1 >> $O/s2g_traffic.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C -d $O/p_$L_$C -o p --output-format csv -- python3 tools/s2g_traffic_probe.py $L > /dev/null 2>&1
    python3 - $(find $O/p_$L_$C -name "*counter_collection.csv" | head -1) $C >> $O/s2g_traffic.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "s2" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:60]][0] += 1; d[r["Kernel_Name"][:60]][1] += float(r["Counter_Value"])
f = 2048.0 if sys.argv[2] == "FETCH_SIZE" else 1024.0
for k, (n, v) in d.items():
    print("    %s %s: %.1f MB per launch (%d launches)" % (k, sys.argv[2], f * v / n / 1e6, n))
PY
    rm -rf $O/p_$L_$C
  done
done
cat $O/s2g_traffic.txt
