#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run8; mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fetch_calib_probe.hip -o /tmp/fetch_calib 2> $O/build.err
rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- /tmp/fetch_calib > $O/run.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- /tmp/fetch_calib >> $O/run.log 2>&1
python3 - $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/kt -name "*kernel_stats.csv" | head -1) > $O/fetch_calib.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        d[r["Kernel_Name"]][0] += 1; d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
t = {r["Name"]: float(r["TotalDurationNs"]) / int(r["Calls"]) for r in csv.DictReader(open(sys.argv[2]))}
print("kernel (reads 1 GiB = 1073.7 MB exactly once)            FETCH_SIZE raw MB   raw / true    us per launch   TB/s")
for k, (n, v) in sorted(d.items()):
    raw = 1024.0 * v / n
    us = t.get(k, 0.0) / 1e3
    print("%-56s %12.1f %12.3f %14.1f %8.2f" % (k[:56], raw / 1e6, raw / float(1 << 30), us, (1 << 30) / max(us, 1e-9) / 1e6))
PY
cat $O/fetch_calib.txt
rm -rf $O/pf $O/kt
