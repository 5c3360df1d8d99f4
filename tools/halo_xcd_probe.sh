#!/bin/bash
# Halo kernels with the XCD-contiguous tile walk (HALO_XCD=1) against round-robin: paired time (tools/ab_inproc.py, 40 maps of 256x256 per launch) and
# fabric read bytes per kernel of a bench step (rocprofv3 --pmc FETCH_SIZE, 320 maps per launch).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
python3 tools/ab_inproc.py $L $L A:HALO_XCD=0 B:HALO_XCD=1 only=halo 2>&1 | grep -v amdgpu.ids
for w in 0 1; do
  rm -rf /tmp/hx$w
  V2X_HALO_XCD=$w rocprofv3 --pmc FETCH_SIZE -d /tmp/hx$w -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-extras --no-roofline > /dev/null 2>&1
  python3 - $(find /tmp/hx$w -name "*counter_collection.csv") $w <<'PY'
import collections, csv, sys
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and ("halo" in r["Kernel_Name"] or "pair_bits" in r["Kernel_Name"]):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
for k, (n, tot) in sorted(d.items()):
    print("HALO_XCD=%s  %-48s FETCH_SIZE %7.1f MB per launch (%d launches)" % (sys.argv[2], k, 2.0 * 1024.0 * tot / n / 1e6, n))
PY
done
