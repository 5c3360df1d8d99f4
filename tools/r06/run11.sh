#!/bin/bash
# ConvGRU (conv3x3_stream8g_kernel<96, 2>): cache policy of its two LDS-DMA streams -- paired in-process timing (tools/ab_inproc.py: in-tree library = A, variant = B)
# and FETCH_SIZE per launch (tools/conv_layer_run.py gru, 160 maps) for each variant built in tools/r06/ab/ (V2X_STREAM8G_PATCH_AUX / _WEIGHT_AUX: 2 = nt, 1 = sc0, 3 = sc0 nt)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run11; mkdir -p $O
export TMPDIR=/tmp
LIB=v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so
cp $LIB /tmp/libv2x_amd_prod.so
trap 'cp /tmp/libv2x_amd_prod.so '$LIB EXIT
fetch() {
  rocprofv3 --pmc FETCH_SIZE -d $O/pf -o p --output-format csv -- python3 tools/conv_layer_run.py gru > /dev/null 2>&1
  python3 - $(find $O/pf -name "*counter_collection.csv" | head -1) "$1" >> $O/gru_cache_policy.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "stream8g" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:70]][0] += 1; d[r["Kernel_Name"][:70]][1] += float(r["Counter_Value"])
for k, (n, v) in d.items():
    print("    %-14s FETCH_SIZE raw %.1f MB per launch (160 maps)" % (sys.argv[2], 1024.0 * v / n / 1e6))
PY
  rm -rf $O/pf
}
echo "== production (default policy on both streams)" >> $O/gru_cache_policy.txt
fetch production
for tag in PATCH_AUX_2 PATCH_AUX_1 PATCH_AUX_3 WEIGHT_AUX_2; do
  echo "== $tag" >> $O/gru_cache_policy.txt
  python3 tools/ab_inproc.py /tmp/libv2x_amd_prod.so tools/r06/ab/libv2x_amd_$tag.so only=ConvGRU 2>&1 | grep -v amdgpu.ids | grep -i "gru" >> $O/gru_cache_policy.txt
  cp tools/r06/ab/libv2x_amd_$tag.so $LIB
  fetch $tag
  cp /tmp/libv2x_amd_prod.so $LIB
done
cat $O/gru_cache_policy.txt
