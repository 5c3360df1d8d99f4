#!/bin/bash
# Round 3: why is the 32x32x16 form of stream8g slower?  co-issue probe (companion wave beside 32x32 MFMAs), SQ counters of both forms, prefetch-point / priority variants.
cd "$(dirname "$0")/.."
O=gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue_probe.hip -o /tmp/coissue && timeout 300 /tmp/coissue > $O/coissue.txt 2>&1
cat $O/coissue.txt
for m in 0 1; do
  for w in half gru; do
    export V2X_STREAM_M32=$m
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq_${w}_$m -o sq --output-format csv -- python3 tools/conv_layer_run.py $w > $O/sq_${w}_$m.log 2>&1
    python3 tools/pmc_sq_summary.py $(find $O/sq_${w}_$m -name "*counter_collection.csv") > $O/sq_${w}_m32_$m.csv 2>/dev/null
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM -d $O/sq2_${w}_$m -o sq --output-format csv -- python3 tools/conv_layer_run.py $w > $O/sq2_${w}_$m.log 2>&1
    python3 - $(find $O/sq2_${w}_$m -name "*counter_collection.csv") > $O/sq2_${w}_m32_$m.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "stream8g" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in agg.items(): print(k, dict(a))
PY
    rm -rf $O/sq_${w}_$m $O/sq2_${w}_$m
  done
done
unset V2X_STREAM_M32
grep stream8g $O/sq_*_m32_*.csv; cat $O/sq2_*_m32_*.txt
for v in "-DV2X_STREAM_M32PF_BUILD=1" "-DV2X_STREAM_M32PF_BUILD=4" "-DV2X_STREAM_PRIO_BUILD=1"; do
  echo "== A: default (M32, PF 2)   B: $v" | tee -a $O/ab_variants.txt
  timeout 1200 bash tools/ab_inproc.sh "" "$v" 2>&1 | head -5 | tee -a $O/ab_variants.txt
done
