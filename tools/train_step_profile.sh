#!/bin/bash
# per-kernel time of one training step (all launches: this library's and PyTorch's elementwise / reduction / Adam kernels); round 6: + HBM bytes per kernel
# (separate rocprofv3 --pmc passes) and the step's totals against both roofs (tools/train_traffic_table.py)
cd "$(dirname "$0")/.."
O=gpurun_out/train_prof
mkdir -p $O
export TMPDIR=/tmp
for m in faf v2v faf40; do
  A=$m; F=2
  if [ $m = faf40 ]; then A=faf; F=8; fi
  rocprofv3 --kernel-trace --stats -d $O/$m -o t --output-format csv -- python3 tools/train_step_run.py $A $F > $O/$m.log 2>&1
  cp $(find $O/$m -name "*kernel_stats.csv") $O/${m}_kernel_stats.csv
  python3 - $O/${m}_kernel_stats.csv > $O/${m}_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.2f ms (24 steps traced, the first one with the lazy packings and the optimizer-state fills)" % (tot / 24e6))
lib = [r for r in rows if any(t in r["Name"].lower() for t in ("miopen", "cijk", "naive_conv", "igemm_fwd", "igemm_bwd", "igemm_wrw", "gridwise"))]
print("vendor-library convolution / GEMM kernels in the step: %s" % (", ".join(sorted(set(r["Name"][:60] for r in lib))) or "none"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%8.1f us/step  x%-5.1f %5.1f %%  %s" % (float(r["TotalDurationNs"]) / 24e3, int(r["Calls"]) / 24.0, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
PY
  rm -rf $O/$m
done
# HBM bytes per kernel: FaFNet at 10 and at 40 maps (counters only; 4 steps each)
for m in faf faf40; do
  F=2; MAPS=10
  if [ $m = faf40 ]; then F=8; MAPS=40; fi
  V2X_TRAIN_RUN_STEPS=4 rocprofv3 --pmc FETCH_SIZE -d $O/pf_$m -o p --output-format csv -- python3 tools/train_step_run.py faf $F > $O/pf_$m.log 2>&1
  V2X_TRAIN_RUN_STEPS=4 rocprofv3 --pmc WRITE_SIZE -d $O/pw_$m -o p --output-format csv -- python3 tools/train_step_run.py faf $F > $O/pw_$m.log 2>&1
  python3 tools/train_traffic_table.py $O/${m}_kernel_stats.csv 24 $(find $O/pf_$m -name "*counter_collection.csv" | head -1) $(find $O/pw_$m -name "*counter_collection.csv" | head -1) 4 $MAPS $O/${m}_traffic.json > $O/${m}_traffic.txt 2>&1
  rm -rf $O/pf_$m $O/pw_$m
done
head -50 $O/faf_kernels.txt; head -50 $O/v2v_kernels.txt; head -30 $O/faf40_kernels.txt; cat $O/faf40_traffic.txt; tail -3 $O/faf_traffic.txt
