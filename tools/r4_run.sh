#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
F="--steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-calibration --no-shard-check --no-roofline"
{
for i in 1 2 3; do
  for wt in 1 2; do
    V2X_STREAM_WT=$wt timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('STREAM_WT=$wt', round(r['value'],1), 'frames/s', round(r['ms_per_step'],3), 'ms')"
  done
done
} > gpurun_out/r04_gru_wt_ab.txt 2>&1
cat gpurun_out/r04_gru_wt_ab.txt
