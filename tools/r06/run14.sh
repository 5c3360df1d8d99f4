#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run14; mkdir -p $O
F="--no-extras --no-cpu-baseline --no-gpu-baseline --no-calibration --no-live-traffic --no-roofline --no-shard-check"
for rep in 1 2 3; do
 for steps in 20 60; do
  for m in "1 0" "5 0" "5 1"; do
    set -- $m
    V2X_BENCH_STAGGER=$2 python3 bench.py $F --graph $1 --steps $steps 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('graph $1 stagger $2 steps $steps: %.1f frames/s  %.3f ms/step  graph_equals_eager %s' % (d['value'], d['ms_per_step'], d['graph_equals_eager']))" >> $O/free_run_ab.txt
  done
 done
done
sort $O/free_run_ab.txt
