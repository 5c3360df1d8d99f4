#!/bin/bash
# round 4, second GPU call: the full GPU suite on the new dispatch rules, the bench line with calibration + self-check, the forced-dist path
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite_a.txt 2>&1
tail -15 gpurun_out/r04_gpu_suite_a.txt
python3 bench.py > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
tail -3 gpurun_out/r04_bench_b.err
python3 -c "
import json
r=json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','graph_equals_eager','sharded_equals_unsharded','calibration','latency','strong'): print(k, r.get(k))
print(r['roofline']['frac'], r['roofline'].get('frac_of_measured_ceiling'))
for k,v in r['configs']['configs'].items(): print(k, v)
print(r['training'])
"
V2X_FORCE_DIST=1 python3 bench.py --steps 10 --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_forced_dist.json 2> gpurun_out/r04_bench_forced_dist.err
tail -3 gpurun_out/r04_bench_forced_dist.err
python3 -c "
import json
r=json.loads(open('gpurun_out/r04_bench_forced_dist.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','graph_equals_eager','sharded_equals_unsharded','strong','exposed_exchange_ms_per_step'): print(k, r.get(k))
"
V2X_FORCE_DIST=1 python3 bench.py --steps 10 --scaling strong --frames-per-gpu 64 --no-cpu-baseline --no-roofline --no-calibration > gpurun_out/r04_bench_forced_dist_strong.json 2> gpurun_out/r04_bench_forced_dist_strong.err
tail -c 600 gpurun_out/r04_bench_forced_dist_strong.json
