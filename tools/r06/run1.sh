#!/bin/bash
# round-6 GPU call 1: the new parity tests, the default bench line through the new N = 1 orchestrator, and the --pmc order A/B
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_direct_oracle.py tests/test_gpu_decisions.py tests/test_gpu_bits_input.py "tests/test_gpu_stages.py::test_voxelize_ring_sweep_bit_exact" -m gpu -q -s -x > $O/new_tests.txt 2>&1
echo "new tests rc=$?" | tee -a $O/new_tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" >> $O/bench_default.err
F="--no-extras --no-cpu-baseline --no-gpu-baseline --no-calibration --no-shard-check"
for i in 1 2 3; do
  for order in before after; do
    V2X_BENCH_PMC_ORDER=$order timeout 600 python bench.py $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$order run $i: %.1f frames/s  %.3f ms/step  traffic_source: %s' % (d['value'], d['ms_per_step'], (d['roofline'].get('traffic_source') or '')[:40]))" >> $O/pmc_order_ab.txt
  done
done
V2X_BENCH_LIVE_TRAFFIC=0 timeout 600 python bench.py $F 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no passes at all: %.1f frames/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> $O/pmc_order_ab.txt
cat $O/pmc_order_ab.txt
tail -c 2500 $O/bench_default.json
