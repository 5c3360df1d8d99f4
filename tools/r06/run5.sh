#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run5; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_suite.txt 2>&1
echo "rc=$?" >> $O/gpu_suite.txt; tail -5 $O/gpu_suite.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" >> $O/bench_default.err
tail -c 1800 $O/bench_default.json
