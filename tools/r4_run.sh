#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_gpu_suite_e.txt
timeout 900 python bench.py > gpurun_out/r04_bench_n1_i.json 2> gpurun_out/r04_bench_n1_i.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -3 gpurun_out/r04_gpu_suite_e.txt; wc -l gpurun_out/r04_bench_n1_i.json; cut -c1-200 gpurun_out/r04_bench_n1_i.json
