#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run38; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 600 $O/bench.json
