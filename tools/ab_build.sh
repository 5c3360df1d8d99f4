#!/bin/bash
# usage: tools/ab_build.sh <file.hip stem | all> <macro> <kernel-name filter, comma separated> <v1> <v2> ...
#   -> rebuilds with -D<macro>=<v> and runs bench for each value, printing the per-launch time of the kernels whose name contains a filter term
cd "$(dirname "$0")/.."
STEM=$1; MACRO=$2; FILTER=$3; shift 3
for rep in 1 2; do
for v in "$@"; do
    if [ "$STEM" = all ]; then rm -f v2x-sim_amd/csrc/build/*.o; else rm -f v2x-sim_amd/csrc/build/$STEM.o; fi
    make -s -j8 -C v2x-sim_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -D$MACRO=$v" > /dev/null 2>&1
    python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$MACRO=$v', round(d['value'],1), round(d['ms_per_step'],3), {n.replace('conv3x3_','').replace('_kernel',''): round(v['us_per_step']/v['launches_per_step'],1) for n,v in k.items() if any(t in n for t in ('$FILTER'.split(',')))})"
done
done
rm -f v2x-sim_amd/csrc/build/*.o
make -s -j8 -C v2x-sim_amd/csrc > /dev/null 2>&1
