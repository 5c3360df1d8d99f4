#!/bin/bash
# usage (GPU box): tools/ab_tail_builds.sh "<flags A>" "<flags B>" ...   -- rebuilds conv_tail.o from tools/probes/conv_tail_probe.hip with each flag set and times the detection tail (tools/ab_tail.py)
cd "$(dirname "$0")/.."
for X in "$@"; do
    rm -f v2x-sim_amd/csrc/build/conv_tail.o
    make -s -C v2x-sim_amd/csrc PROBE=conv_tail FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $X" > /dev/null 2>&1
    echo "== flags: $X"
    python3 tools/ab_tail.py 320 20 2>&1 | grep -E "detection tail"
done
rm -f v2x-sim_amd/csrc/build/conv_tail.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
