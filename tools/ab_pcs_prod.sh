#!/bin/bash
# usage (GPU box): tools/ab_pcs_prod.sh "<flags A>" "<flags B>" ...   -- rebuilds the PRODUCTION conv_stream_pc.o with each flag set and times the streamed
# parity-class layers against their 9-tap forms (tools/ab_parity_class.py: the 9-tap kernel is the common yardstick of the variants)
cd "$(dirname "$0")/.."
for X in "$@"; do
    rm -f v2x-sim_amd/csrc/build/conv_stream_pc.o
    make -s -C v2x-sim_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $X" > /dev/null 2>&1
    echo "== flags: $X"
    python3 tools/ab_parity_class.py 320 20 2>&1 | grep -E "conv5_1|conv6_1|conv7_1"
done
rm -f v2x-sim_amd/csrc/build/conv_stream_pc.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
