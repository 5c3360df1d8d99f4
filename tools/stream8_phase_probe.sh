#!/bin/bash
# Which phase bounds the 8-wave streamed kernel's step?  Rebuilds libv2x_amd.so on the GPU box with one phase compiled out at a
# time (tools/probes/conv_stream_probe.hip: V2X_STREAM_DBG_BUILD; results are garbage in those builds) and times the dominant kernel.
#   bash tools/stream8_phase_probe.sh   ->  gpurun_out/stream8_probe.txt   (the default build is restored at the end)
cd "$(dirname "$0")/.."
OUT=gpurun_out/stream8_probe.txt
mkdir -p gpurun_out; : > $OUT
for d in 0 1 2 3 4 8 7; do
    rm -f v2x-sim_amd/csrc/build/conv_stream.o
    make -s -C v2x-sim_amd/csrc PROBE=conv_stream FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DV2X_STREAM_DBG_BUILD=$d" > /dev/null 2>&1
    python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('dbg=$d', {n.replace('conv3x3_','').replace('_kernel',''): round(v['us_per_step']/v['launches_per_step'],1) for n,v in k.items() if 'stream8' in n})" >> $OUT
done
rm -f v2x-sim_amd/csrc/build/conv_stream.o
make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
cat $OUT
