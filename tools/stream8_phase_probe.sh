#!/bin/bash
# Which phase bounds the 8-wave streamed kernel's step?  Rebuilds libv2x_amd.so on the GPU box with one phase compiled out at a
# time (the generated probe copy of conv_stream.hip: V2X_STREAM_DBG_BUILD; results are garbage in those builds) and times the dominant kernel.
#   bash tools/stream8_phase_probe.sh   ->  gpurun_out/stream8_probe.txt   (the default build is restored at the end, also when interrupted)
cd "$(dirname "$0")/.."
. tools/probe_env.sh
OUT=gpurun_out/stream8_probe.txt
mkdir -p gpurun_out; : > $OUT
for d in 0 1 2 3 4 8 7; do
    probe_build conv_stream "-DV2X_STREAM_DBG_BUILD=$d"
    python bench.py --no-cpu-baseline --no-gpu-baseline --no-live-traffic --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('dbg=$d', {n.replace('conv3x3_','').replace('_kernel',''): round(v['us_per_step']/v['launches_per_step'],1) for n,v in k.items() if 'stream8' in n})" >> $OUT
done
cat $OUT
