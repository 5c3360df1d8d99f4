#!/bin/bash
# round 4, first GPU call: streaming / MFMA ceilings, per-config profiles, quick bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_mix_probe.hip -o /tmp/hbm_mix_probe && /tmp/hbm_mix_probe > gpurun_out/r04_hbm_mix_probe.txt 2>&1
tail -8 gpurun_out/r04_hbm_mix_probe.txt
python3 bench.py --no-extras > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -c 1500 gpurun_out/r04_bench_a.json
for c in when2com upperbound seg who2com lowerbound; do
    timeout 600 bash tools/profile_round.sh --config $c r04a > gpurun_out/r04a_$c.log 2>&1
    tail -30 gpurun_out/r04a_$c.log
done
