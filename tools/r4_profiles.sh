#!/bin/bash
# round 4: the rocprofv3 passes of the default bench workload on the final code (the per-config passes: tools/profile_round.sh --config <name> <tag>)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 bash tools/profile_round.sh ${1:-r04g} > gpurun_out/${1:-r04g}_profile.log 2>&1
tail -5 gpurun_out/${1:-r04g}_profile.log
