#!/bin/bash
# usage (GPU box): tools/ab_tail_builds.sh "<flags A>" "<flags B>" ...   -- rebuilds conv_tail.o from the generated probe copy (tools/probes/gen_probe.sh) with each flag set and times the detection tail (tools/ab_tail.py)
cd "$(dirname "$0")/.."
. tools/probe_env.sh
for X in "$@"; do
    probe_build conv_tail "$X"
    echo "== flags: $X"
    python3 tools/ab_tail.py 320 20 2>&1 | grep -E "detection tail"
done
