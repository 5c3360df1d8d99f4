#!/bin/bash
# usage (GPU box): tools/ab_pcs_prod.sh "<flags A>" "<flags B>" ...   -- rebuilds the PRODUCTION conv_stream_pc.o with each flag set and times the streamed
# parity-class layers against their 9-tap forms (tools/ab_parity_class.py: the 9-tap kernel is the common yardstick of the variants)
cd "$(dirname "$0")/.."
. tools/probe_env.sh
for X in "$@"; do
    prod_build conv_stream_pc "$X"
    echo "== flags: $X"
    python3 tools/ab_parity_class.py 320 20 2>&1 | grep -E "conv5_1|conv6_1|conv7_1"
done
