#!/bin/bash
# usage: tools/probes/gen_probe.sh <name> [out.hip]
# Generates the INSTRUMENTED copy of v2x-sim_amd/csrc/<name>.hip (phase-removal switches, time stamps: tools/*_probe.sh, tools/ab_tail_builds.sh) from the
# production source and the committed patch tools/probes/<name>_probe.patch -- the probe is never a hand-maintained copy: when the
# production kernel changes the patch either still applies (the probe follows) or fails here, and tests/test_probes_cpu.py checks that the generated file
# built WITHOUT any -DV2X_*_DBG_BUILD flag has the production object's instruction stream.  Results of instrumented builds are garbage by design.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
NAME="$1"
OUT="${2:-$ROOT/tools/probes/${NAME}_probe.hip}"
[ -f "$ROOT/tools/probes/${NAME}_probe.patch" ] || { echo "no patch for $NAME (have: $(cd "$ROOT/tools/probes" && ls *.patch | sed 's/_probe.patch//' | tr '\n' ' '))" >&2; exit 2; }
patch -s -o "$OUT" "$ROOT/v2x-sim_amd/csrc/${NAME}.hip" "$ROOT/tools/probes/${NAME}_probe.patch"
echo "$OUT"
