#!/bin/bash
cd "$(dirname "$0")/../.."
for c in lowerbound upperbound when2com who2com seg; do
  bash tools/profile_round.sh --config $c r06 > gpurun_out/r06_profile_$c.log 2>&1
  head -1 gpurun_out/r06_profile_$c.log
done
