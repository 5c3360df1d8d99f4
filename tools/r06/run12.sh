#!/bin/bash
cd "$(dirname "$0")/../.."
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -1 gpurun_out/r06_smoke.txt
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -3 gpurun_out/r06_profile_round.log
bash tools/profile_round.sh --config v2vnet r06 > gpurun_out/r06_profile_v2vnet.log 2>&1
tail -25 gpurun_out/r06_profile_v2vnet.log
