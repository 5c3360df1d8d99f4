#!/bin/bash
# pair_bits occupancy experiment: the production library (2 workgroups per CU) against a build whose pair kernel asks for 10 KiB more LDS (1 workgroup per CU)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run13; mkdir -p $O
python3 tools/ab_inproc.py v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so tools/r06/ab/libv2x_amd_pair_1wg.so only=pair 2>&1 | grep -v amdgpu.ids | tee $O/pair_occupancy.txt
