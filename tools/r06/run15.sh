#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bits_input.py tests/test_gpu_direct_oracle.py -m gpu -q > $O/t.txt 2>&1; tail -3 $O/t.txt
python3 tools/ab_inproc.py tools/r06/ab/libv2x_amd_pair_old.so v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so only=pair 2>&1 | grep -v amdgpu.ids | tee $O/pair_diet_ab.txt
python3 tools/ab_inproc.py tools/r06/ab/libv2x_amd_pair_old.so v2x-sim_amd/v2x_sim_amd/lib/libv2x_amd.so only=pair 2>&1 | grep -v amdgpu.ids | tee -a $O/pair_diet_ab.txt
