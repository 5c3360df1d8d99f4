#!/bin/bash
# Runs on the GPU box.  usage: tools/ab_inproc.sh "<flags of variant A>" "<flags of variant B>"      e.g.  "" "-DV2X_STREAM_H1_BUILD=2"
# AB_CSRC_A=<dir>: build variant A from that copy of csrc/ (its include of ../../include/v2x_amd.h must resolve: keep it two levels below the repo root).
# Builds libv2x_amd.so twice into /tmp and alternates the two inside one process (tools/ab_inproc.py); the in-tree build is not touched.
cd "$(dirname "$0")/.."
for v in A B; do
    if [ $v = A ]; then X="$1"; else X="$2"; fi
    rm -rf /tmp/ab_$v; mkdir -p /tmp/ab_$v/build
    SRC=v2x-sim_amd/csrc
    if [ $v = A ] && [ -n "${AB_CSRC_A:-}" ]; then SRC=$AB_CSRC_A; fi      # variant A from another source directory (e.g. the previous commit's csrc)
    ( cd $SRC && for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $X -c $f -o /tmp/ab_$v/build/${f%.hip}.o & done; wait
      /opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link -shared -fPIC /tmp/ab_$v/build/*.o -o /tmp/ab_$v/libv2x_amd_$v.so ) 2>&1 | grep -E "error" 
done
shift 2
python3 tools/ab_inproc.py /tmp/ab_A/libv2x_amd_A.so /tmp/ab_B/libv2x_amd_B.so "$@" 2>&1 | grep -v amdgpu.ids
