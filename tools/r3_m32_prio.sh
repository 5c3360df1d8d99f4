#!/bin/bash
# Round 3: the LDS-DMA of a companion wave is starved beside back-to-back 32x32x16 MFMAs (12.9x, tools/mfma_coissue_probe.hip).  Does a raised
# priority of the load-phase wave help?  (probe with -DCOMP_PRIO=3, and the kernel's V2X_STREAM_LPRIO_BUILD.)
cd "$(dirname "$0")/.."
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DCOMP_PRIO=3 tools/mfma_coissue_probe.hip -o /tmp/coissue3 && timeout 300 /tmp/coissue3 > $O/coissue_prio3.txt 2>&1
cat $O/coissue_prio3.txt
for v in "-DV2X_STREAM_LPRIO_BUILD=1" "-DV2X_STREAM_LPRIO_BUILD=3"; do
  echo "== A: M32 default   B: M32 $v" | tee -a $O/ab_lprio.txt
  timeout 1200 bash tools/ab_inproc.sh "" "$v" 2>&1 | head -5 | tee -a $O/ab_lprio.txt
done
