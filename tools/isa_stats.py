#!/usr/bin/env python3
"""ISA statistics of the gfx950 kernels of one source file (cross-compiled, no GPU): registers, scratch, LDS, and instruction counts
(MFMA, ds_read, LDS-DMA, vector stores, barriers, waits) per kernel.   python3 tools/isa_stats.py conv_halo.hip [name-substring] [-D...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2x-sim_amd", "csrc")


def main():
    src = sys.argv[1]
    pat = [a for a in sys.argv[2:] if not a.startswith("-")]
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"] + defs +
                              [os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        asm = open(out).read()
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)\.wavefront_size", asm, flags=re.S)}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", asm, flags=re.S | re.M):
        name, body = m.group(1), m.group(0)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat and not any(p in dem for p in pat):
            continue
        md = meta.get(name, "")
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, md) or [0, "?"])[1]
        cnt = lambda r: len(re.findall(r"^\s+%s" % r, body, flags=re.M))
        print("%s\n    vgpr %s agpr %s sgpr %s scratch %s lds %s | mfma %d ds_read %d ds_write %d lds_dma %d gload %d gstore %d barrier %d waitcnt %d valu %d salu %d lines %d"
              % (dem[:150], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"),
                 cnt("v_mfma"), cnt("ds_read"), cnt("ds_write"), cnt(r"global_load_lds"), cnt(r"global_load_(?!lds)"), cnt("global_store"), cnt("s_barrier"),
                 cnt("s_waitcnt"), cnt(r"v_(?!mfma)"), cnt("s_(?!waitcnt|barrier)"), body.count("\n")))


if __name__ == "__main__":
    main()
