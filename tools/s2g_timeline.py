#!/usr/bin/env python3
"""Where does a step of conv3x3_s2g_kernel go?  Needs a library built with -DV2X_S2G_DBG_BUILD=64 (tools/s2g_timeline.sh builds it on the GPU
box and restores the default build): lane 0 of wave 0 of each group of workgroup 0 stamps the shader clock at six points of each step --
0 top of the load phase, 1 DMAs issued, 2 fragments read and load-phase wait done, 3 first barrier passed, 4 MFMAs done, 5 end-of-phase wait
done; the second barrier's exit is the next step's point 0.  Prints mean cycles per span, per group and per step of the chunk (j = 0, 1, 2)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from v2x_sim_amd import _lib, ops, packing
    lib = _lib.load()
    fetch = lib.v2x_debug_s2g_timeline
    fetch.restype, fetch.argtypes = C.c_int, [C.c_void_p]
    dev = torch.device("cuda:0")
    for name, cin, cout, hw, n in (("conv2_1 64->128 @128", 64, 128, 128, 160), ("conv3_1 128->256 @64", 128, 256, 64, 320), ("conv4_1 256->512 @32", 256, 512, 32, 320)):
        w = torch.randn(cout, cin, 3, 3) * 0.05
        pc = packing.pack_conv_stream(name, w, torch.ones(cout), torch.zeros(cout), C0=cin, relu=True, stride=2, device=dev)
        x = torch.randn(n, hw, hw, cin, device=dev).to(torch.bfloat16)
        for _ in range(3):
            ops.conv2d(pc, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(pc, x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        buf = np.zeros(2 * 96 * 8, dtype=np.uint32)
        assert fetch(buf.ctypes.data_as(C.c_void_p)) == 0
        t = buf.reshape(2, 96, 8).astype(np.int64)
        print("%s: %.1f us (instrumented build), %d steps per tile" % (name, us, cin // 32 * 3))
        for g in range(2):
            tg = t[g]
            k = int((tg[:, 5] != 0).sum())
            tg = tg[:k]
            d = lambda a, b: (tg[:, b] - tg[:, a]) & 0xffffffff
            nxt = (tg[1:, 0] - tg[:-1, 5]) & 0xffffffff
            per = (tg[1:, 0] - tg[:-1, 0]) & 0xffffffff
            for j in range(3):
                sl = np.arange(3 + j, k - 1, 3)
                print("  group %d step j=%d: period %5.0f | DMA issue %4.0f | frag reads + wait %4.0f | barrier 1 %4.0f | MFMA phase %4.0f | end wait %4.0f | barrier 2 + loop %4.0f"
                      % (g, j, per[sl].mean(), d(0, 1)[sl].mean(), d(1, 2)[sl].mean(), d(2, 3)[sl].mean(), d(3, 4)[sl].mean(), d(4, 5)[sl].mean(), nxt[sl].mean()))
        print("  48 MFMAs back to back = 768 cycles")


if __name__ == "__main__":
    main()
