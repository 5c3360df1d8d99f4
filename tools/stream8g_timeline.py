#!/usr/bin/env python3
"""Where does a step of conv3x3_stream8g_kernel go?  Needs a library built with -DV2X_STREAM_DBG_BUILD=16 (tools/stream8g_timeline.sh does
that on the GPU box and restores the default build): lane 0 of one wave per group of workgroup 0 stamps s_memrealtime (100 MHz) at four
points of each step -- top of the load phase (0), loads issued and waits done (1), first barrier passed = MFMA phase begins (2), MFMA phase
done (3); the second barrier's exit is the next step's point 0.  Prints the mean duration of each span per group, in ns."""
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
    fetch = lib.v2x_debug_stream_timeline
    fetch.restype, fetch.argtypes = C.c_int, [C.c_void_p]
    dev = torch.device("cuda:0")
    for name, cin, cout, hw, n in (("384->128 @64 (36 steps per tile)", 384, 128, 64, 320), ("128->128 @64 (12 steps per tile)", 128, 128, 64, 320),
                                   ("768->256 @32 (72 steps per tile)", 768, 256, 32, 320)):
        w = torch.randn(cout, cin, 3, 3) * 0.05
        pc = packing.pack_conv_stream(name, w, torch.ones(cout), torch.zeros(cout), C0=cin, relu=True, device=dev)
        x = torch.randn(n, hw, hw, cin, device=dev).to(torch.bfloat16)
        for _ in range(3):
            y = ops.conv2d(pc, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = ops.conv2d(pc, x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        flops = 2.0 * n * hw * hw * cout * 9 * cin
        buf = np.zeros(2 * 128 * 4, dtype=np.uint32)
        rc = fetch(buf.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        t = buf.reshape(2, 128, 4).astype(np.int64) * 10            # ns
        print("%s: %.1f us, %.0f TFLOP/s (instrumented build)" % (name, us, flops / us / 1e6))
        for g in range(2):
            tg = t[g]
            valid = tg[:, 3] > 0
            k = int(valid.sum())
            tg = tg[:k]
            load = tg[:, 1] - tg[:, 0]
            bar1 = tg[:, 2] - tg[:, 1]
            mfma = tg[:, 3] - tg[:, 2]
            bar2 = tg[1:, 0] - tg[:-1, 3]
            period = tg[1:, 0] - tg[:-1, 0]
            sl = slice(4, k - 1)
            print("  group %d (%s): period %.0f ns | load phase %.0f | wait at barrier 1 %.0f | MFMA phase (96 MFMAs) %.0f | barrier 2 + loop %.0f   [median period %.0f, max %.0f]"
                  % (g, "patch filler" if g == 0 else "weight streamer", period[sl].mean(), load[sl].mean(), bar1[sl].mean(), mfma[sl].mean(), bar2[sl].mean(),
                     np.median(period[sl]), period[sl].max()))
        # tile boundaries (round 4): steps S3 - 1 -> S3: "barrier 2 + loop" of the last step of a tile contains the epilogue, its stores and the tile switch
        S3 = 3 * cin // 32
        for g in range(2):
            tg = t[g]
            k = int((tg[:, 3] > 0).sum())
            tg = tg[:k]
            period = tg[1:, 0] - tg[:-1, 0]
            bar2 = tg[1:, 0] - tg[:-1, 3]
            load = tg[:, 1] - tg[:, 0]
            bar1 = tg[:, 2] - tg[:, 1]
            last = [i for i in range(S3 - 1, k - 2, S3)]
            inner = [i for i in range(4, k - 2) if (i + 1) % S3 and i % S3 > 1]
            if last:
                print("  group %d tile boundary: epilogue + tile switch (barrier 2 + loop of a tile's last step) %.0f ns against %.0f inside a tile; the next tile's first step: load phase %.0f "
                      "(inner %.0f), wait at barrier 1 %.0f (inner %.0f); period of a tile's last step %.0f, of the next tile's first two steps %.0f / %.0f (inner %.0f)"
                      % (g, np.mean([bar2[i] for i in last]), np.mean([bar2[i] for i in inner]), np.mean([load[i + 1] for i in last]), np.mean([load[i] for i in inner]),
                         np.mean([bar1[i + 1] for i in last]), np.mean([bar1[i] for i in inner]), np.mean([period[i] for i in last]), np.mean([period[i + 1] for i in last]),
                         np.mean([period[i + 2] for i in last if i + 2 < len(period)]), np.mean([period[i] for i in inner])))
        print("  96 MFMAs back to back = %.0f ns at 2.1 GHz" % (96 * 16 / 2.1))


if __name__ == "__main__":
    main()
