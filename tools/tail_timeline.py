#!/usr/bin/env python3
"""Where does a tile of conv3x3_tail_kernel go?  Needs the probe build with time stamps (tools/tail_timeline.sh): wave 0 of each group of workgroup 0
stamps s_memrealtime (100 MHz) at five points of 32 iterations -- top (0), stage A done (1), barrier X passed (2), stage B done (3), barrier Y passed (4)."""
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
    from v2x_sim_amd import _lib, ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    lib = _lib.load()
    fetch = lib.v2x_debug_tail_stamps
    fetch.restype, fetch.argtypes = C.c_int, [C.c_void_p]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = FaFNet(Config("test")).to(dev).eval()
    pk = m.packed(dev)
    last, heads = pk["dec"][-1], pk["heads"]
    x = torch.relu(torch.randn(320, 256, 256, 32)).to(torch.bfloat16).to(dev)
    for _ in range(3):
        ops.conv2d_tail(last.halo, heads.halo, x, heads.split)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 32 * 5, dtype=np.uint32)
    assert fetch(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(2, 32, 5).astype(np.int64) * 10            # ns
    for g in range(2):
        tg = t[g]
        a, wx, b, wy = tg[:, 1] - tg[:, 0], tg[:, 2] - tg[:, 1], tg[:, 3] - tg[:, 2], tg[:, 4] - tg[:, 3]
        per = np.diff(tg[:, 0])
        print("group %d: stage A %.0f ns, wait at X %.0f, stage B (+ window DMA issue, stores) %.0f, wait at Y %.0f; period %.0f ns (median over 32 tiles)"
              % (g, np.median(a), np.median(wx), np.median(b), np.median(wy), np.median(per)))


if __name__ == "__main__":
    main()
