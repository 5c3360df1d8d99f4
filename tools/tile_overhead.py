#!/usr/bin/env python3
"""Where does a persistent streamed launch spend its time beyond the K loop?  Times one layer at batch sizes that give every workgroup 1, 2, 5,
10 ... tiles and fits T(tiles) = L + tiles * tau; with two layers of the same map extent but different K (steps per tile) tau = steps * s + o
separates the per-step time s from the per-tile overhead o (epilogue, output stores, tile switch).  usage: python3 tools/tile_overhead.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from v2x_sim_amd import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cases = [("conv6_2: 128 -> 128 @64 (12 steps)", 128, 0, 128, 64, 0), ("conv6_1: (256 half + 128) -> 128 @64 (36 steps)", 256, 128, 128, 64, 1),
         ("conv3_2: 256 -> 256 @32 (24 steps)", 256, 0, 256, 32, 0), ("conv5_1: (512 half + 256) -> 256 @32 (72 steps)", 512, 256, 256, 32, 1)]
for name, c0, c1, cout, hw, up in cases:
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    pc = packing.pack_conv_stream(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0 if c1 else c0 + c1, C1=c1, up0=up,
                                  relu=True, device=dev)
    tiles_per_map = (hw // 16) * (hw // 32) * (cout // 128)
    rows = []
    for tiles in (1, 2, 3, 5, 10, 20):
        n = tiles * 256 // tiles_per_map
        if n < 1 or n * tiles_per_map != tiles * 256:
            continue
        if c1:
            h0 = hw // 2 if up else hw
            x0 = torch.relu(torch.randn(n, h0, h0, c0, generator=g)).to(torch.bfloat16).to(dev)
            x1 = torch.relu(torch.randn(n, hw, hw, c1, generator=g)).to(torch.bfloat16).to(dev)
        else:
            x0, x1 = torch.relu(torch.randn(n, hw, hw, c0, generator=g)).to(torch.bfloat16).to(dev), None
        ts = []
        for r in range(-3, 30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(pc, x0, x1)
            e1.record()
            torch.cuda.synchronize()
            if r >= 0:
                ts.append(e0.elapsed_time(e1) * 1e3)
        rows.append((tiles, n, float(np.median(ts))))
        del x0, x1
    A = np.array([[1.0, t] for t, _, _ in rows])
    y = np.array([us for _, _, us in rows])
    (L, tau), *_ = np.linalg.lstsq(A, y, rcond=None)
    print("%-52s" % name, "  ".join("%d tiles/WG (%d maps): %.1f us" % r for r in rows))
    print("%-52s fit: fixed L = %.1f us, per tile tau = %.2f us" % ("", L, tau))
