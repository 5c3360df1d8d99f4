#!/usr/bin/env python3
"""Paired in-process A/B of the parity-class form of a decoder `_1` layer against its 9-tap form: the same launch (320 maps, post-ReLU-like
operands) alternates between the two packings inside ONE process, so box drift cancels.   python3 tools/ab_parity_class.py [maps] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from v2x_sim_amd import ops, packing  # noqa: E402


def main():
    maps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, c0, c1, cout, hw in (("conv8_1", 64, 32, 32, 256), ("conv5_1", 512, 256, 256, 32), ("conv6_1", 256, 128, 128, 64), ("conv7_1", 128, 64, 64, 128)):
        w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * (2.0 / ((c0 + c1) * 9)) ** 0.5
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        if cout == 32:
            forms = {"9-tap": packing.pack_conv_halo(name, w, sc, sh, C0=c0, C1=c1, device=dev),
                     "parity": packing.pack_conv_halo_parity(name, w, sc, sh, C0=c0, C1=c1, device=dev)}
        else:
            forms = {"9-tap": packing.pack_conv_stream(name, w, sc, sh, C0=c0, C1=c1, up0=1, device=dev),
                     "parity": packing.pack_conv_stream_parity(name, w, sc, sh, C0=c0, C1=c1, device=dev)}
        x0 = torch.relu(torch.randn(maps, hw // 2, hw // 2, c0, generator=g)).to(torch.bfloat16).to(dev)
        x1 = torch.relu(torch.randn(maps, hw, hw, c1, generator=g)).to(torch.bfloat16).to(dev)
        out = torch.empty((maps, hw, hw, cout), dtype=torch.bfloat16, device=dev)
        for pc in forms.values():
            for _ in range(3):
                ops.conv2d(pc, x0, x1, out=out)
        torch.cuda.synchronize()
        t = {k: [] for k in forms}
        for _ in range(reps):
            for k, pc in forms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv2d(pc, x0, x1, out=out)
                e1.record()
                e1.synchronize()
                t[k].append(e0.elapsed_time(e1) * 1e3)
        a, b = np.array(t["9-tap"]), np.array(t["parity"])
        d = (b - a) / a
        gmac9, gmacp = maps * hw * hw * cout * 9 * (c0 + c1) / 1e9, maps * hw * hw * cout * (4 * c0 + 9 * c1) / 1e9
        print("%s at %d maps: 9-tap %.1f us (%.0f TFLOP/s)   parity-class %.1f us (%.0f TFLOP/s executed, %.0f TFLOP/s reference-equivalent)   paired diff %+.1f %% +- %.1f %%"
              % (name, maps, a.mean(), 2 * gmac9 / a.mean() * 1e3, b.mean(), 2 * gmacp / b.mean() * 1e3, 2 * gmac9 / b.mean() * 1e3, 100 * d.mean(),
                 100 * d.std() / np.sqrt(len(d))))


if __name__ == "__main__":
    main()
