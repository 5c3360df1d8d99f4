#!/usr/bin/env python3
"""How long does a stand-alone 1x1 layer take (the gather kernel)?  python3 tools/time_1x1.py  -- conv3d_2's shape (128 -> 128 @64x64, 320 maps) and conv3d_1's
(64 -> 64 @128x128): the price of UN-chaining them from conv2_2 / conv1_2."""
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
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, c, hw, n in (("conv3d_2: 128 -> 128 @64", 128, 64, 320), ("conv3d_1: 64 -> 64 @128", 64, 128, 320)):
        conv = torch.nn.Conv2d(c, c, 1)
        bn = torch.nn.BatchNorm2d(c).eval()
        lay = packing.layer_conv_bn(name, conv, bn, device=dev)
        x = torch.relu(torch.randn(n, hw, hw, c, generator=g)).to(torch.bfloat16).to(dev)
        for _ in range(3):
            ops.run_layer(lay, x)
        torch.cuda.synchronize()
        t = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.run_layer(lay, x)
            e1.record()
            e1.synchronize()
            t.append(e0.elapsed_time(e1) * 1e3)
        print("%s at %d maps: %.1f us (%.2f TB/s algorithmic)" % (name, n, np.mean(t), 2 * x.numel() * 2 / np.mean(t) / 1e6))


if __name__ == "__main__":
    main()
