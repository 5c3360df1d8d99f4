#!/usr/bin/env python3
"""In-process A/B of the 4-wave vs 8-wave streamed-weights kernels on the default-batch layer shapes
(160 maps).  usage: python tools/ab_stream_waves.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "v2x-sim_amd"))
from v2x_sim_amd import ops, packing, tuning  # noqa: E402

dev = torch.device("cuda:0")
N = 160
SHAPES = [("conv3_2", 0, 256, 256, 32, False), ("conv5_1", 512, 256, 256, 32, False), ("conv5_2", 0, 256, 256, 32, False),
          ("conv6_1", 256, 128, 128, 64, False), ("conv6_2", 0, 128, 128, 64, False), ("conv2_2", 0, 128, 128, 64, False),
          ("gru", 256, 256, 256, 32, True)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, cup, c, cout, hw, gru in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, hw, hw, c, generator=g).to(torch.bfloat16).to(dev)
    if gru:
        x0 = torch.randn(N, hw, hw, cup, generator=g).to(torch.bfloat16).to(dev)
        pc = packing.pack_gru_stream("g", torch.randn(3 * cout, cup + c, 3, 3, generator=g) * 0.02,
                                     torch.zeros(3 * cout), torch.zeros(3 * cout), C0=cup, C1=c, device=dev)
        fn = lambda: ops.conv2d(pc, x0, x)
        flops = 2.0 * N * hw * hw * 3 * cout * (cup + c) * 9
    else:
        w = torch.randn(cout, cup + c, 3, 3, generator=g) * 0.02
        pc = packing.pack_conv_stream("t", w, torch.ones(cout), torch.zeros(cout), C0=cup if cup else c,
                                      C1=c if cup else 0, up0=1 if cup else 0, device=dev)
        if cup:
            x0 = torch.randn(N, hw // 2, hw // 2, cup, generator=g).to(torch.bfloat16).to(dev)
            fn = lambda: ops.conv2d(pc, x0, x)
        else:
            fn = lambda: ops.conv2d(pc, x)
        flops = 2.0 * N * hw * hw * cout * (cup + c) * 9
    res = {}
    for rnd in range(2):
        for waves in ("4", "8", "8p"):
            os.environ.pop("V2X_STREAM_WAVES", None)
            os.environ.pop("V2X_STREAM_PERSIST", None)
            if waves == "4":
                tuning.set("STREAM_WAVES", int("4"))
            elif waves == "8":
                tuning.set("STREAM_PERSIST", int("0"))
            res.setdefault(waves, []).append(timed(fn))
    t4, t8, t8p = min(res["4"]), min(res["8"]), min(res["8p"])
    print("%-8s 4-wave %7.1f us (%6.0f TF/s)   8-wave %7.1f us (%6.0f TF/s)   8-wave persistent %7.1f us (%6.0f TF/s)  x%.3f" % (
        name, t4, flops / t4 / 1e6, t8, flops / t8 / 1e6, t8p, flops / t8p / 1e6, t8 / t8p), flush=True)
