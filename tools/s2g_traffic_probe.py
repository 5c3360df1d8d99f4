#!/usr/bin/env python3
"""Runs ONE stride-2 streamed layer of the encoder at the bench extent (320 maps), a few launches, for per-LAYER counter passes (which of the two
layers that share conv3x3_s2g_kernel<8, 32> over-fetches, against its algorithmic bytes):
    rocprofv3 --pmc FETCH_SIZE -d out -o p --output-format csv -- python3 tools/s2g_traffic_probe.py {conv2_1|conv3_1|conv4_1} [xcd 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from v2x_sim_amd import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "conv2_1"
cin, cout, hw = {"conv2_1": (64, 128, 128), "conv3_1": (128, 256, 64), "conv4_1": (256, 512, 32)}[which]
n = 320
w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
pc = packing.pack_conv_stream(which, w, torch.ones(cout), torch.zeros(cout), stride=2, relu=True, device=dev)
x = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)
for _ in range(5):
    y = ops.conv2d(pc, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = ops.conv2d(pc, x)
e1.record()
torch.cuda.synchronize()
alg = x.numel() * 2 + y.numel() * 2 + w.numel() * 2
print("%s: %s  %.1f us per launch; algorithmic bytes %.1f MB (in %.1f + out %.1f)" % (which, ops.conv_kernel_name(pc, hw, hw, N=n), e0.elapsed_time(e1) * 100.0, alg / 1e6,
                                                                                         x.numel() * 2 / 1e6, y.numel() * 2 / 1e6))
