#!/usr/bin/env python3
"""Runs ONE streamed conv layer a few times (for rocprofv3 --pmc passes on a single kernel):
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA ... -- python3 tools/conv_layer_run.py {full|half|gru}
full: 384 -> 128 @64x64, half: (384 half-res + 128) -> 128 @64x64 (upsample + concat), gru: ConvGRU 512 -> 3x256 @32x32; 160 maps."""
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
which = sys.argv[1] if len(sys.argv) > 1 else "full"
n = 160
if which == "gru":
    w = torch.randn(768, 512, 3, 3, generator=g) * 0.02
    pc = packing.pack_gru_stream("gru", w, torch.randn(768, generator=g) * 0.1, torch.randn(768, generator=g) * 0.1, C0=256, C1=256, device=dev)
    x0 = torch.randn(n, 32, 32, 256, generator=g).to(torch.bfloat16).to(dev)
    x1 = torch.randn(n, 32, 32, 256, generator=g).to(torch.bfloat16).to(dev)
else:
    c0, c1, cout, hw, up = {"full": (384, 0, 128, 64, 0), "half": (384, 128, 128, 64, 1)}[which]
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    pc = packing.pack_conv_stream(which, w, torch.ones(cout), torch.zeros(cout), C0=c0 if c1 else c0 + c1, C1=c1, up0=up, relu=True, device=dev)
    if c1:
        x0 = torch.randn(n, hw // 2, hw // 2, c0, generator=g).to(torch.bfloat16).to(dev)
        x1 = torch.randn(n, hw, hw, c1, generator=g).to(torch.bfloat16).to(dev)
    else:
        x0 = torch.randn(n, hw, hw, c0, generator=g).to(torch.bfloat16).to(dev)
        x1 = None
for _ in range(3):
    y = ops.conv2d(pc, x0, x1)
torch.cuda.synchronize()
