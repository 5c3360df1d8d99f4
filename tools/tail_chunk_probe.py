#!/usr/bin/env python3
"""Does running the decoder tail (conv8_1 -> conv8_2 -> heads, 11 GB of HBM traffic per 320 maps, 3.5 ms) in CHUNKS of maps keep the two
32-channel intermediates (4.2 MB per map each) in the 256-MB Infinity Cache between the launches?  Times the three layers over 320 maps in
one go against chunks of 160 / 64 / 32 / 16 maps (same kernels, same bits: a map's tile does not know its batch).
usage: python tools/tail_chunk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
from v2x_sim_amd import ops  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import V2VNet  # noqa: E402
from v2x_sim_amd.utils.synthetic import init_synthetic_weights  # noqa: E402

dev = torch.device("cuda:0")
model = init_synthetic_weights(V2VNet(Config("test")), seed=0).to(dev)
pk = model.packed(dev) if callable(getattr(model, "packed", None)) else model.packed
dec, heads = pk["dec"], pk["heads"]
c81, c82 = dec[-2], dec[-1]
N = 320
g = torch.Generator().manual_seed(0)
y7 = (torch.randn(N, 128, 128, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
x0 = (torch.randn(N, 256, 256, 32, generator=g) * 0.5).to(torch.bfloat16).to(dev)


def tail(lo, hi):
    a = ops.run_layer(c81, y7[lo:hi], x0[lo:hi])
    b = ops.run_layer(c82, a)
    return ops.run_layer(heads, b)


def run(chunk):
    outs = []
    for lo in range(0, N, chunk):
        outs.append(tail(lo, min(N, lo + chunk)))
    return outs


ref = run(N)
for chunk in (320, 160, 64, 32, 16, 320):
    for _ in range(2):
        outs = run(chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        outs = run(chunk)
    e1.record()
    torch.cuda.synchronize()
    same = torch.equal(torch.cat([o[0] for o in outs]), ref[0][0]) and torch.equal(torch.cat([o[1] for o in outs]), ref[0][1])
    del outs
    print("chunks of %3d maps: %.3f ms per 320 maps (conv8_1 + conv8_2 + heads), outputs %s" % (chunk, e0.elapsed_time(e1) / 5, "bit-identical" if same else "DIFFER"), flush=True)
