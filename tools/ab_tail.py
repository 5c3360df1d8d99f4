#!/usr/bin/env python3
"""Paired in-process A/B of the detection tail: conv8_2 + heads as two launches against the one-launch form (csrc/conv_tail.hip), 320 maps of 256 x 256,
alternating inside ONE process so that box drift cancels.   python3 tools/ab_tail.py [maps] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from v2x_sim_amd import ops  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import FaFNet  # noqa: E402


def main():
    maps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = FaFNet(Config("test")).to(dev).eval()
    pk = m.packed(dev)
    last, heads = pk["dec"][-1], pk["heads"]
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(maps, 256, 256, 32, generator=g)).to(torch.bfloat16).to(dev)
    forms = {"two launches": lambda: ops.run_layer(heads, ops.run_layer(last, x)),
             "one launch": lambda: ops.conv2d_tail(last.halo, heads.halo, x, heads.split)}
    outs = {}
    for k, f in forms.items():
        for _ in range(2):
            outs[k] = f()
    torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs["two launches"], outs["one launch"]))
    del outs
    t = {k: [] for k in forms}
    for _ in range(reps):
        for k, f in forms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f()
            e1.record()
            e1.synchronize()
            del r
            t[k].append(e0.elapsed_time(e1) * 1e3)
    a, b = np.array(t["two launches"]), np.array(t["one launch"])
    d = (b - a) / a
    gb2, gb1 = maps * 65536 * (64 + 64 + 64 + 192) / 1e9, maps * 65536 * (64 + 192) / 1e9
    print("detection tail at %d maps: two launches %.1f us (%.2f TB/s algorithmic)   one launch %.1f us (%.2f TB/s)   paired diff %+.1f %% +- %.1f %%   bit-identical: %s"
          % (maps, a.mean(), gb2 / a.mean() * 1e3, b.mean(), gb1 / b.mean() * 1e3, 100 * d.mean(), 100 * d.std() / np.sqrt(len(d)), same))


if __name__ == "__main__":
    main()
