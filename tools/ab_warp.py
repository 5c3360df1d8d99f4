#!/usr/bin/env python3
"""Paired in-process A/B of the warp + fuse kernel forms (tuning switch WARP_LDS: 0 direct, 1 LDS-staged, 2 LDS-staged with shared set-ups) at the
bench geometry: 64 frames x 5 agents, 32 x 32 x 256 maps, mean over the four neighbours.   python3 tools/ab_warp.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from v2x_sim_amd import ops, tuning  # noqa: E402
from v2x_sim_amd._lib import V2X_FUSE_MEAN  # noqa: E402
from v2x_sim_amd.utils.synthetic import synthetic_poses  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    A, Bt = 5, 64
    g = torch.Generator().manual_seed(0)
    feat = torch.relu(torch.randn(A * Bt, 32, 32, 256, generator=g)).to(torch.bfloat16).to(dev)
    T = torch.from_numpy(synthetic_poses(Bt, A, seed=3)).to(dev)
    items = [(a, f) for a in range(A) for f in range(Bt)]
    it = ops.items_tensor(items, A, Bt, dev)
    coef = torch.ones(len(items), A)
    for m, (a, f) in enumerate(items):
        coef[m, a] = 0
    coef = coef.to(dev)
    out = torch.empty_like(feat)
    forms = (1, 2)
    res = {}
    for v in forms:
        tuning.set("WARP_LDS", v)
        for _ in range(3):
            ops.warp_fuse(feat, A, Bt, T, it, coef, V2X_FUSE_MEAN, out=out)
        res[v] = out.clone()
    torch.cuda.synchronize()
    t = {v: [] for v in forms}
    for r in range(reps):
        for v in (forms if r % 2 == 0 else forms[::-1]):
            tuning.set("WARP_LDS", v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.warp_fuse(feat, A, Bt, T, it, coef, V2X_FUSE_MEAN, out=out)
            e1.record()
            e1.synchronize()
            t[v].append(e0.elapsed_time(e1) * 1e3)
    a, b = np.array(t[1]), np.array(t[2])
    d = (b - a) / a
    print("warp + mean, 320 output maps: per-item set-ups %.1f us, shared set-ups %.1f us, paired diff %+.1f %% +- %.1f %%, outputs %s"
          % (a.mean(), b.mean(), 100 * d.mean(), 100 * d.std() / np.sqrt(len(d)), "bit-identical" if torch.equal(res[1], res[2]) else "DIFFER"))


if __name__ == "__main__":
    main()
