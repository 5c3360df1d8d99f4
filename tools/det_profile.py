#!/usr/bin/env python3
"""Times the detection tail at 320 maps, stage by stage: heads -> logits, v2x_det_postprocess, heads -> candidates (V2X_EPI_DET) at several
candidate densities, v2x_det_nms_candidates.   python3 tools/det_profile.py"""
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
from v2x_sim_amd.utils import postprocess as P  # noqa: E402
from v2x_sim_amd.utils.synthetic import init_synthetic_weights  # noqa: E402

dev = torch.device("cuda:0")
cfg = Config("test")
m = init_synthetic_weights(FaFNet(cfg), seed=0).to(dev)
pk = m.packed(dev)
N = 320
g = torch.Generator().manual_seed(0)
x = (torch.randn(N, 256, 256, 32, generator=g) * 0.7).relu().to(torch.bfloat16).to(dev)
anchors = torch.from_numpy(P.build_anchor_map(cfg).reshape(-1, 6)).to(dev)


def timed(fn, reps=10):
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


cls, loc = ops.run_layer(pk["heads"], x)
cls, loc = cls.reshape(N, -1, 2), loc.reshape(N, -1, 6)
fg = torch.softmax(cls[:4].float(), -1)[..., 1].flatten()
print("heads -> logits (EPI2 = 2)                         %8.1f us" % timed(lambda: ops.run_layer(pk["heads"], x)))
for q in (0.999, 0.9999, 1.0):
    thr = float(torch.quantile(fg[:4000000], q)) if q < 1.0 else 0.999999
    keys, codes, counts = ops.conv2d_det(pk["heads"].det, x, thr, 4096)
    print("quantile %.4f (thr %.4f, %d candidates per map):" % (q, thr, int(counts.float().mean())))
    print("   v2x_det_postprocess on the logits               %8.1f us" % timed(lambda: ops.det_postprocess(cls, loc, anchors, thr, 0.01, 4096)))
    print("   heads -> candidates (EPI2 = 3)                  %8.1f us" % timed(lambda: ops.conv2d_det(pk["heads"].det, x, thr, 4096)))
    print("   v2x_det_nms_candidates                          %8.1f us" % timed(lambda: ops.det_nms_candidates(keys, codes, counts, anchors, 0.01)))
print("conv8_2 (32 -> 32, for scale)                      %8.1f us" % timed(lambda: ops.run_layer(pk["dec"][-1], x)))
