#!/usr/bin/env python3
"""Latency legs of bench.py (1 / 8 / 32 collaborative frames per replay) against the split-K target of declared latency launches (tuning SPLITK_TARGET),
interleaved over `rounds` passes in one process.   python tools/latency_target_sweep.py [target ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from v2x_sim_amd import tuning  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import V2VNet  # noqa: E402
from v2x_sim_amd.utils.synthetic import init_synthetic_weights  # noqa: E402


def main(targets=(320, 400, 480, 640), rounds=3):
    dev = torch.device("cuda:0")
    model = init_synthetic_weights(V2VNet(Config("test")), seed=0).to(dev).eval()
    res = {t: {} for t in targets}
    for _ in range(rounds):
        for t in targets:
            tuning.set("SPLITK_TARGET", t)
            for mode in (True, False):
                out = bench.measure_latency(model, dev, (1, 8, 32), reps=30, small_batch=mode)
                for b in (1, 8, 32):
                    res[t].setdefault((mode, b), []).append(out["b%d_ms" % b])
    print("# ms per replay, median of %d interleaved passes (each a median of 30 replays); sharded runner SMALL_BATCH=1 | plain model (declares its own latency launches)" % rounds)
    print("%8s  %26s  |  %26s" % ("target", "b1      b8      b32", "b1      b8      b32"))
    for t in targets:
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        print("%8d  %s  |  %s" % (t, "  ".join("%6.3f" % med(res[t][(True, b)]) for b in (1, 8, 32)), "  ".join("%6.3f" % med(res[t][(False, b)]) for b in (1, 8, 32))))


if __name__ == "__main__":
    main(tuple(int(a) for a in sys.argv[1:]) or (320, 400, 480, 640))
