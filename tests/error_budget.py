#!/usr/bin/env python3
"""Where the end-to-end tolerance comes from: a DERIVED error budget for bf16 storage through the whole network, printed next to the measured
errors.  CPU only.  Lives under tests/ because it drives the oracle (oracle/coperception_ref.py is test infrastructure: only tests/, smoke() and bench.py's baseline legs may import it).

Method (first-order propagation through the REAL network, not a formula): the bf16-emulating oracle rounds at known places -- `_q(...)`: a layer's input (a no-op
when the producer already rounded it), its weights, its output.  With every rounding OFF the graph is the fp32 oracle; with ALL of them on it is the emulating
oracle, the comparator the HIP kernels match to 1 bf16 ulp per stage.  Here each rounding SITE n is switched on ALONE, the network is run, and the error e_n it
leaves in the logits is measured (rms and max, relative to max|ref| as tests/test_gpu_models.py normalises).  Rounding errors of different sites are independent to
first order, so the budget is

    rms_total  = sqrt(sum_n rms_n^2)                                    (quadrature)
    max_total ~= rms_total * sqrt(2 ln N) * k,  N = number of logits     (Gaussian extreme value; k = the kurtosis allowance measured on the full emulation)

and the tests' bounds (3e-2 max, 3e-3 mean of max|ref|) can be read against it.  Also printed: the error of the full emulation (all sites on) against fp32, which
is what any correct bf16 pipeline -- the HIP path included -- shows up to a re-draw of the rounding noise (two correct pipelines with different fp32 summation orders
decorrelate to exactly this level: tests/test_gpu_models.py's header).

    python3 tests/error_budget.py [--model v2vnet|fafnet] [--seed 0] [--measured profiles/r06_e2e_errors_gpu.txt]
"""
import argparse
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import coperception_ref as R  # noqa: E402
from oracle import voxelize_ref as VR  # noqa: E402


class Sites:
    """Replaces R._q: counts the rounding sites of one forward pass and rounds only at the enabled one(s)."""

    def __init__(self):
        self.n = 0
        self.enabled = None      # None: all; set of ints: only those
        self.shapes = []
        self.effective = []

    def __call__(self, x, emulate):
        if not emulate:
            return x
        i = self.n
        self.n += 1
        if len(self.shapes) <= i:
            self.shapes.append(tuple(x.shape))
            self.effective.append(False)
        if self.enabled is None or i in self.enabled:
            r = x.to(torch.bfloat16).to(torch.float32)
            if self.enabled is None:        # the full emulation: does THIS site change anything there?  (a layer's input is its producer's output, rounded already)
                self.effective[i] = bool((r != x).any())
            return r
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="v2vnet", choices=("v2vnet", "fafnet"))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--measured", default=None, help="a pytest -s log of tests/test_gpu_models.py (HIP vs oracle); its worst lines are quoted at the end")
    args = ap.parse_args()
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet, V2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    A = 5 if args.model == "v2vnet" else 2
    pm = init_synthetic_weights((V2VNet if args.model == "v2vnet" else FaFNet)(Config("train")), seed=args.seed)
    om = (R.V2VNet if args.model == "v2vnet" else R.FaFNet)().eval()
    om.load_state_dict(pm.state_dict())
    pts = synthetic_points(A, 20000, seed=1)
    bev = torch.from_numpy(np.stack([VR.voxelize_occupy(p) for p in pts])[:, None])
    T = torch.from_numpy(synthetic_poses(1, A, seed=2))
    nat = torch.full((1, A), A)

    def run():
        with torch.no_grad():
            return om(bev, T, nat, batch_size=1) if args.model == "v2vnet" else om(bev)

    sites = Sites()
    R._q = sites
    om.emulate_bf16 = False
    ref = run()
    scale = {k: float(ref[k].abs().max()) for k in ("cls", "loc")}
    om.emulate_bf16 = True
    sites.n, sites.enabled = 0, None
    full = run()
    n_sites = sites.n
    print("# error budget of bf16 storage, %s (seed %d, %d maps of 256 x 256): %d rounding sites per forward pass" % (args.model, args.seed, A, n_sites))
    print("# errors relative to max|ref| (cls %.3f, loc %.3f), as tests/test_gpu_models.py::check normalises" % (scale["cls"], scale["loc"]))
    rows = []
    for n in range(n_sites):
        if not sites.effective[n]:
            continue
        sites.n, sites.enabled = 0, {n}
        out = run()
        e = {}
        for k in ("cls", "loc"):
            d = (out[k] - ref[k]).abs() / scale[k]
            e[k] = (float(d.pow(2).mean().sqrt()), float(d.max()), float(d.mean()))
        rows.append((n, sites.shapes[n], e))
    print("%4s %-26s %11s %11s %11s %11s" % ("site", "tensor rounded", "cls rms", "cls max", "loc rms", "loc max"))
    for n, shp, e in rows:
        if max(e["cls"][1], e["loc"][1]) > 0:
            print("%4d %-26s %11.3e %11.3e %11.3e %11.3e" % (n, "x".join(str(s) for s in shp), e["cls"][0], e["cls"][1], e["loc"][0], e["loc"][1]))
    print("# (%d of the %d sites are no-ops in the full emulation and are left out: the input of a layer whose producer already rounded that tensor, and exact {0, 1} inputs)"
          % (n_sites - len(rows), n_sites))
    print()
    for k in ("cls", "loc"):
        rms_q = math.sqrt(sum(e[k][0] ** 2 for _, _, e in rows))
        d = (full[k] - ref[k]).abs() / scale[k]
        rms_f, max_f, mean_f = float(d.pow(2).mean().sqrt()), float(d.max()), float(d.mean())
        N = d.numel()
        gauss = math.sqrt(2.0 * math.log(N))
        print("%s: DERIVED rms (quadrature of the sites alone) %.3e   |   full emulation vs fp32: rms %.3e  mean %.3e  max %.3e" % (k, rms_q, rms_f, mean_f, max_f))
        print("%s: Gaussian extreme of N = %d logits = %.2f sigma -> derived max %.3e; measured max / measured rms = %.2f sigma (heavier tails: errors scale with the local"
              " activation magnitude)" % (k, N, gauss, rms_q * gauss, max_f / max(rms_f, 1e-30)))
        # a second, independent bf16 pipeline (different fp32 summation order = a re-draw of the flips) sits sqrt(2) further from this one than either from fp32 at most
        print("%s: two correct bf16 pipelines decorrelate to <= sqrt(2) x that: derived pair bound rms %.3e, max ~%.3e  -- the asserted bounds are mean 3e-3, max 3e-2"
              % (k, rms_q * math.sqrt(2.0), max_f * math.sqrt(2.0)))
    if args.measured and os.path.exists(args.measured):
        worst = {}
        for line in open(args.measured):
            m = re.match(r"\.?(.*?)\s+max (\S+)\s+mean (\S+)\s+\(rel", line)
            if m:
                key = "cls" if " cls " in line else "loc" if " loc " in line else "other"
                mx, mean = float(m.group(2)), float(m.group(3))
                w = worst.setdefault(key, [0.0, "", 0.0, ""])
                if mx > w[0]:
                    w[0], w[1] = mx, m.group(1).strip()
                if mean > w[2]:
                    w[2], w[3] = mean, m.group(1).strip()
        print()
        print("# MEASURED on the MI355X, HIP path vs the oracle (%s): worst over every model / fusion / precision of the oracle" % os.path.basename(args.measured))
        for k, w in worst.items():
            print("%s: worst max %.3e (%s), worst mean %.3e (%s)" % (k, w[0], w[1], w[2], w[3]))


if __name__ == "__main__":
    main()
