#!/usr/bin/env python3
"""One BASELINE.json config, eager launches, for the rocprofv3 passes of `tools/profile_round.sh --config <name> <tag>`.

    python3 tools/config_run.py --config when2com [--frames 64] [--reps 5] [--layers out.json]

--layers: instead of the plain repetition loop, one instrumented pass (HIP events around every launch, v2x_sim_amd.ops.PROFILE) written as JSON:
per kernel name the launches, the algorithmic FLOPs and bytes per launch (the figures DESIGN.md section 6 prices against) and the event time."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def main():
    import bench_configs as bc
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=bc.CONFIG_NAMES)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layers", default=None)
    args = ap.parse_args()
    from v2x_sim_amd import ops
    cs = bc.ConfigSet(args.frames)
    fn = cs.build(args.config)
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        if args.layers:
            agg, layers = {}, {}
            for _ in range(args.reps):
                ops.PROFILE = []
                fn()
                torch.cuda.synchronize()
                recs, ops.PROFILE = ops.PROFILE, None
                for name, fl, by, e0, e1, layer in recs:
                    a = agg.setdefault(name, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})
                    a["launches"] += 1
                    a["flops"] += fl
                    a["bytes"] += by
                    a["ms"] += e0.elapsed_time(e1)
                    la = layers.setdefault(layer, {"kernel": name, "launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})
                    la["launches"] += 1
                    la["flops"] += fl
                    la["bytes"] += by
                    la["ms"] += e0.elapsed_time(e1)
            for d in (agg, layers):
                for a in d.values():
                    n = a["launches"]
                    a["alg_flops_per_launch"], a["alg_bytes_per_launch"], a["event_us_per_launch"] = a.pop("flops") / n, a.pop("bytes") / n, a.pop("ms") * 1e3 / n
                    a["launches_per_pass"] = n // args.reps
                    del a["launches"]
            json.dump({"config": args.config, "frames": args.frames, "maps_per_launch": 5 * args.frames, "kernels": agg, "layers": layers},
                      open(args.layers, "w"), indent=1)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print(json.dumps({"config": args.config, "frames": args.frames, "ms_per_pass": ms, "frames_per_s": args.frames / ms * 1e3}))


if __name__ == "__main__":
    main()
