#!/usr/bin/env python3
"""PCIe-inclusive throughput (row f-2): V2VNet points -> logits when the sweeps start in (pinned) HOST memory, as they do behind a
DataLoader.  Three arms on the same weights and the same step (64 frames x 5 agents per launch, eager):
  resident  -- sweeps already in HBM (what bench.py times),
  inline    -- `points.to(device)` inside the step (upstream's loop),
  prefetch  -- datasets.DevicePrefetcher: copies on their own stream, `depth` batches ahead.
usage: python tools/stream_points.py [frames_per_step] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.datasets import DevicePrefetcher  # noqa: E402
from v2x_sim_amd.models.det import V2VNet  # noqa: E402
from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet  # noqa: E402
from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses  # noqa: E402


def main(Bt=64, steps=12, keep_outputs=False):
    dev = torch.device("cuda:0")
    model = init_synthetic_weights(V2VNet(Config("test")), seed=0).to(dev)
    shard = AgentShard(5, Bt, 0, 1)
    runner = ShardedV2VNet(model, shard)
    n_pts = torch.full((5 * Bt,), 65536, dtype=torch.int32, device=dev)
    trans = torch.from_numpy(synthetic_poses(Bt, 5, seed=2)).to(dev)
    plan = shard.fusion_plan(torch.full((Bt, 5), 5), dev)
    host = [torch.from_numpy(synthetic_points(5 * Bt, 65536, seed=s)).pin_memory() for s in range(3)]   # a pinned ring
    mb = host[0].numel() * 4 / 1e6

    def step(points):
        with torch.no_grad():
            return runner.forward_points(points, n_pts, trans, plan)

    resident = host[0].to(dev)
    for _ in range(3):
        out = step(resident)
    torch.cuda.synchronize()
    res = {"resident": 0.0, "inline": 0.0, "prefetch": 0.0}
    for _rep in range(2):   # best of two passes per arm: 8 eager steps are short enough for one host hiccup to halve an arm
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step(resident)
        torch.cuda.synchronize()
        res["resident"] = max(res["resident"], Bt * steps / (time.perf_counter() - t0))
        t0 = time.perf_counter()
        for s in range(steps):
            out = step(host[s % 3].to(dev, non_blocking=True))
        torch.cuda.synchronize()
        res["inline"] = max(res["inline"], Bt * steps / (time.perf_counter() - t0))
        t0 = time.perf_counter()
        for batch in DevicePrefetcher(({"points": host[s % 3]} for s in range(steps)), dev, depth=2):
            out = step(batch["points"])
        torch.cuda.synchronize()
        res["prefetch"] = max(res["prefetch"], Bt * steps / (time.perf_counter() - t0))
    print("%d frames/step, %.0f MB of sweeps per step: resident %.0f frames/s, inline copy %.0f, prefetched %.0f (%.0f %% of resident)"
          % (Bt, mb, res["resident"], res["inline"], res["prefetch"], 100 * res["prefetch"] / res["resident"]), file=sys.stderr)
    if keep_outputs:
        # the last step of every arm read ring slot (steps - 1) % 3: its logits must equal the resident run on that slot's sweeps
        outs = {"inline": None, "prefetch": out, "resident_same_ring_slot": step(host[(steps - 1) % 3].to(dev))}
        outs["inline"] = step(host[(steps - 1) % 3].to(dev, non_blocking=True))
        torch.cuda.synchronize()
        return res, outs
    return res, out


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:3]))
