#!/usr/bin/env python3
"""Headline benchmark: BEV frames/sec of V2VNet 5-agent detection on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one pass of the whole hot path over one batch of synthetic frames, inputs already
resident in HBM:  LiDAR points -> voxel scatter (a1) -> encoder (a2) -> [RCCL all-gather of the
fusion-layer maps when N > 1] -> warp + ConvGRU (a3, a4) -> decoder (a6) -> det heads (a7).
Work items are (agent, frame) maps sharded agent-major over the ranks (v2x_sim_amd/parallel.py);
per-GPU work is fixed as N grows (weak scaling): frames = frames_per_gpu * N, run as two half-batches whose
all-gathers are asynchronous and hidden under the other half's compute.

Prints ONE JSON line (rank 0).  `roofline` is computed from HIP events recorded live around every
kernel launch of an instrumented pass on the launch stream; `cpu_baseline` times the CPU oracle
(oracle/, PyTorch-CPU fp32) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s spec, bf16 MFMA ~2.5 PF dense
PEAK_HBM_GBS = 8000.0
PEAK_MFMA_TFLOPS = 2500.0
AGENTS = 5
POINTS_PER_SWEEP = 65536


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-gpu", type=int, default=128,
                    help="frames per step and GPU (two half-batches of 64 = 320 maps: every conv layer then splits into a "
                         "whole number of rounds of 256 workgroups; 64 frames/GPU leaves the 32x32 layers at 2.5 rounds, -4 %%)")
    ap.add_argument("--gnn-iters", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph (N=1 only)")
    return ap.parse_args()


def cpu_baseline(model_state, gnn_iters, budget_s=20.0):
    """Oracle V2VNet (PyTorch-CPU fp32 + numpy voxelizer) on whole 5-agent frames, same synthetic
    generator; at least 1 timed frame, stops after ~budget_s."""
    from oracle import coperception_ref as R
    from oracle import voxelize_ref as VR
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    # measured on the MI355X host (256 hardware threads): 8 thr 0.44, 16 thr 0.40, 32 thr 0.40,
    # 64 thr 0.79, 256 thr 45 s/frame -- oneDNN oversubscribes on this small batch, so cap at 32.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    om = R.V2VNet(gnn_iter_times=gnn_iters).eval()
    om.load_state_dict(model_state)
    pts = synthetic_points(AGENTS, POINTS_PER_SWEEP, seed=1234)
    T = torch.from_numpy(synthetic_poses(1, AGENTS, seed=99))
    nat = torch.full((1, AGENTS), AGENTS)

    def frame():
        bev = np.stack([VR.voxelize_occupy(p) for p in pts])[:, None]
        with torch.no_grad():
            om(torch.from_numpy(bev), T, nat, batch_size=1)

    frame()  # warm-up (oneDNN primitive setup)
    n, t0 = 0, time.perf_counter()
    while True:
        frame()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40:
            break
    return {"value": n / el, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d whole 5-agent V2VNet frames (65536 pts/agent, 256x256x13 BEV), oracle fp32 PyTorch-CPU, "
                      "%d threads, after 1 warm-up frame" % (n, torch.get_num_threads())}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # V2X_FORCE_DIST=1: exercise the RCCL code path (process group, bf16 all-gather, barrier, all-reduce) even with
    # one rank -- the only way to smoke-test it on a 1-GPU box
    force_dist = os.environ.get("V2X_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses

    if args.frames_per_gpu % 2:
        raise SystemExit("--frames-per-gpu must be even (the step runs as two half-batches)")
    Bt = args.frames_per_gpu * world            # frames per step, whole job
    Bh = Bt // 2                                # frames per half-batch
    model = init_synthetic_weights(V2VNet(Config("test"), gnn_iter_times=args.gnn_iters, num_agent=AGENTS), seed=0)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    # The step is two independent half-batches of Bh frames, each agent-sharded over all ranks.  Half A's all-gather
    # is started asynchronously and flies under half B's encoder; half B's flies under half A's fusion/decoder/heads.
    # The decomposition is the same for every N (at N = 1 there is simply nothing to gather): weak scaling.
    shard = AgentShard(AGENTS, Bh, rank, world)
    runner = ShardedV2VNet(model, shard)
    if force_dist and world == 1:
        class _ForcedWorld1(ShardedV2VNet):     # take the world > 1 code path (async RCCL all-gather) on one rank
            def begin(self, points, n_pts):
                pk = self.model.packed(points.device)
                feats = self.encode_points(points, n_pts, pk)
                local = feats[self.model.layer].contiguous()
                out = torch.empty_like(local)
                return feats, out, dist.all_gather_into_tensor(out, local, async_op=True)
        runner = _ForcedWorld1(model, shard)
    halves = []
    for h in range(2):
        # synthetic sweeps of this rank's (agent, frame) items of half h, resident in HBM
        pts = np.concatenate([synthetic_points(1, POINTS_PER_SWEEP, seed=1000 + 100000 * h + r) for r in shard.rows])
        halves.append({"points": torch.from_numpy(pts).to(dev),
                       "n_pts": torch.full((shard.per_rank,), POINTS_PER_SWEEP, dtype=torch.int32, device=dev),
                       "trans": torch.from_numpy(synthetic_poses(Bh, AGENTS, seed=99 + h)).to(dev),
                       "plan": shard.fusion_plan(torch.full((Bh, AGENTS), AGENTS), dev)})
    model.packed(dev)

    def step():
        with torch.no_grad():
            a, b = halves
            fa = runner.begin(a["points"], a["n_pts"])
            fb = runner.begin(b["points"], b["n_pts"])
            out_a = runner.finish(*fa, a["trans"], a["plan"])
            out_b = runner.finish(*fb, b["trans"], b["plan"])
            return out_a, out_b

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(max(args.warmup, 1) if args.graph and not use_dist else args.warmup):
        out = step()
    torch.cuda.synchronize()

    use_graph = bool(args.graph) and not use_dist
    if use_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        run = g.replay
        for _ in range(2):
            run()
    else:
        run = step
    torch.cuda.synchronize()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    fps = Bt * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        # instrumented pass: HIP events around every launch, on the launch stream (eager, not the graph)
        ops.PROFILE = []
        n_inst = 3
        for _ in range(n_inst):
            step()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        groups = {}
        for name, fl, by, e0, e1, _layer in recs:
            gdict = groups.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            gdict["ms"] += e0.elapsed_time(e1)
            gdict["flops"] += fl
            gdict["bytes"] += by
            gdict["launches"] += 1
        total_ms = sum(v["ms"] for v in groups.values())
        dom = max(groups, key=lambda k: groups[k]["ms"])
        d = groups[dom]
        ai = d["flops"] / max(d["bytes"], 1.0)
        if ai >= PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_MFMA_TFLOPS, "traffic": None}
        else:
            achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBS, "traffic": None}
        # HBM traffic per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, gfx950-corrected by tools/pmc_traffic.py); valid for the default workload (128 frames/GPU) only
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tfile) and args.frames_per_gpu == 128 and world == 1:  # 2 half-batches of 64 = the profiled launches
            with open(tfile) as fh:
                tk = json.load(fh)["kernels"].get(dom)
            if tk:
                roofline["traffic"] = tk["hbm_bytes_per_launch"]
        roofline.update({"kernel": dom, "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                         "launches_per_step": d["launches"] // n_inst,
                         "share_of_kernel_time": d["ms"] / total_ms,
                         "alg_flops_per_launch": d["flops"] / d["launches"],
                         "alg_bytes_per_launch": d["bytes"] / d["launches"]})
        kernels = {k: {"us_per_step": v["ms"] * 1e3 / n_inst, "launches_per_step": v["launches"] // n_inst,
                       "tflops": v["flops"] / max(v["ms"], 1e-9) / 1e9, "gbs": v["bytes"] / max(v["ms"], 1e-9) / 1e6}
                   for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}
    else:
        kernels = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(state, args.gnn_iters)

    if rank == 0:
        rec = {
            "metric": "BEV frames/sec, V2VNet 5-agent detection (256x256 BEV)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded 65536-pt sweeps per agent, random SE(2) poses, He-init weights)",
            "config": {"workload": "V2VNet 5-agent detection, points->logits (a1-a7), gnn_iter=%d" % args.gnn_iters,
                       "agents": AGENTS, "frames_per_step": Bt, "frames_per_gpu": args.frames_per_gpu,
                       "half_batches": 2,
                       "points_per_agent": POINTS_PER_SWEEP, "bev": [256, 256, 13],
                       "sharding": "agent-major (agent,frame) items, contiguous slices; async RCCL all-gather of the fusion "
                                   "maps of one half-batch overlapped with the other half's compute"
                                   if world > 1 else "single GPU, no collective",
                       "hip_graph": use_graph},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
    if use_dist:
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which (on a pipe) is only flushed at exit and would land
    # AFTER our JSON line: flush C stdio first so that the JSON record is the last line of stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
