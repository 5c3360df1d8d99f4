#!/usr/bin/env python3
"""Headline benchmark: BEV frames/sec of V2VNet 5-agent detection on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher environment: this process starts `python -m torch.distributed.run --nproc-per-node N`
      on itself BEFORE any GPU call and relays rank 0's JSON line and the exit code; under torchrun (WORLD_SIZE set)
      it is a rank.  `--dry-run` walks launcher + rendezvous + shard plan + collective on gloo without a GPU.

One step = one pass of the whole hot path over one batch of synthetic frames, inputs already
resident in HBM:  LiDAR points -> voxel scatter (a1) -> encoder (a2) -> [RCCL all-gather of the
fusion-layer maps when N > 1] -> warp + ConvGRU (a3, a4) -> decoder (a6) -> det heads (a7).
Work items are (agent, frame) maps sharded agent-major over the ranks (v2x_sim_amd/parallel.py);
per-GPU work is fixed as N grows (weak scaling): frames = frames_per_gpu * N, run as two half-batches whose
all-gathers are asynchronous and hidden under the other half's compute.

Execution mode is the SAME for every N (`exec_mode` in the record): the step's four collective-free segments
(encoder A, encoder B, fusion+decoder+heads A, ... B) are replayed from hipGraphs and the exchange runs between them
(at N = 1 there is nothing to exchange), so the 1 -> 8 curve compares equals.  The two half-batches run on two streams
(own graph memory pools) and are joined at the end of the step: one half's kernels fill the tails and the HBM-bound
phases of the other (+2.3...3.0 % at N = 1 against `--graph 4`, the one-stream order).

Prints ONE JSON line (rank 0).  `roofline` is computed from HIP events recorded live around every
kernel launch of an instrumented pass on the launch stream; `cpu_baseline` times the CPU oracle
(oracle/, PyTorch-CPU fp32) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
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
# committed PMC traffic summaries, newest first (tools/profile_round.sh -> tools/pmc_traffic.py)
TRAFFIC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
# algorithmic FLOPs of one 5-agent frame, points -> logits: encoder + decoder + heads, + one ConvGRU pass per GNN round
# (h0 = 0: W_hh is never multiplied and not counted)
GFLOP_PER_FRAME_BASE, GFLOP_PER_GNN_ROUND = 155.8, 36.2


def parse(argv=None):
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
    ap.add_argument("--no-calibration", action="store_true", help="skip the streaming / MFMA calibration probes (the `calibration` sub-record)")
    ap.add_argument("--graph", type=int, default=1,
                    help="1: four hipGraph segments with the exchange between them, the two half-batches on two streams (every N); "
                         "4: the same on one stream; 0: eager launches; 2: the whole step as ONE hipGraph (N = 1 only, for comparison)")
    ap.add_argument("--transport", choices=("allgather", "needed"), default="allgather",
                    help="fusion-map exchange: RCCL all-gather (default) or grouped point-to-point of the needed rows only")
    ap.add_argument("--layout", choices=("spread", "agent-per-gpu"), default="spread",
                    help="spread (default): the agent-major (agent, frame) items in equal contiguous slices over ALL N ranks; "
                         "agent-per-gpu: the north_star's literal layout -- rank a < 5 owns agent a's frames, ranks >= 5 idle "
                         "(needs N >= 5; frames = frames_per_gpu * 5)")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency and other-config sub-records")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launcher, rendezvous (gloo), shard plan, collective, JSON relay")
    return ap.parse_args(argv)


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N > 1 run started as plain `python bench.py --gpus N`: starts the ranks as CHILD processes through
    torch.distributed.run (one per GPU, RCCL over xGMI; 127.0.0.1 rendezvous) and relays their output.  Nothing here
    touches the GPU -- a process that has initialised HIP must never be re-exec'd or forked into ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    record = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            record = line.rstrip("\n")          # printed last, after everything else the ranks wrote
        else:
            sys.stdout.write(line)
    rc = proc.wait()
    sys.stdout.flush()
    if record is not None:
        print(record, flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks exited 0 without a JSON record", file=sys.stderr)
    return rc


def shard_layout(args, world, rank):
    """-> (shard_world, shard_rank or None for an idle rank, process group of the shard or None = the default group).
    `agent-per-gpu` runs the 5 agents on ranks 0..4 of the job (their own sub-group, created collectively by ALL ranks); the other
    ranks only take part in the job-wide barriers and the max-over-ranks timing."""
    if args.layout == "spread":
        return world, rank, None
    if world < AGENTS:
        raise SystemExit("--layout agent-per-gpu needs --gpus >= %d (one GPU per agent)" % AGENTS)
    group = dist.new_group(ranks=list(range(AGENTS))) if world > AGENTS else None
    return AGENTS, (rank if rank < AGENTS else None), group


def dry_run(args, world, rank):
    """CPU walk through everything around the kernels: rendezvous, agent-major partition, fusion plan, the exchange
    calls (gloo instead of RCCL), barrier + max-over-ranks timing, one JSON line from rank 0."""
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sworld, srank, group = shard_layout(args, world, rank)
    Bh = args.frames_per_gpu * sworld // 2
    ok = torch.tensor([1])
    t0 = time.perf_counter()
    per_rank = AGENTS * Bh // sworld
    if srank is not None:
        shard = AgentShard(AGENTS, Bh, srank, sworld)

        class _NoModel:
            gnn_iter_num, neighbor_source, layer = 1, "initial", 3
        runner = ShardedV2VNet(_NoModel(), shard, group=group, transport=args.transport)
        plan = shard.fusion_plan(torch.full((Bh, AGENTS), AGENTS), "cpu")
        local = torch.stack([torch.full((2, 2, 8), float(r)) for r in shard.rows])  # fp32: row ids beyond 256 stay exact
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if srank is not None:
        for _ in range(args.steps):
            gathered, work = runner.start_exchange(local)
            runner.wait(work)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if srank is not None:
        # every map an owned ego reads (all agents of its frame) must sit at its agent-major row
        ok = torch.tensor([int(all(float(gathered[j * Bh + f, 0, 0, 0]) == j * Bh + f
                                   for _, f in plan["items"].tolist() for j in range(AGENTS)))])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "BEV frames/sec, V2VNet 5-agent detection (256x256 BEV)", "value": None, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "ranks_seen": world,
                          "exchange_ok": bool(int(ok)), "transport": args.transport, "layout": args.layout, "active_ranks": sworld,
                          "frames_per_step": 2 * Bh, "items_per_rank": per_rank, "elapsed_s": float(t)}), flush=True)
    return 0 if int(ok) else 1


def measure_latency(model, dev, frames_list=(1, 8, 32), reps=30, small_batch=True):
    """Latency mode (SURVEY.md 8d batch sizes): ONE hipGraph replay of points -> logits for `frames` collaborative frames,
    host-synchronised per replay, median over `reps`.  small_batch: with the tuning switch SMALL_BATCH = 1 (split-K for the streamed layers
    whose launch has fewer tiles than CUs; v2x_sim_amd/ops.py::small_batch_splitk) -- what a caller serving single frames turns on."""
    from v2x_sim_amd import tuning
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    prev = tuning.set("SMALL_BATCH", 1 if small_batch else 0)
    try:
        return _measure_latency(model, dev, frames_list, reps, small_batch)
    finally:
        tuning.set("SMALL_BATCH", prev)


def _measure_latency(model, dev, frames_list, reps, small_batch):
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    out = {}
    for frames in frames_list:
        sh = AgentShard(AGENTS, frames, 0, 1)
        rn = ShardedV2VNet(model, sh)
        pts = torch.from_numpy(np.concatenate([synthetic_points(1, POINTS_PER_SWEEP, seed=5000 + r) for r in sh.rows])).to(dev)
        n_pts = torch.full((sh.per_rank,), POINTS_PER_SWEEP, dtype=torch.int32, device=dev)
        trans = torch.from_numpy(synthetic_poses(frames, AGENTS, seed=7)).to(dev)
        plan = sh.fusion_plan(torch.full((frames, AGENTS), AGENTS), dev)
        with torch.no_grad():
            for _ in range(2):
                rn.forward_points(pts, n_pts, trans, plan)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                res = rn.forward_points(pts, n_pts, trans, plan)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out["b%d_ms" % frames] = ts[len(ts) // 2]
        out["b%d_frames_per_s" % frames] = frames / ts[len(ts) // 2] * 1e3
        del g, res
    out["mode"] = "one hipGraph replay of points->logits per batch, host-synchronised, median of %d; %s" % (
        reps, "SMALL_BATCH = 1 (split-K for launches with fewer tiles than CUs)" if small_batch else "default kernel dispatch (the throughput step's)")
    return out


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if args.frames_per_gpu % 2:
        raise SystemExit("--frames-per-gpu must be even (the step runs as two half-batches)")
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))
    if not torch.cuda.is_available():
        print("bench.py needs the MI355X: the product path has no CPU fallback (use --dry-run for the launcher walk)",
              file=sys.stderr)
        sys.exit(3)
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

    # layout of the (agent, frame) items over the ranks: all of them (spread) or one agent per rank on ranks 0..4 (agent-per-gpu)
    sworld, srank, sgroup = shard_layout(args, world, rank) if use_dist else (1, 0, None)
    active = srank is not None
    Bt = args.frames_per_gpu * sworld           # frames per step, whole job (per-GPU work fixed: weak scaling)
    Bh = Bt // 2                                # frames per half-batch
    model = init_synthetic_weights(V2VNet(Config("test"), gnn_iter_times=args.gnn_iters, num_agent=AGENTS), seed=0)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    L = model.layer
    # The step is two independent half-batches of Bh frames, each agent-sharded over all ranks.  Half A's exchange
    # is started asynchronously and flies under half B's encoder; half B's flies under half A's fusion/decoder/heads.
    # The decomposition is the same for every N (at N = 1 there is simply nothing to gather): weak scaling.
    shard = AgentShard(AGENTS, Bh, srank if active else 0, sworld)      # (an idle rank builds rank 0's tables and never launches)
    runner = ShardedV2VNet(model, shard, group=sgroup, transport=args.transport)
    if force_dist and world == 1:
        class _ForcedWorld1(ShardedV2VNet):     # take the world > 1 code path (async RCCL all-gather) on one rank
            def start_exchange(self, local, out=None, counts=None):
                local = local.contiguous()
                out = torch.empty_like(local) if out is None else out
                return out, dist.all_gather_into_tensor(out, local, async_op=True)
        runner = _ForcedWorld1(model, shard)
    halves = []
    for h in range(2 if active else 0):
        # synthetic sweeps of this rank's (agent, frame) items of half h, resident in HBM
        pts = np.concatenate([synthetic_points(1, POINTS_PER_SWEEP, seed=1000 + 100000 * h + r) for r in shard.rows])
        halves.append({"points": torch.from_numpy(pts).to(dev),
                       "n_pts": torch.full((shard.per_rank,), POINTS_PER_SWEEP, dtype=torch.int32, device=dev),
                       "trans": torch.from_numpy(synthetic_poses(Bh, AGENTS, seed=99 + h)).to(dev),
                       "plan": shard.fusion_plan(torch.full((Bh, AGENTS), AGENTS), dev)})
    if active:
        model.packed(dev)
    wait_events = []    # (start, end) HIP events around the stream-level wait for an exchange: the EXPOSED part of it

    def timed_wait(work):
        if work is None:
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        runner.wait(work)
        e1.record()
        wait_events.append((e0, e1))

    def step():
        with torch.no_grad():
            a, b = halves
            fa, ga, wa = runner.begin(a["points"], a["n_pts"])
            fb, gb, wb = runner.begin(b["points"], b["n_pts"])
            timed_wait(wa)
            out_a = runner.decode(fa, ga, a["trans"], a["plan"])
            timed_wait(wb)
            out_b = runner.decode(fb, gb, b["trans"], b["plan"])
            return out_a, out_b

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range((max(args.warmup, 1) if args.graph else args.warmup) if active else 0):
        out = step()
    torch.cuda.synchronize()
    barrier()               # every rank's warm-up collectives are finished before anybody starts capturing
    wait_events.clear()

    mode = args.graph if active else -1
    if mode == 2 and use_dist:
        raise SystemExit("--graph 2 (whole step in one hipGraph) is for N = 1 without V2X_FORCE_DIST")
    if mode == 2:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        run = g.replay
        exec_mode = "one hipGraph per step"
    elif mode in (1, 4):
        # four collective-free segments (encoder A, encoder B, fusion + decoder + heads A, B), each a hipGraph; the exchange (RCCL, eager) runs
        # between them on static buffers.  mode 1 (default): the two half-batches on TWO STREAMS -- kernels of one half fill the tails and the
        # HBM-bound phases (heads, conv8_2, conv1_1) of the other: +2.3...3.0 % at N = 1 against the one-stream order (mode 4), same box;
        # four quarter-batches on four streams: -1.4 %.  The halves of a step are joined before the next step starts.
        # capture_error_mode="thread_local": RCCL's watchdog thread polls the events of earlier collectives while this
        # thread captures -- under the default global mode that query is "operation not permitted when stream is
        # capturing" and takes the process down (seen with V2X_FORCE_DIST=1 on one GPU)
        two = mode == 1
        shared_pool = torch.cuda.graph_pool_handle()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()] if two else [torch.cuda.current_stream()] * 2
        with torch.no_grad():
            for h, st in zip(halves, streams):
                h["xbuf"] = None
                h["stream"] = st
                # concurrent halves must not share intermediate buffers: one memory pool per half (one-stream order: one pool, replayed in capture order)
                h["cap"] = dict(pool=torch.cuda.graph_pool_handle() if two else shared_pool, capture_error_mode="thread_local")
                if use_dist:
                    h["xbuf"] = torch.empty((sworld * shard.per_rank, 32, 32, 256), dtype=torch.bfloat16, device=dev)
            for h in halves:
                h["g_enc"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(h["g_enc"], **h["cap"]):
                    h["feats"] = runner.encode(h["points"], h["n_pts"])
            for h in halves:
                h["g_dec"] = torch.cuda.CUDAGraph()
                gathered = h["xbuf"] if use_dist else h["feats"][L]
                with torch.cuda.graph(h["g_dec"], **h["cap"]):
                    h["out"] = runner.decode(h["feats"], gathered, h["trans"], h["plan"])

        def run():
            cur = torch.cuda.current_stream()
            works = []
            for h in halves:                    # encoder A, exchange A (async), encoder B, exchange B: the same host order on every rank
                if two:
                    h["stream"].wait_stream(cur)
                with torch.cuda.stream(h["stream"]):
                    h["g_enc"].replay()
                    works.append(runner.start_exchange(h["feats"][L], out=h["xbuf"])[1] if use_dist else None)
            for h, w in zip(halves, works):
                with torch.cuda.stream(h["stream"]):
                    timed_wait(w)
                    h["g_dec"].replay()
            if two:
                for h in halves:
                    cur.wait_stream(h["stream"])
        exec_mode = ("4 hipGraph segments per step (encoder A, encoder B, fusion+decoder+heads A, B); exchange between them; "
                     + ("the two half-batches on two streams, joined at the end of the step" if two else "one stream"))
    elif mode == 0:
        run = step
        exec_mode = "eager launches"
    else:
        run = lambda: None      # noqa: E731 -- a rank without items (agent-per-gpu, rank >= 5): barriers and timing only
        exec_mode = "idle rank"
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    wait_events.clear()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    graph_equals_eager = None           # checked at N = 1 in the graph modes
    if mode in (1, 4) and active and not use_dist:
        # the graphs' outputs (the last replay; the two halves ran concurrently in mode 1) against an eager, one-stream recomputation of the same
        # step: every kernel is deterministic and the halves are independent, so the logits must agree bit for bit
        ref = step()
        torch.cuda.synchronize()
        graph_equals_eager = all(torch.equal(h["out"][k], r[k]) for h, r in zip(halves, ref) for k in ("cls", "loc"))
        if not graph_equals_eager:      # reported in the record (never silently): the measurement above is of a step whose results are in doubt
            print("bench.py: WARNING: the hipGraph step's logits differ from the eager step's", file=sys.stderr, flush=True)
        del ref
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    fps = Bt * args.steps / elapsed
    # exposed exchange time per step on every rank (0 when the collective finished under the other half's compute)
    exposed = sum(e0.elapsed_time(e1) for e0, e1 in wait_events) / max(args.steps, 1)
    wait_events.clear()
    exposed_all = [exposed]
    if use_dist:
        ex = torch.tensor([exposed], dtype=torch.float64, device=dev)
        gl = [torch.zeros_like(ex) for _ in range(world)]
        dist.all_gather(gl, ex)
        exposed_all = [float(x) for x in gl]
    ranks_seen = dist.get_world_size() if use_dist else 1

    roofline = None
    kernels = None
    if not args.no_roofline and active:
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
        roofline["traffic_source"] = None
        for tname in TRAFFIC_FILES:
            tfile = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tfile) and args.frames_per_gpu == 128:  # 2 half-batches of 64 = the profiled launches
                with open(tfile) as fh:
                    tk = json.load(fh)["kernels"].get(dom)
                if tk:
                    roofline["traffic"] = tk["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/%s (committed rocprofv3 --pmc passes of this workload; not re-measured in this run)" % tname
                    break
        roofline.update({"kernel": dom, "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                         "launches_per_step": d["launches"] // n_inst,
                         "share_of_kernel_time": d["ms"] / total_ms,
                         "alg_flops_per_launch": d["flops"] / d["launches"],
                         "alg_bytes_per_launch": d["bytes"] / d["launches"],
                         "timing": "HIP events around every launch of a separate EAGER pass of the same step (not the timed "
                                   "graph replay; the per-kernel sum therefore exceeds ms_per_step by the event overhead)"})
        kernels = {k: {"us_per_step": v["ms"] * 1e3 / n_inst, "launches_per_step": v["launches"] // n_inst,
                       "tflops": v["flops"] / max(v["ms"], 1e-9) / 1e9, "gbs": v["bytes"] / max(v["ms"], 1e-9) / 1e6}
                   for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}

    # what THIS box sustains (v2x_calib_stream / v2x_calib_mfma, ~0.3 s): a streaming kernel at the read : write mixes of the HBM-bound layers and
    # a register-resident MFMA loop on random operands, with the shader clock it held -- lets fractions and rounds be compared box-free
    calibration = None
    if rank == 0 and active and not args.no_calibration:
        try:
            from v2x_sim_amd.calibrate import calibrate
            calibration = calibrate(dev)
            if roofline is not None:
                ceil = calibration["mfma_tflops"] if roofline["bound"] == "mfma" else 1e3 * min(calibration["copy_1_1_tbs"], calibration["copy_1_3_tbs"])
                roofline["frac_of_measured_ceiling"] = roofline["achieved"] / ceil
        except Exception as e:      # the headline record must not die with a side table
            calibration = {"error": repr(e)}

    latency = configs = training = None
    if rank == 0 and world == 1 and not args.no_extras and not force_dist:
        # free the step's graphs and buffers first: the extras build their own
        for h in halves:
            for k in ("g_enc", "g_dec", "feats", "out", "xbuf"):
                h.pop(k, None)
        out = run = None
        torch.cuda.empty_cache()
        latency = measure_latency(model, dev)
        latency["default_dispatch"] = measure_latency(model, dev, small_batch=False)
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
            bc = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bc)
            configs = bc.run_configs(64, 5, dev)
        except Exception as e:  # the headline record must not die with a side table
            configs = {"error": repr(e)}
        try:
            torch.cuda.empty_cache()
            training = bc.run_training(dev)
        except Exception as e:
            training = {"error": repr(e)}

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
                       "layout": args.layout, "active_ranks": sworld,
                       "half_batches": 2,
                       "points_per_agent": POINTS_PER_SWEEP, "bev": [256, 256, 13],
                       "sharding": ("agent-major (agent,frame) items, contiguous slices; async RCCL %s of the fusion "
                                    "maps of one half-batch overlapped with the other half's compute"
                                    % ("all-gather" if args.transport == "allgather" else "grouped send/recv of the needed rows"))
                                   if world > 1 else "single GPU, no collective",
                       "exec_mode": exec_mode, "hip_graph": bool(mode)},
            "whole_step_frac": (GFLOP_PER_FRAME_BASE + GFLOP_PER_GNN_ROUND * args.gnn_iters) * 1e9 * fps / world / (PEAK_MFMA_TFLOPS * 1e12),
            "graph_equals_eager": graph_equals_eager,
            "ranks_seen": ranks_seen, "exposed_exchange_ms_per_step": exposed_all,
            "exposed_exchange_note": "HIP-event time a half-batch's stream waits for its exchange; with the two half-batches on two streams the GPU runs "
                                     "the other half's kernels during that wait (one-stream order: --graph 4)",
            "roofline": roofline, "calibration": calibration, "cpu_baseline": cpu, "latency": latency, "configs": configs, "training": training, "kernels": kernels,
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
