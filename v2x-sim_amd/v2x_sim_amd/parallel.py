"""Agent-sharded multi-GPU execution (new functionality -- the reference
is single-process, so the correctness oracle is "R-rank result == 1-rank result").

Work items are the (agent, frame) maps of the agent-major batch (row = agent*Bt + frame, the
reference's own batching of agents).  Rank r of R owns the contiguous slice
[r*L, (r+1)*L), L = A*Bt/R.  Encoder, decoder and heads are independent per item; the ONLY
exchange is the fusion-layer feature map (256x32x32 bf16 = 512 KiB per item): one RCCL
all-gather over xGMI per GNN round puts every item's map on every rank, after which each rank
warps/fuses only the (ego, frame) pairs it owns.  Poses and agent counts are replicated (tiny).
With R = 1 there is no collective at all.

Two transports move the maps:
  * "allgather" (default for V2VNet, the north_star's layout): one RCCL all-gather per GNN round;
  * "needed" (`plan_row_exchange` + `sparse_exchange`): grouped point-to-point send/recv
    (`batch_isend_irecv` = one RCCL group call) of exactly the rows a rank reads.  V2VNet needs the
    maps of the frames in which the rank owns an ego ((A-1)/A of what the all-gather delivers at
    R >= A ... 4/7 at R = 8); when2com / who2com need only the (source, frame) maps whose attention
    coefficient for one of the rank's egos is non-zero -- the communication-sparsity path of
    BASELINE.json config 4 (keys and queries, 4 KiB per item, are all-gathered first so that every
    rank derives the same plan from the same handshake).
"""
import torch
import torch.distributed as dist

from . import ops
from ._lib import V2X_FUSE_MEAN
from .models.det.base import INPUT_C_PAD, LidarDecoder, LidarEncoder


class AgentShard:
    """Partition of the A*Bt agent-major work items over `world` ranks."""

    def __init__(self, agent_num, total_frames, rank=0, world=1):
        total = agent_num * total_frames
        if total % world != 0:
            raise ValueError("A*Bt=%d items do not divide over %d ranks: choose the frame count as a multiple of "
                             "the rank count" % (total, world))
        self.A, self.Bt, self.rank, self.world = agent_num, total_frames, rank, world
        self.per_rank = total // world
        self.lo, self.hi = rank * self.per_rank, (rank + 1) * self.per_rank
        self.rows = list(range(self.lo, self.hi))
        self.items = [(r // total_frames, r % total_frames) for r in self.rows]  # (agent, frame)

    def fusion_plan(self, num_agent_tensor, device):
        """items / coef of the maps this rank fuses (real agents only) + their local row index."""
        nat = num_agent_tensor.detach().to("cpu") if isinstance(num_agent_tensor, torch.Tensor) else num_agent_tensor
        counts = [int(nat[f][0]) for f in range(self.Bt)]
        if min(counts) < 2:
            raise RuntimeError("V2VNet needs >= 2 agents in every frame (stack expects a non-empty TensorList)")
        sel, coef = [], []
        for local, (a, f) in enumerate(self.items):
            if a < counts[f]:
                sel.append(local)
                coef.append([1.0 if (j != a and j < counts[f]) else 0.0 for j in range(self.A)])
        items = [self.items[i] for i in sel]
        full = len(sel) == self.per_rank
        return {"items": ops.items_tensor(items, self.A, self.Bt, device),
                "coef": torch.tensor(coef, dtype=torch.float32, device=device).view(-1, self.A),
                "local_rows": None if full else torch.tensor(sel, device=device), "n": len(sel)}


def exchange_features(local_feat, world, group=None, out=None):
    """All-gather of the per-rank fusion-layer maps: (L, H, W, C) -> (world*L, H, W, C), rank-major
    (= agent-major row order, because shards are contiguous).  RCCL on GPUs, gloo in the CPU tests."""
    if world == 1:
        return local_feat
    shape = (world * local_feat.shape[0],) + tuple(local_feat.shape[1:])
    if out is None:
        out = torch.empty(shape, dtype=local_feat.dtype, device=local_feat.device)
    src = local_feat.contiguous()
    if src.dtype == torch.bfloat16 and not src.is_cuda:
        # gloo has no bf16: move the bits (tests only; RCCL handles bf16 natively)
        dist.all_gather_into_tensor(out.view(torch.uint8), src.view(torch.uint8), group=group)
    else:
        dist.all_gather_into_tensor(out, src, group=group)
    return out


def plan_row_exchange(needs, per_rank):
    """needs[r] = iterable of GLOBAL row ids rank r must hold (rows it owns are dropped).  Rows are owned in contiguous
    slices of `per_rank`.  -> per rank {"send": [(dst, lo, hi)], "recv": [(src, lo, hi)], "rows": n_received}; ranges
    are half-open global row intervals that never cross an owner boundary, and for every (src, dst) pair the send list
    of src and the recv list of dst hold the same ranges in the same (ascending) order -- what a grouped
    send/recv needs to match.  Pure host logic, replicated identically on every rank."""
    world = len(needs)
    plans = [{"send": [], "recv": [], "rows": 0} for _ in range(world)]
    for dst in range(world):
        rows = sorted(set(int(x) for x in needs[dst]) - set(range(dst * per_rank, (dst + 1) * per_rank)))
        i = 0
        while i < len(rows):
            j = i
            while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and rows[j + 1] // per_rank == rows[i] // per_rank:
                j += 1
            lo, hi, src = rows[i], rows[j] + 1, rows[i] // per_rank
            if not 0 <= src < world:
                raise ValueError("row %d has no owner among %d ranks of %d rows" % (lo, world, per_rank))
            plans[dst]["recv"].append((src, lo, hi))
            plans[src]["send"].append((dst, lo, hi))
            plans[dst]["rows"] += hi - lo
            i = j + 1
    for p in plans:
        p["send"].sort()
        p["recv"].sort()
    return plans


def _wire(t):
    """gloo has no bf16 (and no int16 collectives): the CPU tests move the bits as uint8 (RCCL sends bf16 natively)."""
    return t.view(torch.uint8) if (t.dtype == torch.bfloat16 and not t.is_cuda) else t


def sparse_exchange(local, full, plan, rank, per_rank, group=None):
    """Point-to-point transport of the rows in `plan` (this rank's entry of plan_row_exchange).
    local (per_rank, ...) = the rows this rank owns; full (world*per_rank, ...) = agent-major buffer whose needed rows
    are filled (the rest is left untouched -- nothing reads it).  Returns the list of Work handles (wait() on each makes
    the current stream wait, as with the all-gather); the own rows are copied on the current stream."""
    base = rank * per_rank
    full[base:base + per_rank].copy_(local)
    # the plan speaks in ranks OF THE SHARD (0..R-1); P2POp's `peer` is a GLOBAL rank -- translate when the shard lives on a sub-group
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    ops_ = []
    for dst, lo, hi in plan["send"]:
        ops_.append(dist.P2POp(dist.isend, _wire(local[lo - base:hi - base]), peer(dst), group))
    for src, lo, hi in plan["recv"]:
        ops_.append(dist.P2POp(dist.irecv, _wire(full[lo:hi]), peer(src), group))
    return dist.batch_isend_irecv(ops_) if ops_ else []


class ShardedV2VNet:
    """Runs a v2x_sim_amd V2VNet over this rank's shard.  `exchange` is injectable so that a
    single GPU can emulate R ranks in the equivalence test."""

    def __init__(self, model, shard, exchange=None, group=None, transport="allgather"):
        if transport not in ("allgather", "needed"):
            raise ValueError("transport must be 'allgather' or 'needed'")
        self.model, self.shard, self.group, self.transport = model, shard, group, transport
        self._custom_exchange = exchange is not None
        self.exchange = exchange or (lambda t: exchange_features(t, shard.world, group))
        self.grid = ops.VoxelGrid()
        self._xplan = None

    def needed_plan(self, counts=None):
        """Row-exchange plan of the "needed" transport: rank r reads the rows (j, f) of every agent j < count[f] for the
        frames f in which it owns a real ego.  Replicated host logic (every rank computes all ranks' needs)."""
        sh = self.shard
        if counts is None:
            counts = [sh.A] * sh.Bt
        needs = []
        for r in range(sh.world):
            frames = set()
            for row in range(r * sh.per_rank, (r + 1) * sh.per_rank):
                a, f = divmod(row, sh.Bt)
                if a < counts[f]:
                    frames.add(f)
            needs.append([j * sh.Bt + f for f in frames for j in range(counts[f])])
        return plan_row_exchange(needs, sh.per_rank)

    def encode(self, points, n_pts):
        """a1 + a2 for this rank's items (no collective: capturable in a hipGraph)."""
        return self.encode_points(points, n_pts, self.model.packed(points.device))

    def exchange_round(self, cur):
        """Synchronous exchange of a later GNN round's UPDATED maps (neighbor_source='updated'): the same transport and -- for
        'needed' -- the same row plan as the first round (which rows a rank reads depends on the frames it owns, not on the round)."""
        sh = self.shard
        if sh.world == 1 or self._custom_exchange or self.transport == "allgather":
            return self.exchange(cur)
        gathered, work = self.start_exchange(cur)
        self.wait(work)
        return gathered

    def start_exchange(self, local, out=None, counts=None):
        """START the exchange of the fusion-layer maps without waiting for it -> (gathered, work).  `out` = optional
        static (world*per_rank, H, W, C) destination (bench.py keeps it outside its hipGraph segments).  work is None,
        one Work (all-gather) or a list of Works (point-to-point)."""
        sh = self.shard
        if sh.world == 1 or self._custom_exchange:
            return self.exchange(local), None
        local = local.contiguous()
        if out is None:
            out = torch.empty((sh.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if self.transport == "needed":
            if self._xplan is None or counts is not None:
                self._xplan = self.needed_plan(counts)
            return out, sparse_exchange(local, out, self._xplan[sh.rank], sh.rank, sh.per_rank, self.group)
        return out, dist.all_gather_into_tensor(_wire(out), _wire(local), group=self.group, async_op=True)

    @staticmethod
    def wait(work):
        """Stream-level wait for start_exchange's handle(s); the host does not block on RCCL."""
        if work is None:
            return
        for w in (work if isinstance(work, (list, tuple)) else (work,)):
            w.wait()

    def decode(self, feats, gathered, trans, plan):
        """a3 + a4 + a6 + a7 for this rank's items once the maps are there (no collective: capturable)."""
        m = self.model
        pk = m.packed(gathered.device)
        feats = list(feats)
        feats[m.layer] = self.fuse_local(feats, trans, plan, pk, gathered0=gathered)
        return m.decode_heads(pk, feats)

    def encode_points(self, points, n_pts, pk):
        """points (L, max_pts, stride) fp32 of this rank's items -> encoder pyramid."""
        X, Y, Z = self.grid.dims
        bits = ops.voxelize_bits(points, n_pts, self.grid)
        return LidarEncoder.run(pk["enc"], bits, zbits=Z)  # conv_pre_1 reads the bit grid directly

    def begin(self, points, n_pts):
        """Encoder of this rank's items, then START the exchange of the fusion-layer maps without waiting for it:
        -> (feats, gathered, work).  With world > 1 and the default transport the all-gather runs asynchronously on
        RCCL's stream (`async_op=True`), so the caller can launch more encoder work before calling finish()."""
        feats = self.encode(points, n_pts)
        gathered, work = self.start_exchange(feats[self.model.layer])
        return feats, gathered, work

    def finish(self, feats, gathered, work, trans, plan):
        """Wait for the exchange started by begin(), then warp + ConvGRU + decoder + heads for this rank's items."""
        self.wait(work)  # makes the current stream wait for the collective; the host does not block
        return self.decode(feats, gathered, trans, plan)

    def fuse_local(self, feats, trans, plan, pk, gathered0=None):
        m, sh = self.model, self.shard
        local = feats[m.layer]
        cur = local
        if gathered0 is None:
            gathered0 = self.exchange(local)
        rounds = m.gnn_rounds() if hasattr(m, "gnn_rounds") else m.gnn_iter_num
        for it in range(rounds):
            src = gathered0 if (m.neighbor_source != "updated" or it == 0) else self.exchange_round(cur)
            mean = ops.warp_fuse(src, sh.A, sh.Bt, trans, plan["items"], plan["coef"], V2X_FUSE_MEAN)
            rows = plan["local_rows"]
            ego = cur if rows is None else cur.index_select(0, rows)
            h = ops.run_layer(pk["gru"], ego, mean)
            if rows is None:
                cur = h
            else:
                cur = cur.clone()
                cur.index_copy_(0, rows, h)
        return cur

    def forward_points(self, points, n_pts, trans, plan):
        m = self.model
        pk = m.packed(points.device)
        feats = self.encode_points(points, n_pts, pk)
        feats[m.layer] = self.fuse_local(feats, trans, plan, pk)
        return m.decode_heads(pk, feats)


def when2com_needs(shard, coef, counts):
    """Rows every rank must hold after the handshake: (k, f) for each owned real ego (q, f) and each real source k != q
    whose coefficient coef[f][k][q] is non-zero.  coef: (Bt, A, A) host tensor / nested list, identical on every rank
    (each rank runs the same handshake on the same all-gathered keys and queries)."""
    needs = []
    for r in range(shard.world):
        rows = set()
        for row in range(r * shard.per_rank, (r + 1) * shard.per_rank):
            q, f = divmod(row, shard.Bt)
            if q >= counts[f]:
                continue
            for k in range(counts[f]):
                if k != q and float(coef[f][k][q]) != 0.0:
                    rows.add(k * shard.Bt + f)
        needs.append(rows)
    return needs


class ShardedWhen2com:
    """when2com / who2com over an agent shard (BASELINE.json config 4).

    Exchange = (1) the per-item key (1024 fp32) and query (32 fp32) vectors -- 4 KiB per item, all-gathered; every rank
    then runs the 5x5 handshake of every frame itself (it is ~5 kFLOP) and so holds the same coefficients -- and (2) the
    fusion-layer maps.  transport="sparse" (default) is when2com's communication sparsity made literal: the
    coefficients go to the host (one small D2H copy), every rank derives the same row-exchange plan, and one grouped
    send/recv moves ONLY the maps with a non-zero coefficient for an ego the receiver owns (`self.last_comm` reports the
    rows moved next to what the all-gather would have moved).  'softmax' inference / training weights are dense, so
    that mode (and transport="allgather") all-gathers the maps; warp_fuse skips zero-coefficient sources either way."""

    def __init__(self, model, shard, exchange=None, group=None, transport="sparse"):
        if transport not in ("allgather", "sparse"):
            raise ValueError("transport must be 'allgather' or 'sparse'")
        self.model, self.shard, self.group, self.transport = model, shard, group, transport
        self._custom_exchange = exchange is not None
        self.exchange = exchange or (lambda t: exchange_features(t, shard.world, group))
        self.grid = ops.VoxelGrid()
        self.last_comm = None

    def plan(self, num_agent_tensor, device):
        sh = self.shard
        nat = num_agent_tensor.detach().to("cpu") if isinstance(num_agent_tensor, torch.Tensor) else num_agent_tensor
        counts = [int(nat[f][0]) for f in range(sh.Bt)]
        sel = [i for i, (a, f) in enumerate(sh.items) if a < counts[f]]
        items = [sh.items[i] for i in sel]
        mask = torch.zeros((len(items), sh.A), dtype=torch.float32)
        for m, (a, f) in enumerate(items):
            mask[m, :counts[f]] = 1.0
        it = torch.tensor(items, dtype=torch.int64).view(-1, 2)
        return {"items": ops.items_tensor(items, sh.A, sh.Bt, device), "mask": mask.to(device),
                "local_rows": None if len(sel) == sh.per_rank else torch.tensor(sel, device=device),
                "q_idx": it[:, 0].to(device), "f_idx": it[:, 1].to(device), "counts": counts}

    def fetch_maps(self, local, coef, counts, mode):
        """Second exchange: -> (A*Bt, H, W, C) agent-major buffer holding at least every map this rank's fusion reads.
        Pure torch.distributed (runs under gloo in the CPU tests)."""
        sh = self.shard
        if sh.world == 1 or self._custom_exchange:
            return self.exchange(local)
        dense_rows = (sh.world - 1) * sh.per_rank
        if self.transport == "allgather" or mode == "softmax":
            self.last_comm = {"transport": "allgather", "rows_received": dense_rows, "rows_allgather": dense_rows,
                              "bytes_received": dense_rows * local[0].numel() * local.element_size()}
            return exchange_features(local, sh.world, self.group)
        plans = plan_row_exchange(when2com_needs(sh, coef.detach().to("cpu"), counts), sh.per_rank)
        mine = plans[sh.rank]
        local = local.contiguous()
        full = torch.empty((sh.world * sh.per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for w in sparse_exchange(local, full, mine, sh.rank, sh.per_rank, self.group):
            w.wait()
        map_bytes = local[0].numel() * local.element_size()
        self.last_comm = {"transport": "sparse", "rows_received": mine["rows"], "rows_allgather": dense_rows,
                          "rows_sent": sum(hi - lo for _, lo, hi in mine["send"]),
                          "bytes_received": mine["rows"] * map_bytes,
                          "bytes_sent": sum(hi - lo for _, lo, hi in mine["send"]) * map_bytes,
                          "messages": len(mine["send"]) + len(mine["recv"])}
        return full

    def forward_bits(self, bits, zbits, trans, plan, training=False, inference="activated"):
        """bits: this rank's (L, X, Y) int32 occupancy words."""
        from ._lib import V2X_FUSE_WSUM
        from .models.det.When2com import KmGenerator
        m, sh = self.model, self.shard
        pk = m.packed(bits.device)
        feats = LidarEncoder.run(pk["enc"], bits, zbits=zbits)
        y = LidarEncoder.run(pk["tower"]["enc"], bits, zbits=zbits)[4]
        for layer in pk["tower"]["convs"]:
            y = ops.run_layer(layer, y)
        k_loc, q_loc = KmGenerator.run_pair(pk["keyquery"], y)     # both MLPs as one chain of three launches
        keys = self.exchange(k_loc)                                # (A*Bt, 1024) on every rank
        querys = self.exchange(q_loc)                              # (A*Bt, 32)
        mode = "softmax" if (training or inference == "softmax") else inference
        prob, coef = ops.attn_handshake(keys, querys, pk["w_lin"], pk["b_lin"], sh.A, sh.Bt, mode)
        if getattr(m, "renormalize", False) and mode == "activated":          # the second readings of ASSUMPTIONS rows 31 / 30, as in When2com.forward_nhwc
            coef = m.renormalized(coef)
        transposed = getattr(m, "attn_index", "kq") == "qk"
        coef_items = (coef[plan["f_idx"], plan["q_idx"], :] if transposed else coef[plan["f_idx"], :, plan["q_idx"]]).contiguous() * plan["mask"]
        # (the sparse transport plans from coef[f][k][q] = "target q reads source k": hand it that orientation)
        gathered = self.fetch_maps(feats[m.layer], coef.transpose(1, 2) if transposed else coef, plan["counts"], mode)
        fused_items = ops.warp_fuse(gathered, sh.A, sh.Bt, trans, plan["items"], coef_items, V2X_FUSE_WSUM)
        if plan["local_rows"] is None:
            fused = fused_items
        else:
            fused = torch.zeros_like(feats[m.layer])
            fused.index_copy_(0, plan["local_rows"], fused_items)
        feats[m.layer] = fused
        res = m.decode_heads(pk, feats)
        res["prob_action"], res["coef"] = prob, coef
        return res
