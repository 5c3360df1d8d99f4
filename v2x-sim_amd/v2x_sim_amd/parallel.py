"""Agent-sharded multi-GPU execution (new functionality -- the reference
is single-process, so the correctness oracle is "R-rank result == 1-rank result").

Work items are the (agent, frame) maps of the agent-major batch (row = agent*Bt + frame, the
reference's own batching of agents).  Rank r of R owns the contiguous slice
[r*L, (r+1)*L), L = A*Bt/R.  Encoder, decoder and heads are independent per item; the ONLY
exchange is the fusion-layer feature map (256x32x32 bf16 = 512 KiB per item): one RCCL
all-gather over xGMI per GNN round puts every item's map on every rank, after which each rank
warps/fuses only the (ego, frame) pairs it owns.  Poses and agent counts are replicated (tiny).
With R = 1 there is no collective at all.
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
        return {"items": torch.tensor(items, dtype=torch.int32, device=device).view(-1, 2),
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


class ShardedV2VNet:
    """Runs a v2x_sim_amd V2VNet over this rank's shard.  `exchange` is injectable so that a
    single GPU can emulate R ranks in the equivalence test."""

    def __init__(self, model, shard, exchange=None, group=None):
        self.model, self.shard, self.group = model, shard, group
        self._custom_exchange = exchange is not None
        self.exchange = exchange or (lambda t: exchange_features(t, shard.world, group))
        self.grid = ops.VoxelGrid()

    def encode_points(self, points, n_pts, pk):
        """points (L, max_pts, stride) fp32 of this rank's items -> encoder pyramid."""
        X, Y, Z = self.grid.dims
        bits = ops.voxelize_bits(points, n_pts, self.grid)
        return LidarEncoder.run(pk["enc"], bits, zbits=Z)  # conv_pre_1 reads the bit grid directly

    def begin(self, points, n_pts):
        """Encoder of this rank's items, then START the exchange of the fusion-layer maps without waiting for it:
        -> (feats, gathered, work).  With world > 1 and the default transport the all-gather runs asynchronously on
        RCCL's stream (`async_op=True`), so the caller can launch more encoder work before calling finish()."""
        m, sh = self.model, self.shard
        pk = m.packed(points.device)
        feats = self.encode_points(points, n_pts, pk)
        local = feats[m.layer]
        if sh.world == 1 or self._custom_exchange:
            return feats, self.exchange(local), None
        out = torch.empty((sh.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=self.group, async_op=True)
        return feats, out, work

    def finish(self, feats, gathered, work, trans, plan):
        """Wait for the exchange started by begin(), then warp + ConvGRU + decoder + heads for this rank's items."""
        if work is not None:
            work.wait()  # makes the current stream wait for the collective; the host does not block
        m = self.model
        pk = m.packed(gathered.device)
        feats[m.layer] = self.fuse_local(feats, trans, plan, pk, gathered0=gathered)
        x = LidarDecoder.run(pk["dec"], *feats)
        return m.get_cls_loc_result(x, pk["heads"])

    def fuse_local(self, feats, trans, plan, pk, gathered0=None):
        m, sh = self.model, self.shard
        local = feats[m.layer]
        cur = local
        if gathered0 is None:
            gathered0 = self.exchange(local)
        for it in range(m.gnn_iter_num):
            src = gathered0 if (m.neighbor_source == "initial" or it == 0) else self.exchange(cur)
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
        x = LidarDecoder.run(pk["dec"], *feats)
        return m.get_cls_loc_result(x, pk["heads"])


class ShardedWhen2com:
    """when2com / who2com over an agent shard (BASELINE.json config 4).

    Exchange = (1) the per-item key (1024 fp32) and query (32 fp32) vectors -- 4 KiB per item, every rank then runs the
    5x5 handshake of every frame itself (it is ~5 kFLOP) -- and (2) the fusion-layer maps.  `sparse_fetch` names the
    communication-sparsity idea of when2com: only maps with a non-zero coefficient for one of this rank's items are
    needed; with an all-gather transport every map travels anyway, so the saving here is the skipped warps
    (warp_fuse never touches a source whose coefficient is 0).  A point-to-point fetch of just the needed maps is
    the natural next step on xGMI and is left for a later round."""

    def __init__(self, model, shard, exchange=None, group=None):
        self.model, self.shard, self.group = model, shard, group
        self.exchange = exchange or (lambda t: exchange_features(t, shard.world, group))
        self.grid = ops.VoxelGrid()

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
        return {"items": torch.tensor(items, dtype=torch.int32, device=device).view(-1, 2), "mask": mask.to(device),
                "local_rows": None if len(sel) == sh.per_rank else torch.tensor(sel, device=device),
                "q_idx": it[:, 0].to(device), "f_idx": it[:, 1].to(device)}

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
        keys = self.exchange(KmGenerator.run(pk["key"], y).contiguous())       # (A*Bt, 1024) on every rank
        querys = self.exchange(KmGenerator.run(pk["query"], y).contiguous())   # (A*Bt, 32)
        mode = "softmax" if (training or inference == "softmax") else inference
        prob, coef = ops.attn_handshake(keys, querys, pk["w_lin"], pk["b_lin"], sh.A, sh.Bt, mode)
        coef_items = coef[plan["f_idx"], :, plan["q_idx"]].contiguous() * plan["mask"]
        gathered = self.exchange(feats[m.layer])
        fused_items = ops.warp_fuse(gathered, sh.A, sh.Bt, trans, plan["items"], coef_items, V2X_FUSE_WSUM)
        if plan["local_rows"] is None:
            fused = fused_items
        else:
            fused = torch.zeros_like(feats[m.layer])
            fused.index_copy_(0, plan["local_rows"], fused_items)
        feats[m.layer] = fused
        x = LidarDecoder.run(pk["dec"], *feats)
        res = m.get_cls_loc_result(x, pk["heads"])
        res["prob_action"], res["coef"] = prob, coef
        return res
