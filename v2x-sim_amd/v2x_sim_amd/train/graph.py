"""Differentiable graph of the detection baselines for TRAINING.

Inference runs the hand-written HIP kernels (models/det/*.forward).  Training needs batch-statistics BatchNorm and
gradients; SURVEY.md section 8(b) allows the backward to "stay in PyTorch-ROCm ops", and that is what this module is:
the same graph written with torch ops on the MI355X (MIOpen convolutions, autograd), over the SAME parameter tree --
the nn.Conv2d / nn.BatchNorm2d containers that models/det/base.py holds under the upstream names -- so an optimizer
steps the very tensors the HIP engine packs (DetModelBase.packed() notices the version bump and re-packs).

Graph = upstream Backbone.py / DetModelBase.py / V2VNet.py / When2com.py / {Sum,Mean,Max,Cat}Fusion.py / DiscoNet.py semantics (PyTorch 1.8 per README.md:88-95): nearest x2
upsample, concat (up, skip), two-step affine_grid/grid_sample warp with align_corners=False, ConvGRU step with h0 = 0.
It is NOT a fallback for inference: models refuse to run forward() off the GPU library, and this graph is only
reached through FaFModule.step() / train_forward().
"""
import torch
import torch.nn.functional as F

from .. import tuning


def _hip_train():
    return tuning.get("TRAIN_HIP_CONV") == 1


def _conv(x, conv):
    """nn.Conv2d forward + backward: MIOpen through torch.autograd (default), or -- V2X_TRAIN_HIP_CONV=1, eligible 3x3 stride-1
    layers on the MI355X -- the hand-written HIP forward / dgrad / wgrad kernels (train/hip_conv.py)."""
    if x.is_cuda and _hip_train():
        from . import hip_conv
        if hip_conv.eligible(conv.weight, conv.stride, conv.padding, x.shape[2], x.shape[3]):
            return hip_conv.conv2d_nchw(x, conv)
    return conv(x)


def _cbr(x, conv, bn):
    return F.relu(bn(_conv(x, conv)))


def _cbr_each(x, conv, bn, order=None):
    """conv + BN + ReLU where upstream calls the layer once per map (batch of 1) inside its per-ego loop: with batch
    statistics on, each map is normalised by ITS OWN statistics and the running averages take one update per map, in order."""
    y = conv(x)
    if not bn.training:
        return F.relu(bn(y))
    out = [None] * y.shape[0]
    for i in (order if order is not None else range(y.shape[0])):   # upstream's call order: the running averages depend on it
        out[i] = bn(y[i:i + 1])
    return F.relu(torch.cat(out, 0))


def _conv3d_1x1(x, m):
    """upstream Conv3D on a length-1 sequence: (N, C, H, W) -> 1x1x1 conv3d + BN3d + ReLU."""
    return F.relu(m.bn3d(m.conv3d(x.unsqueeze(2)))).squeeze(2)


def encoder(e, x):
    """x (N, Z, X, Y) fp32 -> [x, x_1, x_2, x_3, x_4]."""
    x = _cbr(x, e.conv_pre_1, e.bn_pre_1)
    x = _cbr(x, e.conv_pre_2, e.bn_pre_2)
    x_1 = _cbr(x, e.conv1_1, e.bn1_1)
    x_1 = _conv3d_1x1(_cbr(x_1, e.conv1_2, e.bn1_2), e.conv3d_1)
    x_2 = _cbr(x_1, e.conv2_1, e.bn2_1)
    x_2 = _conv3d_1x1(_cbr(x_2, e.conv2_2, e.bn2_2), e.conv3d_2)
    x_3 = _cbr(_cbr(x_2, e.conv3_1, e.bn3_1), e.conv3_2, e.bn3_2)
    x_4 = _cbr(_cbr(x_3, e.conv4_1, e.bn4_1), e.conv4_2, e.bn4_2)
    return [x, x_1, x_2, x_3, x_4]


def decoder(d, x, x_1, x_2, x_3, x_4):
    up = lambda t: F.interpolate(t, scale_factor=(2, 2))  # noqa: E731  (nearest)
    y = _cbr(_cbr(torch.cat((up(x_4), x_3), 1), d.conv5_1, d.bn5_1), d.conv5_2, d.bn5_2)
    y = _cbr(_cbr(torch.cat((up(y), x_2), 1), d.conv6_1, d.bn6_1), d.conv6_2, d.bn6_2)
    y = _cbr(_cbr(torch.cat((up(y), x_1), 1), d.conv7_1, d.bn7_1), d.conv7_2, d.bn7_2)
    return _cbr(_cbr(torch.cat((up(y), x), 1), d.conv8_1, d.bn8_1), d.conv8_2, d.bn8_2)


def heads(model, x):
    c, bp = model.classification, model.regression.box_prediction
    cls = c.conv2(_cbr(x, c.conv1, c.bn1)).permute(0, 2, 3, 1).contiguous()
    loc = bp[3](_cbr(x, bp[0], bp[1])).permute(0, 2, 3, 1).contiguous()
    return {"cls": cls.view(cls.shape[0], -1, model.category_num),
            "loc": loc.view(-1, loc.size(1), loc.size(2), model.anchor_num_per_loc, model.out_seq_len,
                            model.box_code_size)}


# train/hip_graph.py sets this while it runs a fusion stage: (feat, theta) -> resampled maps on the hand-written kernels (forward AND data
# gradient; the PyTorch backward of grid_sample is an atomic scatter -- 17 % of a V2VNet training step, and not bit-reproducible)
_affine_sample_override = None


def warp_batch(feat, T):
    """feat (P, C, H, W); T (P, 4, 4) pose of the source w.r.t. the ego -> maps in the ego frame
    (upstream feature_transformation: rotate about the map centre, then translate by (4*T03/128, -4*T13/128))."""
    P = feat.shape[0]
    z = torch.zeros(P, device=feat.device, dtype=feat.dtype)
    o = torch.ones(P, device=feat.device, dtype=feat.dtype)
    rot = torch.stack([torch.stack([T[:, 0, 0], T[:, 0, 1], z], 1), torch.stack([T[:, 1, 0], T[:, 1, 1], z], 1)], 1)
    tr = torch.stack([torch.stack([o, z, 4 * T[:, 0, 3] / 128], 1), torch.stack([z, o, -4 * T[:, 1, 3] / 128], 1)], 1)
    if _affine_sample_override is not None and feat.is_cuda and feat.dtype == torch.float32:
        return _affine_sample_override(_affine_sample_override(feat, rot), tr)
    g1 = F.affine_grid(rot, feat.shape, align_corners=False)
    g2 = F.affine_grid(tr, feat.shape, align_corners=False)
    y = F.grid_sample(feat, g1, mode="bilinear", padding_mode="zeros", align_corners=False)
    return F.grid_sample(y, g2, mode="bilinear", padding_mode="zeros", align_corners=False)


class _GruGatesHip(torch.autograd.Function):
    """The gate arithmetic of _gru_step as one launch forward and one backward (csrc/gru_train.hip) instead of 8 + ~14 elementwise PyTorch ops."""

    @staticmethod
    def forward(ctx, gi, bias_hh):
        from .. import ops
        gi = gi.contiguous()
        bhh = bias_hh.detach().contiguous()
        ctx.save_for_backward(gi, bhh)
        return ops.gru_gates(gi, bhh)

    @staticmethod
    def backward(ctx, dh):
        from .. import ops
        gi, bhh = ctx.saved_tensors
        dgi, dn_r = ops.gru_gates_backward(gi, bhh, dh.contiguous())
        C = bhh.numel() // 3
        dbhh = torch.cat((dgi[:, :2 * C].sum((0, 2, 3)), dn_r.sum((0, 2, 3)))) if ctx.needs_input_grad[1] else None
        return dgi, dbhh


def _gru_step(g, x, conv=None):
    """convolutional_rnn.Conv2dGRU cell with hidden=None (h0 = 0): gh = b_hh exactly, h = n + z*(0 - n).
    conv: optional replacement of the input convolution (train/hip_graph.py runs it on the hand-written kernels)."""
    gi = F.conv2d(x, g.weight_ih_l0, g.bias_ih_l0, 1, g.kernel_size // 2) if conv is None else conv(x, g.weight_ih_l0, g.bias_ih_l0)
    if conv is not None and gi.is_cuda and gi.dtype == torch.float32 and (gi.shape[2] * gi.shape[3]) % 4 == 0 and g.bias_hh_l0.dtype == torch.float32:
        from .. import tuning
        if tuning.get("TRAIN_GATES_HIP") != 0:            # the HIP engine's fusion stage (hip_graph passes its convolution): gates on the kernels too
            return _GruGatesHip.apply(gi, g.bias_hh_l0)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = g.bias_hh_l0.view(1, -1, 1, 1).chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    zg = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return n - zg * n


class _GatherRowsDup(torch.autograd.Function):
    """base.index_select(0, src) where every row of `base` is selected exactly K times (a frame's map is warped into each of its K = A - 1
    neighbours' frames).  index_select's own backward is index_add_ -- atomic fp32 adds, a different sum order every run; here the gradient of
    row r is the dense sum over its K uses in pair order (inv[r]): the V2VNet training step becomes bit-reproducible run to run."""

    @staticmethod
    def forward(ctx, base, src, inv):
        ctx.save_for_backward(inv)
        return base.index_select(0, src)

    @staticmethod
    def backward(ctx, dy):
        inv, = ctx.saved_tensors
        R, K = inv.shape
        return dy.index_select(0, inv.reshape(-1)).view((R, K) + tuple(dy.shape[1:])).sum(1), None, None


def v2v_fuse(model, feat, trans, num_agent_tensor, B, gru_conv=None):
    """feat (A*B, C, H, W) agent-major -> updated maps (same shape).  gru_conv: see _gru_step."""
    A = model.agent_num
    counts, items, rows = model.frame_plan(num_agent_tensor, B, A)
    if min(counts) < 2:
        raise RuntimeError("V2VNet needs >= 2 agents in every frame (stack expects a non-empty TensorList)")
    dev = feat.device
    # the plan's index tensors are cached per (agent table, B, device): building them is a host -> device copy, which must not happen
    # inside a hipGraph capture (train/graph_step.py captures the step after warm-up calls have filled this cache)
    A1, A2 = trans.shape[1], trans.shape[2]
    key = (tuple(counts), B, A, A1, A2, str(dev), str(feat.dtype))
    cache = model.__dict__.setdefault("_v2v_plan_cache", {})
    if key not in cache:
        pairs = [(m, j * B + f, f, a, j) for m, (a, f) in enumerate(items) for j in range(counts[f]) if j != a]
        cache.clear()
        # inv[r] = the pairs that read row r of the maps, in pair order -- defined when every row is read equally often (every frame full)
        uses = {}
        for pi, p in enumerate(pairs):
            uses.setdefault(p[1], []).append(pi)
        K = len(pairs) // max(feat.shape[0], 1)
        inv = None
        if K >= 1 and len(uses) == feat.shape[0] and all(len(v) == K for v in uses.values()):
            inv = torch.tensor([uses[r] for r in range(feat.shape[0])], device=dev)
        cache[key] = (torch.tensor([p[1] for p in pairs], device=dev), torch.tensor([p[0] for p in pairs], device=dev),
                      torch.tensor([(f * A1 + a) * A2 + j for (_, _, f, a, j) in pairs], device=dev),
                      torch.tensor([counts[f] - 1 for (_, f) in items], device=dev, dtype=feat.dtype).view(-1, 1, 1, 1),
                      torch.tensor(rows, device=dev), inv)
    src, dst, tsel, cnt, rows_t, inv = cache[key]
    Tp = trans.reshape(-1, 4, 4).index_select(0, tsel).to(feat.dtype)        # trans[f, a, j] of every (ego item, neighbour) pair
    cur = feat
    for _ in range(model.gnn_rounds()):
        base = cur if model.neighbor_source == "updated" else feat
        warped = warp_batch(_GatherRowsDup.apply(base, src, inv) if inv is not None else base.index_select(0, src), Tp)
        if min(counts) == max(counts):
            # every frame has the same number of agents: the pairs of an ego item are consecutive (see `pairs`), so the mean over its neighbours is
            # a reduction over a dense axis -- one deterministic kernel instead of zeros + index_add_ (atomic adds) + a division
            mean = warped.view((len(items), counts[0] - 1) + tuple(feat.shape[1:])).mean(1)
        else:
            mean = torch.zeros((len(items),) + tuple(feat.shape[1:]), device=dev, dtype=feat.dtype).index_add_(0, dst, warped) / cnt
        h = _gru_step(model.convgru, torch.cat([cur.index_select(0, rows_t), mean], 1), gru_conv)
        cur = cur.index_copy(0, rows_t, h)
    return cur


def when2com_fuse(model, x, feat, trans, num_agent_tensor, B, training=True, inference="softmax"):
    """when2com / who2com fusion (upstream When2com.py): policy tower -> key / query MLPs -> scores = key . Linear(query),
    softmax over the keys; training uses the soft scores, inference 'activated' / 'argmax_test' as the HIP path.
    x (A*B, Z, X, Y) input, feat (A*B, C, H, W) fusion-layer maps -> fused maps (zeros for padding agents)."""
    A = model.agent_num
    qk = model.query_key_net
    y = encoder(qk.lidar_encoder, x)[4]
    for i in range(1, 6):
        u = getattr(qk, "conv%d" % i).cbr_unit
        y = _cbr(y, u[0], u[1])
    flat = y.reshape(y.shape[0], -1)                       # NCHW flatten, as upstream
    keys, querys = model.key_net.fc(flat), model.query_net.fc(flat)
    key_mat = torch.stack([keys[B * i:B * (i + 1)] for i in range(A)], 1)       # (B, A, K)
    query_mat = torch.stack([querys[B * i:B * (i + 1)] for i in range(A)], 1)   # (B, A, Q)
    prob = torch.softmax(torch.bmm(key_mat, model.attention_net.linear(query_mat).transpose(2, 1)), dim=1)  # (B, k, q)
    if training or inference == "softmax":
        coef = prob
    elif inference == "activated":
        coef = prob * (prob > 0.2).to(prob.dtype)
    elif inference == "argmax_test":
        coef = F.one_hot(prob.max(dim=1)[1], num_classes=A).to(prob.dtype).transpose(1, 2)
    else:
        raise ValueError("Incorrect inference mode")
    counts, items, rows = model.frame_plan(num_agent_tensor, B, A)
    dev = feat.device
    pairs = [(m, k * B + f, f, q, k) for m, (q, f) in enumerate(items) for k in range(counts[f])]
    src = torch.tensor([p[1] for p in pairs], device=dev)
    dst = torch.tensor([p[0] for p in pairs], device=dev)
    Tp = torch.stack([trans[f, q, k] for (_, _, f, q, k) in pairs]).to(feat.dtype)
    own = torch.tensor([1.0 if q == k else 0.0 for (_, _, _, q, k) in pairs], device=dev, dtype=feat.dtype).view(-1, 1, 1, 1)
    maps = feat.index_select(0, src)
    val = own * maps + (1.0 - own) * warp_batch(maps, Tp)                       # the ego map is taken unwarped
    w = torch.stack([coef[f, k, q] for (_, _, f, q, k) in pairs]).view(-1, 1, 1, 1)
    fused_items = torch.zeros((len(items),) + tuple(feat.shape[1:]), device=dev, dtype=feat.dtype).index_add_(0, dst, w * val)
    fused = torch.zeros_like(feat).index_copy(0, torch.tensor(rows, device=dev), fused_items)
    return fused, prob, coef


def _ego_views(model, feat, trans, num_agent_tensor, B):
    """For the simple fusion baselines: every (ego item m, source k) map in the ego's frame (the ego's own map unwarped).
    -> val (n_pairs, C, H, W), dst (n_pairs,) item index, own (n_pairs,) bool 'k is the ego', items, rows, counts,
    pairs [(item, source row, frame, ego agent, source agent)]."""
    A = model.agent_num
    counts, items, rows = model.frame_plan(num_agent_tensor, B, A)
    dev = feat.device
    pairs = [(m, k * B + f, f, q, k) for m, (q, f) in enumerate(items) for k in range(counts[f])]
    src = torch.tensor([p[1] for p in pairs], device=dev)
    dst = torch.tensor([p[0] for p in pairs], device=dev)
    Tp = torch.stack([trans[f, q, k] for (_, _, f, q, k) in pairs]).to(feat.dtype)
    own = torch.tensor([q == k for (_, _, _, q, k) in pairs], device=dev)
    maps = feat.index_select(0, src)
    o = own.to(feat.dtype).view(-1, 1, 1, 1)
    val = o * maps + (1.0 - o) * warp_batch(maps, Tp)
    return val, dst, own, items, rows, counts, pairs


def simple_fuse(model, feat, trans, num_agent_tensor, B):
    """Sum / Mean / Max / Cat fusion (upstream {Sum,Mean,Max,Cat}Fusion.py on FusionBase.py): reduce [ego map, warped
    neighbour maps] per ego; CatFusion = ModulationLayer3(cat(ego, mean))."""
    from .._lib import V2X_FUSE_MAX, V2X_FUSE_MEAN, V2X_FUSE_WSUM
    val, dst, own, items, rows, counts, _ = _ego_views(model, feat, trans, num_agent_tensor, B)
    n = len(items)
    shape = (n,) + tuple(feat.shape[1:])
    mode = model.FUSE_MODE
    if mode == V2X_FUSE_MAX:
        fused = torch.full(shape, float("-inf"), device=feat.device, dtype=feat.dtype).index_reduce(
            0, dst, val, "amax", include_self=True)
    else:
        fused = torch.zeros(shape, device=feat.device, dtype=feat.dtype).index_add(0, dst, val)
        if mode == V2X_FUSE_MEAN:
            cnt = torch.tensor([counts[f] for (_, f) in items], device=feat.device, dtype=feat.dtype).view(-1, 1, 1, 1)
            fused = fused / cnt
        elif mode != V2X_FUSE_WSUM:
            raise ValueError("unknown fusion mode")
    rows_t = torch.tensor(rows, device=feat.device)
    if hasattr(model, "modulation_layer_3"):
        m = model.modulation_layer_3
        order = sorted(range(n), key=lambda i: (items[i][1], items[i][0]))          # upstream loops frames, then egos
        fused = _cbr_each(torch.cat([feat.index_select(0, rows_t), fused], 1), m.conv1_1, m.bn1_1, order)
    return feat.index_copy(0, rows_t, fused)


def disco_fuse(model, feat, trans, num_agent_tensor, B):
    """DiscoNet's pixel-wise weighted fusion (upstream DiscoNet.py, no teacher): score = PixelWeightedFusionSoftmax(
    cat(ego, source)) per source, weights = exp(score) normalised over the ego's sources (no max-subtraction, as
    upstream), fused = sum of weight * source map."""
    val, dst, own, items, rows, counts, pairs = _ego_views(model, feat, trans, num_agent_tensor, B)
    # upstream: frames, then egos, then [ego itself, neighbours in agent order]
    order = sorted(range(len(pairs)), key=lambda i: (pairs[i][2], pairs[i][3], pairs[i][3] != pairs[i][4], pairs[i][4]))
    rows_t = torch.tensor(rows, device=feat.device)
    ego = feat.index_select(0, rows_t).index_select(0, dst)
    m = model.pixel_weighted_fusion
    x = _cbr_each(torch.cat([ego, val], 1), m.conv1_1, m.bn1_1, order)   # upstream scores one (ego, source) pair per call
    x = _cbr_each(x, m.conv1_2, m.bn1_2, order)
    x = _cbr_each(x, m.conv1_3, m.bn1_3, order)
    w = torch.exp(F.relu(m.conv1_4(x)))                                      # (n_pairs, 1, H, W)
    n = len(items)
    total = torch.zeros((n, 1) + tuple(w.shape[2:]), device=feat.device, dtype=feat.dtype).index_add(0, dst, w)
    fused = torch.zeros((n,) + tuple(feat.shape[1:]), device=feat.device, dtype=feat.dtype).index_add(
        0, dst, (w / total.index_select(0, dst)) * val)
    return feat.index_copy(0, rows_t, fused)


def train_forward(model, bevs, trans_matrices=None, num_agent_tensor=None, batch_size=1, inference="softmax"):
    """bevs (A*B, 1, X, Y, Z) dense occupancy (the Dataset format) -> {'loc', 'cls'} with the shapes of the HIP path.
    Uses batch-statistics BN when model.training, running statistics otherwise.
    V2X_TRAIN_HIP=1: in train mode encoder, decoder and heads run as the bf16 NHWC graph on the hand-written kernels (train/hip_graph.py)."""
    if tuning.get("TRAIN_HIP") == 1 and model.training and bevs.is_cuda:
        from . import hip_graph
        return hip_graph.train_forward(model, bevs, trans_matrices, num_agent_tensor, batch_size, inference)
    x = bevs[:, 0].permute(0, 3, 1, 2).to(torch.float32)
    if hasattr(model, "outc"):                      # segmentation variants: det backbone + 1x1 head, NHWC fp32 logits
        if hasattr(model, "stpn"):
            y = decoder(model.stpn.decoder, *encoder(model.stpn.encoder, x))
        else:
            feats = encoder(model.u_encoder, x)
            feats[model.layer] = v2v_fuse(model, feats[model.layer], trans_matrices.to(x.device), num_agent_tensor, batch_size)
            y = decoder(model.decoder, *feats)
        return model.outc.conv(y).permute(0, 2, 3, 1).contiguous()
    if hasattr(model, "stpn"):                      # FaFNet: lowerbound / upperbound
        feats = encoder(model.stpn.encoder, x)
        return heads(model, decoder(model.stpn.decoder, *feats))
    if hasattr(model, "convgru"):                   # V2VNet
        feats = encoder(model.u_encoder, x)
        feats[model.layer] = v2v_fuse(model, feats[model.layer], trans_matrices.to(x.device), num_agent_tensor, batch_size)
        return heads(model, decoder(model.decoder, *feats))
    if hasattr(model, "query_key_net"):             # when2com / who2com
        feats = encoder(model.u_encoder, x)
        feats[model.layer], prob, coef = when2com_fuse(model, x, feats[model.layer], trans_matrices.to(x.device),
                                                       num_agent_tensor, batch_size, model.training, inference)
        res = heads(model, decoder(model.decoder, *feats))
        res["prob_action"], res["coef"] = prob, coef
        return res
    if hasattr(model, "FUSE_MODE"):                 # Sum / Mean / Max / Cat fusion, DiscoNet (no teacher)
        feats = encoder(model.u_encoder, x)
        fuse = disco_fuse if hasattr(model, "pixel_weighted_fusion") else simple_fuse
        feats[model.layer] = fuse(model, feats[model.layer], trans_matrices.to(x.device), num_agent_tensor, batch_size)
        return heads(model, decoder(model.decoder, *feats))
    raise NotImplementedError("training graph exists for FaFNet, V2VNet, When2com and the FusionBase family")
