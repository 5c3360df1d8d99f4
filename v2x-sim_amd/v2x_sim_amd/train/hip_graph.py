"""The detection backbone's TRAINING graph on the hand-written HIP kernels, bf16 NHWC end to end.

`train/graph.py` is upstream's graph on PyTorch-ROCm ops (MIOpen convolutions and BatchNorm over fp32 NCHW, autograd).  This module
is the same graph -- Backbone.py's lidar_encoder / lidar_decoder and DetModelBase.py's heads: conv -> batch-statistics BN -> ReLU,
nearest x2 upsample + concat -- over the SAME parameter tree, with

    3x3 convolutions   forward           v2x_conv2d (the inference kernels, bias in the epilogue, no activation)
                       data gradient     v2x_conv2d on W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]; stride 2: on dy with zeros inserted
                       weight gradient   v2x_conv3x3_wgrad (conv_wgrad.hip; stride 2: the same zero-inserted dy)
    BatchNorm + ReLU   forward/backward  v2x_bn_train_forward / v2x_bn_train_backward (bn_train.hip), running statistics updated
                                         exactly as nn.BatchNorm does

    1x1 layers         forward, dgrad    v2x_conv2d (ksize 1: conv3d_1/2, the heads' last layers with fp32 logits)
                       weight gradient   the centre tap of v2x_conv3x3_wgrad

and the maps staying bf16 NHWC between layers (no layout or precision conversion).  Left on PyTorch-ROCm ops, by design:
upsample/concat and their backward (views, copies and a 2x2 sum), bias gradients (a reduction), the cross-agent fusion of V2VNet
(converted to the fp32 graph at the fusion layer and back) and the loss.  conv4_2's 16x16 map is widened to 16x32 with zero
columns (exact: see conv3x3); maps that do not tile even then fall back to MIOpen.  No atomics anywhere in a FaFNet step.
Packed weights are rebuilt on the GPU after every optimizer step (packing.on_device): nothing crosses PCIe inside a step.

Mixed precision: activations, activation gradients and the MFMA operands are bf16; every sum, the BN statistics, the weight
gradients and the master weights are fp32.  Enabled with V2X_TRAIN_HIP=1 (FaFModule.step / train_forward); there is no CPU path.
"""
import types
import weakref

import torch
import torch.nn.functional as F

from .. import ops, packing, tuning

BF16 = torch.bfloat16


# ------------------------------------------------------------------ packed weights, rebuilt when the parameter changed
_CACHE = {}


def _conv_like(weight, bias, stride):
    k = weight.shape[-1]
    return types.SimpleNamespace(weight=weight, bias=bias, kernel_size=(k, k), stride=(stride, stride), padding=(k // 2, k // 2),
                                 out_channels=weight.shape[0])


def _versions(weight, bias):
    return (packing.param_version(weight), None if bias is None else packing.param_version(bias))


def _layer(kind, weight, bias, stride, cin_pad):
    # keyed on the parameter OBJECT (weak reference checked on a hit): a data_ptr can be reused by another model's tensor
    key = (kind, id(weight), stride, cin_pad)
    ver = _versions(weight, bias)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is weight:
        return hit[1]
    # one device launch per packing (v2x_pack_conv_device); the data gradient = a stride-1 convolution with the flipped, transposed weights
    layer = packing.pack_conv_device("train." + kind, weight, bias, stride=stride, cin_pad=cin_pad, dgrad=kind == "dgrad")
    if len(_CACHE) > 512:
        _CACHE.clear()
        _PLANS.clear()
    _CACHE[key] = (ver, layer, weakref.ref(weight), None if bias is None else weakref.ref(bias))
    return layer


# After an optimizer step EVERY cached packing is stale (~55 per model: forward and data-gradient layers).  Re-packed lazily, one launch each, that
# was 47-55 launches of ~5 us on the launch floor (0.23-0.27 ms of a 6-ms FaFNet step).  train_forward calls _repack_stale() first: the stale
# packings whose parameters are still alive and in place are rebuilt IN PLACE by one launch (packing.RepackPlan: the job table lives on the
# device, keyed by the set of cache entries) and their cache entries marked fresh; anything else takes the lazy path above.  Bit-identical.
_PLANS = {}


def _entry_packs(obj):
    if isinstance(obj, ops.PackedConv):
        return [obj]
    return ([obj.halo] if obj.halo is not None else []) + list(obj.fallback)


def _repack_stale():
    if tuning.get("TRAIN_PACK_BATCH") == 0 or not _CACHE:
        return
    stale, dead = [], []
    for key, ent in _CACHE.items():
        w = ent[2]()
        b = None if ent[3] is None else ent[3]()
        if w is None or (ent[3] is not None and b is None):
            dead.append(key)                           # the parameter is gone (a discarded model): its packed buffers go with it
            continue
        ver = _versions(w, b)
        if ver != ent[0]:
            stale.append((key, ent, ver))
    if dead:
        #  only the weights are weakly referenced: without this the packings and the job tables of discarded models stayed resident
        # until the 512-entry / 16-plan overflow clears.  A plan is keyed by the cache keys it re-packs: drop those that name a dead entry.
        gone = set(dead)
        for key in dead:
            del _CACHE[key]
        for pkey in [k for k in _PLANS if gone.intersection(k) or not _PLANS[k].valid()]:
            del _PLANS[pkey]
    if len(stale) < 2:
        return
    pkey = tuple(k for k, _, _ in stale)
    plan = _PLANS.get(pkey)
    if plan is None or not plan.valid():
        if torch.cuda.is_current_stream_capturing():
            return                                     # the job table cannot be uploaded inside a capture: lazy path (the warm-up steps build the plan)
        packs = [pc for _, ent, _ in stale for pc in _entry_packs(ent[1])]
        if any(pc.repack is None for pc in packs):
            return
        if len(_PLANS) > 16:
            _PLANS.clear()
        plan = _PLANS[pkey] = packing.RepackPlan(packs)
        if not plan.valid():
            return
    plan.launch()
    for key, ent, ver in stale:
        _CACHE[key] = (ver,) + ent[1:]


# (Round 4, measured and removed: the weight-gradient launches on a side stream forked when dy exists and joined at the end of the backward pass.
#  Bit-identical -- once the 1x1 layers' centre-tap slices were made contiguous ON the side stream: AccumulateGrad clones a gradient that has not
#  the parameter's layout at once, on the backward stream -- but a captured step replays 0.3 ms SLOWER with the fork (FaFNet 5.77 -> 6.07 ms,
#  V2VNet 7.18 -> 7.53 at 10 maps: one hipGraph does not run its branches side by side here), and the eager step is host-bound.
#  profiles/r04_train_switch_ab.txt; the code is at commit "Training: batched re-pack ..." minus one.)


def _zero_insert(dy):
    """dy (N, Ho, Wo, C) of a stride-2 layer -> (N, 2Ho, 2Wo, C) with dy at the even positions."""
    N, Ho, Wo, C = dy.shape
    if dy.dtype == BF16 and C % 8 == 0 and dy.is_cuda:
        return ops.zero_insert(dy.contiguous())            # one pass (v2x_zero_insert_bf16) instead of a fill and a strided copy
    z = torch.zeros((N, 2 * Ho, 2 * Wo, C), dtype=dy.dtype, device=dy.device)
    z[:, ::2, ::2] = dy
    return z


def _layer_runs(cin_p, cout, stride, H, W):
    """Will ops.run_layer find a kernel for the ONE packing packing.pack_conv_device makes for this layer (it packs no gather fallback beside
    a halo / streamed layout)?  Mirrors run_layer's conditions exactly (a miss there is an IndexError on the empty fallback list)."""
    layout = packing.train_layout(cin_p, cout, stride)
    if layout == 0:
        return True                                    # the gather kernel takes any extent
    if stride == 2:
        return (H % 8 == 0 and W % 64 == 0) or (H % 16 == 0 and W % 32 == 0)
    return ops.halo_eligible(H, W, layout, cin_p) and (layout != 2 or H * W >= 256)


def hip_eligible(weight, stride, H, W, cin_pad=None):
    """3x3, stride 1 or 2, pad 1, Cout % 32 == 0 (Cin is padded to 32), and every launch of the layer's forward AND backward has a kernel at
    this extent: the forward layer, the data-gradient layer (a stride-1 convolution Cout -> Cin' at the input extent) and the weight-gradient
    kernel (H % 8 == 0, W % 32 == 0).  Anything else takes the F.conv2d path of conv3x3()."""
    cout, cin, kh, kw = weight.shape
    if not (kh == 3 and kw == 3 and stride in (1, 2) and cout % 32 == 0 and H % 8 == 0 and W % 32 == 0):
        return False
    cin_p = cin_pad if cin_pad is not None else (cin + 31) // 32 * 32
    if cin_p % 32:
        return False
    return _layer_runs(cin_p, cout, stride, H, W) and _layer_runs(cout, cin_p, 1, H, W)


def _run_layer(layer, x):
    """ops.run_layer as a DECLARED small-batch launch (TRAIN_SPLITK, round 6): at 10 maps the deep layers of a training step are 40-160 workgroups on 256
    CUs, each walking all of its input chunks; inside ops.latency_dispatch() the streamed layers below the gate split their chunk ranges over 2-6
    workgroups per tile (ops.small_batch_splitk: partial sums added in range order by one more launch -- fixed order, deterministic).  A training step has
    no batch-invariance to keep (batch statistics), so the batch-dependent rule is free here.  Target 480 workgroups / gate 300 tiles from a paired sweep
    (FaFNet 4.40 -> 4.22 ms at 10 maps, 6.50 -> 6.38 at 20; at 40 maps no layer qualifies)."""
    sk = tuning.get("TRAIN_SPLITK")
    if sk != 0:
        with ops.latency_dispatch(target=sk if sk > 1 else 0):
            return ops.run_layer(layer, x)
    return ops.run_layer(layer, x)


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        cin = weight.shape[1]
        cin_pad = x.shape[-1] if x.shape[-1] != cin else None          # conv_pre_1: 13 channels stored as 32
        y = _run_layer(_layer("fwd", weight, bias, stride, cin_pad), x)
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.has_bias = stride, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy_in, dy = dy, dy.contiguous()
        dyz = _zero_insert(dy) if ctx.stride == 2 else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _run_layer(_layer("dgrad", weight, None, 1, None), dyz)
        if ctx.needs_input_grad[1]:
            dw = ops.conv3x3_wgrad(x, dyz, cin_out=weight.shape[1])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _attached_channel_sum(dy_in)
            if db is None:
                db = ops.channel_sum(dy)
        return dx, dw, db, None


def _attached_channel_sum(dy):
    """The per-channel sum the BN backward kernel accumulated while it wrote this gradient (_BnRelu.backward attaches it to the tensor it
    returns; autograd hands that same tensor object to the producing convolution's backward when the convolution's output has one
    consumer).  None when the gradient did not come from there, or was modified since: the caller then reduces it itself."""
    tag = getattr(dy, "_v2x_chsum", None)
    if tag is None or tag[0] != dy._version:
        return None
    if tag[1] is _EXACT_ZERO:
        return _zero_slice(dy.shape[-1], dy.device)
    return tag[1] if tag[1].shape[0] == dy.shape[-1] else None


_EXACT_ZERO = object()      # _BnRelu.backward's tag for "this gradient's channel sums are exactly zero"
_ZERO_ARENA = {}            # device -> [fp32 zeros, cursor]
_ZERO_ARENA_FLOATS = 1 << 15


def _zero_slice(n, device):
    """n fp32 zeros as a FRESH view into a per-device zero buffer.  autograd's AccumulateGrad keeps an incoming gradient without a copy only when nothing
    else references the tensor object: a shared zero vector (rounds 6's first form) was cloned per layer -- 23 device-to-device copies of 128 B ... 2 KiB
    in a FaFNet step, 3.6 us each (tools/train_op_census.py).  A view created here, at the point of use, is kept as it is: `conv.bias.grad` then aliases the
    buffer.  The buffer is only ever READ by this package and by the optimizers; should user code write into a bias gradient in place (weight decay
    added by hand), train_forward re-zeroes the buffer at the start of the next step (one fill), before any slice of it is handed out again."""
    key = str(device)
    st = _ZERO_ARENA.get(key)
    if st is None:
        st = _ZERO_ARENA[key] = [torch.zeros((_ZERO_ARENA_FLOATS,), dtype=torch.float32, device=device), 0]
    span = (n + 63) // 64 * 64        # 256-byte steps: the fused optimizers' vector paths want 16-byte-aligned gradients
    if span > _ZERO_ARENA_FLOATS:
        return torch.zeros((n,), dtype=torch.float32, device=device)
    if st[1] + span > _ZERO_ARENA_FLOATS:
        st[1] = 0                     # wrapped: slices may then alias each other -- all of them are zero
    z = st[0][st[1]:st[1] + n]
    st[1] += span
    return z


def _rezero_arena():
    for st in _ZERO_ARENA.values():
        st[0].zero_()
        st[1] = 0


class _BnRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        y, mean, invstd = ops.bn_train_forward(x, gamma.detach(), beta.detach(), running_mean, running_var, eps, momentum, relu)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        if tuning.get("TRAIN_BN_BIAS_ZERO") != 0:
            # Round 6.  The bias of a convolution in front of a batch-statistics BatchNorm has a gradient that is IDENTICALLY ZERO: dx = a (g - mean(g) -
            # xhat mean(g xhat)) sums to a (sum g - sum g - mean(g xhat) sum xhat) = 0 over the batch, and the bias cannot move the BN's output at all.  What
            # autograd (and this kernel's dx_sum form) returns for it is the rounding residue of a cancelling sum of ~1e5-1e6 terms.  Returning the exact
            # value saves the sum's accumulation in the dx pass and one finish launch per layer (22 launches, ~0.2 ms of a FaFNet step at any batch size);
            # TRAIN_BN_BIAS_ZERO = 0 restores the computed residue (tests/test_gpu_train_kernels.py compares the two).
            dx, dgamma, dbeta = ops.bn_train_backward(x, dy.contiguous(), gamma.detach(), beta.detach(), mean, invstd, ctx.relu)
            dsum = _EXACT_ZERO
        else:
            dx, dgamma, dbeta, dsum = ops.bn_train_backward(x, dy.contiguous(), gamma.detach(), beta.detach(), mean, invstd, ctx.relu, dx_sum=True)
        dx._v2x_chsum = (dx._version, dsum)      # the bias gradient of the convolution in front of this BN (_attached_channel_sum)
        return dx, dgamma, dbeta, None, None, None, None, None


def conv3x3(x, conv):
    """nn.Conv2d (3x3) on a bf16 NHWC map: the HIP kernels where the map tiles, MIOpen through a layout round trip otherwise."""
    s = conv.stride[0]
    if hip_eligible(conv.weight, s, x.shape[1], x.shape[2], x.shape[3]):
        return _Conv3x3.apply(x.contiguous(), conv.weight, conv.bias, s)
    if s == 1 and x.shape[2] % 32 == 16 and hip_eligible(conv.weight, s, x.shape[1], x.shape[2] + 16, x.shape[3]):
        # a 16-pixel-wide map (conv4_2): widen it with 16 zero columns -- a zero column IS the layer's padding, so columns 0..W-1 of
        # the output, of dx and the whole of dW are unchanged (the slice's backward zero-fills dy over the extra columns)
        W = x.shape[2]
        return _Conv3x3.apply(F.pad(x, (0, 0, 0, 16)).contiguous(), conv.weight, conv.bias, s)[:, :, :W]
    y = F.conv2d(x.permute(0, 3, 1, 2).float(), conv.weight, conv.bias, conv.stride, conv.padding)
    return y.permute(0, 2, 3, 1).to(BF16).contiguous()


_PENDING_COUNTERS = []
_DEFER_COUNTERS = [False]      # set by train_forward for the duration of a forward pass


def _flush_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


def bn_relu(x, bn, relu=True):
    """nn.BatchNorm2d / nn.BatchNorm3d in train mode (+ ReLU) on a bf16 NHWC map."""
    if not bn.training:
        raise RuntimeError("hip_graph is the TRAINING graph (batch statistics); evaluation runs the inference engine")
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if _DEFER_COUNTERS[0]:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)  # train_forward bumps them all in ONE launch (22 tiny `+= 1` kernels per step otherwise)
        else:
            bn.num_batches_tracked += 1
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm (momentum=None)")
    return _BnRelu.apply(x.contiguous(), bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, relu)


def cbr(x, conv, bn):
    return bn_relu(conv3x3(x, conv), bn)


def _layer_1x1(kind, weight, bias, f32_out, cout_pad):
    key = (kind, id(weight), f32_out, cout_pad)
    ver = (packing.param_version(weight), None if bias is None else packing.param_version(bias))
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is weight:
        return hit[1]
    # one device launch per packing (v2x_pack_conv_device, gather layout); dgrad: dx = dy . W, the gradient's channels zero-padded to cout_pad
    pc = packing.pack_conv1x1_device("train." + kind + "1x1", weight, bias if kind == "fwd" else None, dgrad=kind != "fwd", cout_pad=cout_pad,
                                     f32_out=f32_out)
    _CACHE[key] = (ver, pc, weakref.ref(weight), None if bias is None else weakref.ref(bias))
    return pc


class _Conv1x1(torch.autograd.Function):
    """1x1 layer on the HIP kernels: forward and data gradient on the implicit-GEMM kernel (v2x_conv2d, ksize 1); the weight gradient
    is the CENTRE TAP of v2x_conv3x3_wgrad (8/9 of its MFMAs are wasted, and it is still 5-10x faster than the fp32 library GEMM
    with K = N*H*W that autograd would run: 58 us against 0.4-1.1 ms per layer at 10 maps)."""

    @staticmethod
    def forward(ctx, x, weight, bias, f32_out):
        y = ops.conv2d(_layer_1x1("fwd", weight, bias, f32_out, 0), x)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        cout = weight.shape[0]
        cp = (cout + 31) // 32 * 32
        db = dyp = None
        if dy.dtype == torch.float32 and tuning.get("TRAIN_HEAD_PACK") != 0:
            # the heads' logit gradients arrive fp32 from the loss: bf16 + channel padding + the bias gradient in one pass (v2x_cast_pad_chsum_f32, round 6)
            # instead of pad (fill + copy), cast, and a separate fp32 reduction -- six launches and four passes over the gradients per head
            packed = ops.cast_pad_chsum(dy.contiguous(), cp)
            if packed is not None:
                dyp, db = packed
        if dyp is None:
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = _attached_channel_sum(dy)
                if db is None:
                    db = dy.float().sum((0, 1, 2))
            dyp = F.pad(dy, (0, cp - cout)).to(BF16).contiguous() if (cp != cout or dy.dtype != BF16) else dy.contiguous()
        if not (ctx.has_bias and ctx.needs_input_grad[2]):
            db = None
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d(_layer_1x1("dgrad", weight, None, False, cp), dyp)
        if ctx.needs_input_grad[1]:
            dw = ops.conv3x3_wgrad(x, dyp)[:cout, :, 1, 1].reshape(weight.shape)
        return dx, dw, db, None


def conv1x1(x, weight, bias, f32_out=False):
    """1x1 layer (nn.Conv2d 1x1 or upstream's Conv3D 1x1x1 on a length-1 sequence) on a bf16 NHWC map."""
    N, H, W, cin = x.shape
    if H % 8 == 0 and W % 32 == 0 and cin % 32 == 0:
        return _Conv1x1.apply(x.contiguous(), weight, bias, f32_out)
    y = F.linear(x.float(), weight.reshape(weight.shape[0], weight.shape[1]), bias)      # odd extent: the library GEMM
    return y if f32_out else y.to(BF16)


def conv3d_1x1(x, m):
    """upstream Conv3D on a length-1 sequence: 1x1x1 conv3d + BatchNorm3d + ReLU."""
    return bn_relu(conv1x1(x, m.conv3d.weight, m.conv3d.bias), m.bn3d)


class _UpCat(torch.autograd.Function):
    """cat(nearest x2 upsample of lo, skip) along the channels, NHWC bf16: one launch forward (v2x_upcat_bf16), one backward (the 2x2 sums of the
    upsampled part + the skip part's slice: v2x_upcat_bwd_bf16) instead of an expand-copy + cat and two slice copies + a bf16 reduction."""

    @staticmethod
    def forward(ctx, lo, skip):
        ctx.c0 = lo.shape[3]
        return ops.upcat(lo.contiguous(), skip.contiguous())

    @staticmethod
    def backward(ctx, dcat):
        return ops.upcat_backward(dcat.contiguous(), ctx.c0)


def _layer_halo(kind, weight, bias, c_up, rows):
    """Cached halo packings of the decoder's full-resolution `_1` layer (packing.pack_conv_device_halo): kind "up8.fwd" | "up8.dA" | "up8.dB"."""
    key = (kind, id(weight), c_up, rows)
    ver = _versions(weight, bias)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is weight:
        return hit[1]
    pc = packing.pack_conv_device_halo("train." + kind, weight, bias, c_up=c_up, dgrad_rows=rows)
    _CACHE[key] = (ver, pc, weakref.ref(weight), None if bias is None else weakref.ref(bias))
    return pc


class _UpCatConv3x3(torch.autograd.Function):
    """conv3x3(cat(up(lo), skip)) for the decoder's full-resolution layer (conv8_1: 64 half-resolution + 32 skip channels -> 32 at 256 x 256).  Until round 6
    this layer -- and only this one -- ran on the gather kernel, forward (96 -> 32) AND data gradient (32 -> 96): 0.44 + 0.49 ms of an 11.9-ms FaFNet step at 40
    maps (no halo kernel fits 96 input or output channels; the streamed kernels want >= 64 output rows).  Now:
      forward        the inference kernel of the 9-tap form, reading lo and skip in place (conv3x3_halo_pp_kernel<64, 32, 32>);
      data gradient  two launches into one 96-channel map: 32 -> 64 (conv3x3_halo_kernel<0, 32, 64>) and 32 -> 32 (conv3x3_halo_sb_kernel), packed from row
                     slices of the transposed weights; then the usual upsample / concat backward (2 x 2 sums + slice);
      weight gradient unchanged (v2x_conv3x3_wgrad on the materialised 96-channel map, which is built for it)."""

    @staticmethod
    def forward(ctx, lo, skip, weight, bias):
        c_up = lo.shape[3]
        cat = ops.upcat(lo, skip)                       # (the weight gradient contracts over it)
        y = ops.conv2d(_layer_halo("up8.fwd", weight, bias, c_up, None), lo, skip)
        ctx.save_for_backward(cat, weight)
        ctx.c_up, ctx.has_bias = c_up, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        cat, weight = ctx.saved_tensors
        dy_in, dy = dy, dy.contiguous()
        c_up, cin = ctx.c_up, weight.shape[1]
        d_lo = d_skip = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dcat = torch.empty_like(cat)
            ops.conv2d(_layer_halo("up8.dA", weight, None, c_up, (0, c_up)), dy, out=dcat, out_coff=0)
            ops.conv2d(_layer_halo("up8.dB", weight, None, c_up, (c_up, cin - c_up)), dy, out=dcat, out_coff=c_up)
            d_lo, d_skip = ops.upcat_backward(dcat, c_up)
        if ctx.needs_input_grad[2]:
            dw = ops.conv3x3_wgrad(cat, dy, cin_out=cin)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = _attached_channel_sum(dy_in)
            if db is None:
                db = ops.channel_sum(dy)
        return d_lo, d_skip, dw, db


def upcat_conv3x3(lo, skip, conv):
    """conv(cat(up(lo), skip)) -- the fused form where the halo kernels cover the shape (64 + 32 -> 32, H % 8 == 0, W % 32 == 0), else conv3x3(upcat(...))."""
    N, H, W, C = lo.shape
    if (tuning.get("TRAIN_UPCAT_CONV") != 0 and (C, skip.shape[3], conv.weight.shape[0]) == (64, 32, 32) and conv.stride[0] == 1
            and lo.dtype == BF16 and skip.dtype == BF16 and skip.shape[:3] == (N, 2 * H, 2 * W) and (2 * H) % 8 == 0 and (2 * W) % 32 == 0
            and conv.weight.shape[1] == C + skip.shape[3] and conv.weight.is_cuda and conv.weight.dtype == torch.float32):
        return _UpCatConv3x3.apply(lo.contiguous(), skip.contiguous(), conv.weight, conv.bias)
    return conv3x3(upcat(lo, skip), conv)


def upcat(lo, skip):
    """cat(nearest x2 upsample of lo, skip) along the channels, NHWC."""
    N, H, W, C = lo.shape
    if (lo.dtype == BF16 and skip.dtype == BF16 and C % 8 == 0 and skip.shape[3] % 8 == 0
            and skip.shape[:3] == (N, 2 * H, 2 * W)):
        return _UpCat.apply(lo, skip)
    up = lo[:, :, None, :, None, :].expand(N, H, 2, W, 2, C).reshape(N, 2 * H, 2 * W, C)   # backward = a 2x2 sum (no atomics: deterministic)
    return torch.cat((up, skip), dim=3)


# ------------------------------------------------------------------ the graph (Backbone.py)
def nhwc_input(bevs):
    """bevs (A*B, 1, X, Y, Z) dense occupancy -> (A*B, X, Y, 32) bf16 (the Z heights are the channels; zero-padded to 32)."""
    x = bevs[:, 0]
    if x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and x.shape[-1] <= 32:
        return ops.dense_to_nhwc(x, c_pad=32)               # one launch (v2x_dense_f32_to_nhwc_bf16) instead of pad + cast (+ copy)
    return F.pad(x, (0, 32 - x.shape[-1])).to(BF16).contiguous()


def encoder(e, x):
    x = cbr(x, e.conv_pre_1, e.bn_pre_1)
    x = cbr(x, e.conv_pre_2, e.bn_pre_2)
    x_1 = cbr(x, e.conv1_1, e.bn1_1)
    x_1 = conv3d_1x1(cbr(x_1, e.conv1_2, e.bn1_2), e.conv3d_1)
    x_2 = cbr(x_1, e.conv2_1, e.bn2_1)
    x_2 = conv3d_1x1(cbr(x_2, e.conv2_2, e.bn2_2), e.conv3d_2)
    x_3 = cbr(cbr(x_2, e.conv3_1, e.bn3_1), e.conv3_2, e.bn3_2)
    x_4 = cbr(cbr(x_3, e.conv4_1, e.bn4_1), e.conv4_2, e.bn4_2)
    return [x, x_1, x_2, x_3, x_4]


def decoder(d, x, x_1, x_2, x_3, x_4):
    y = cbr(cbr(upcat(x_4, x_3), d.conv5_1, d.bn5_1), d.conv5_2, d.bn5_2)
    y = cbr(cbr(upcat(y, x_2), d.conv6_1, d.bn6_1), d.conv6_2, d.bn6_2)
    y = cbr(cbr(upcat(y, x_1), d.conv7_1, d.bn7_1), d.conv7_2, d.bn7_2)
    return cbr(bn_relu(upcat_conv3x3(y, x, d.conv8_1), d.bn8_1), d.conv8_2, d.bn8_2)


def heads(model, x):
    c, bp = model.classification, model.regression.box_prediction
    cls = conv1x1(cbr(x, c.conv1, c.bn1), c.conv2.weight, c.conv2.bias, f32_out=True)           # fp32 NHWC logits
    loc = conv1x1(cbr(x, bp[0], bp[1]), bp[3].weight, bp[3].bias, f32_out=True)
    return {"cls": cls.reshape(cls.shape[0], -1, model.category_num),
            "loc": loc.reshape(-1, loc.size(1), loc.size(2), model.anchor_num_per_loc, model.out_seq_len, model.box_code_size)}


def _gru_conv_hip(x, weight, bias):
    """The ConvGRU's input convolution (2C -> 3C, 3x3: 5.4 GMAC per map -- the heaviest layer of a V2VNet step) forward, data gradient and
    weight gradient on the hand-written kernels: fp32 NCHW in / out for the fp32 fusion graph around it (warp, gate arithmetic), bf16 NHWC
    inside like every other layer of this graph.  Falls back to F.conv2d where the map does not tile."""
    N, Cc, H, W = x.shape
    if not hip_eligible(weight, 1, H, W) or Cc % 32:
        return F.conv2d(x, weight, bias, 1, 1)
    # (layout + precision in ONE copy each way: .to(dtype) alone keeps the permuted strides and a second copy would follow)
    y = _Conv3x3.apply(x.permute(0, 2, 3, 1).to(dtype=BF16, memory_format=torch.contiguous_format), weight, bias, 1)
    return y.permute(0, 3, 1, 2).to(dtype=torch.float32, memory_format=torch.contiguous_format)


class _AffineSample(torch.autograd.Function):
    """grid_sample(x, affine_grid(theta)) (bilinear, zeros, align_corners=False) of fp32 NCHW maps: forward on v2x_warp_affine_f32, the data
    gradient on its exact transpose v2x_warp_affine_bwd_f32 (a deterministic gather).  theta carries no gradient (poses are data)."""

    @staticmethod
    def forward(ctx, x, theta):
        theta = theta.detach().to(torch.float32).contiguous()
        ctx.save_for_backward(theta)
        return ops.warp_affine(x.contiguous(), theta)

    @staticmethod
    def backward(ctx, dy):
        theta, = ctx.saved_tensors
        return ops.warp_affine(dy.contiguous(), theta, backward=True), None


# ---------------------------------------------------------------------------------------------- V2VNet's fusion stage on bf16 NHWC (round 6)
class _V2VMessage(torch.autograd.Function):
    """cur (N, H, W, C) [, base] -> conv_in (M, H, W, 2C) = [ego map | mean over the neighbours of the twice-warped base maps] (csrc/v2v_train.hip)."""

    @staticmethod
    def forward(ctx, cur, base, trans, plan):
        ctx.plan, ctx.N, ctx.two = plan, cur.shape[0], base is not None
        ctx.save_for_backward(trans)
        return ops.v2v_message(cur.contiguous(), None if base is None else base.contiguous(), trans, plan)

    @staticmethod
    def backward(ctx, d):
        trans, = ctx.saved_tensors
        dbase, dcur = ops.v2v_message_backward(d.contiguous(), trans, ctx.plan, ctx.N, ctx.two)
        return (dcur, dbase, None, None) if ctx.two else (dbase, None, None, None)


class _GruGatesNhwc(torch.autograd.Function):
    """graph.py::_gru_step's gate arithmetic on the bf16 NHWC pre-activations of the input convolution; the backward also produces both bias gradients
    (the convolution's, attached to dgi for _Conv3x3.backward, and bias_hh's) from per-workgroup partial sums -- no reduction pass over dgi."""

    @staticmethod
    def forward(ctx, gi, bias_hh):
        bhh = bias_hh.detach().float().contiguous()
        ctx.save_for_backward(gi, bhh)
        return ops.gru_gates_nhwc(gi, bhh)

    @staticmethod
    def backward(ctx, dh):
        gi, bhh = ctx.saved_tensors
        dgi, sums = ops.gru_gates_nhwc_backward(gi, bhh, dh.contiguous())
        c3 = gi.shape[-1]
        dgi._v2x_chsum = (dgi._version, sums[:c3])
        return dgi, (sums[c3:] if ctx.needs_input_grad[1] else None)


def _v2v_plan(model, counts, items, rows, B, A, trans, N, device):
    """The int32 device tables of one (agent table, B): graph.py::v2v_fuse's pair enumeration (pairs of an ego item consecutive, neighbours in agent order)."""
    A1, A2 = trans.shape[1], trans.shape[2]
    key = (tuple(counts), B, A, A1, A2, str(device))
    cache = model.__dict__.setdefault("_v2v_nhwc_plan_cache", {})
    plan = cache.get(key)
    if plan is None:
        pairs = [(m, j * B + f, f, a, j) for m, (a, f) in enumerate(items) for j in range(counts[f]) if j != a]
        K = counts[0] - 1
        uses = {}
        for pi, p in enumerate(pairs):
            uses.setdefault(p[1], []).append(pi)
        ok = len(pairs) == len(items) * K and len(uses) == N and all(len(v) == K for v in uses.values())
        cache.clear()
        if not ok:
            plan = cache[key] = False
        else:
            i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)  # noqa: E731
            item_of_row = [-1] * N
            for m, r in enumerate(rows):
                item_of_row[r] = m
            plan = cache[key] = {"M": len(items), "K": K, "src": i32([p[1] for p in pairs]), "tsel": i32([(f * A1 + a) * A2 + j for (_, _, f, a, j) in pairs]),
                                 "rows": i32(rows), "inv": i32([uses[r] for r in range(N)]), "item_of_row": i32(item_of_row),
                                 "identity": list(rows) == list(range(N)), "rows_long": torch.tensor(rows, device=device)}
    return plan or None


def v2v_fuse_nhwc(model, feat, trans, num_agent_tensor, B):
    """V2VNet's message-passing rounds on the bf16 NHWC fusion-layer maps feat (A*B, H, W, C): per round a message launch, the ConvGRU's input convolution and
    a gates launch (TRAIN_V2V_NHWC 1).  -> the updated maps, or None when this form does not cover the batch (a frame with fewer agents than another, a
    channel count or extent the kernels do not tile): the caller then runs graph.v2v_fuse on the fp32 graph."""
    if tuning.get("TRAIN_V2V_NHWC") == 0 or feat.dtype != BF16 or trans is None:
        return None
    g = model.convgru
    N, H, W, C = feat.shape
    A = model.agent_num
    counts, items, rows = model.frame_plan(num_agent_tensor, B, A)
    if min(counts) < 2:
        raise RuntimeError("V2VNet needs >= 2 agents in every frame (stack expects a non-empty TensorList)")
    w = g.weight_ih_l0
    if min(counts) != max(counts) or tuple(w.shape) != (3 * C, 2 * C, 3, 3) or C % 32 or not hip_eligible(w, 1, H, W) or not ops.gru_gates_nhwc_ok(N * H * W, C) \
            or g.bias_ih_l0 is None or g.bias_hh_l0 is None:
        return None
    plan = _v2v_plan(model, counts, items, rows, B, A, trans, N, feat.device)
    if plan is None:
        return None
    T = trans.detach().to(torch.float32).contiguous()
    cur = feat
    for _ in range(model.gnn_rounds()):
        base = cur if model.neighbor_source == "updated" else feat
        conv_in = _V2VMessage.apply(cur, None if base is cur else base, T, plan)
        h = _GruGatesNhwc.apply(_Conv3x3.apply(conv_in, w, g.bias_ih_l0, 1), g.bias_hh_l0)
        cur = h if plan["identity"] else cur.index_copy(0, plan["rows_long"], h)
    return cur


def _v2v_stage(model, feat, T, num_agent_tensor, batch_size):
    from . import graph
    out = v2v_fuse_nhwc(model, feat, T, num_agent_tensor, batch_size)
    if out is None:      # warp + gate arithmetic on the fp32 graph, the GRU's convolution on the kernels (rounds 3-5's form)
        out, = _fused_on_fp32_graph(graph.v2v_fuse, model, feat, T, num_agent_tensor, batch_size, _gru_conv_hip)
    return out


def _fused_on_fp32_graph(fuse, model, feat, *args):
    """The cross-agent fusion runs on the fp32 NCHW graph (train/graph.py): convert the fusion-layer maps, fuse, convert back.  The warp
    inside it (graph.warp_batch) runs on the hand-written kernels, forward and backward (WARP_HIP=0: F.grid_sample and its atomic backward)."""
    from . import graph
    prev = graph._affine_sample_override
    graph._affine_sample_override = _AffineSample.apply if tuning.get("WARP_HIP") != 0 else None
    try:
        out = fuse(model, feat.permute(0, 3, 1, 2).to(dtype=torch.float32, memory_format=torch.contiguous_format), *args)
    finally:
        graph._affine_sample_override = prev
    extra = ()
    if isinstance(out, tuple):
        out, extra = out[0], out[1:]
    return (out.permute(0, 2, 3, 1).to(dtype=BF16, memory_format=torch.contiguous_format),) + tuple(extra)


def train_forward(model, bevs, trans_matrices=None, num_agent_tensor=None, batch_size=1, inference="softmax"):
    """See _train_forward; the BatchNorm layers' num_batches_tracked counters are bumped together, in one launch, when the forward is complete."""
    prev, _DEFER_COUNTERS[0] = _DEFER_COUNTERS[0], True
    _repack_stale()
    _rezero_arena()
    try:
        return _train_forward(model, bevs, trans_matrices, num_agent_tensor, batch_size, inference)
    finally:
        _DEFER_COUNTERS[0] = prev
        if not prev:
            _flush_counters()


def _train_forward(model, bevs, trans_matrices=None, num_agent_tensor=None, batch_size=1, inference="softmax"):
    """bevs (A*B, 1, X, Y, Z) on the MI355X -> {'loc', 'cls'} fp32 (segmentation variants: NHWC fp32 logits), shapes as
    train/graph.py::train_forward.  Every detection / segmentation baseline: encoder, decoder and heads on the kernels; the cross-agent
    fusion (V2VNet's warp + ConvGRU, when2com's handshake, sum / mean / max / cat / DiscoNet) on the fp32 graph at the fusion layer.
    model.training must be on."""
    from . import graph
    if not bevs.is_cuda:
        raise RuntimeError("train/hip_graph.py runs on the MI355X (no CPU path)")
    x = nhwc_input(bevs)
    T = None if trans_matrices is None else trans_matrices.to(x.device)
    if hasattr(model, "outc"):                      # segmentation variants: det backbone + 1x1 head, NHWC fp32 logits
        if hasattr(model, "stpn"):
            y = decoder(model.stpn.decoder, *encoder(model.stpn.encoder, x))
        else:
            feats = encoder(model.u_encoder, x)
            feats[model.layer] = _v2v_stage(model, feats[model.layer], T, num_agent_tensor, batch_size)
            y = decoder(model.decoder, *feats)
        return conv1x1(y, model.outc.conv.weight, model.outc.conv.bias, f32_out=True)
    if hasattr(model, "stpn"):                      # FaFNet: lowerbound / upperbound
        return heads(model, decoder(model.stpn.decoder, *encoder(model.stpn.encoder, x)))
    feats = encoder(model.u_encoder, x)
    res_extra = {}
    if hasattr(model, "convgru"):                   # V2VNet: the message-passing rounds on the kernels (v2v_fuse_nhwc); a ragged batch on the fp32 graph
        feats[model.layer] = _v2v_stage(model, feats[model.layer], T, num_agent_tensor, batch_size)
    elif hasattr(model, "query_key_net"):           # when2com / who2com: its key / query tower reads the input on the fp32 graph
        x32 = bevs[:, 0].permute(0, 3, 1, 2).to(torch.float32)
        fused, prob, coef = _fused_on_fp32_graph(lambda m, f: graph.when2com_fuse(m, x32, f, T, num_agent_tensor, batch_size, model.training, inference),
                                                 model, feats[model.layer])
        feats[model.layer] = fused
        res_extra = {"prob_action": prob, "coef": coef}
    elif hasattr(model, "FUSE_MODE"):               # Sum / Mean / Max / Cat fusion, DiscoNet (no teacher)
        fuse = graph.disco_fuse if hasattr(model, "pixel_weighted_fusion") else graph.simple_fuse
        feats[model.layer], = _fused_on_fp32_graph(fuse, model, feats[model.layer], T, num_agent_tensor, batch_size)
    else:
        raise NotImplementedError("hip_graph: unknown model family")
    res = heads(model, decoder(model.decoder, *feats))
    res.update(res_extra)
    return res
