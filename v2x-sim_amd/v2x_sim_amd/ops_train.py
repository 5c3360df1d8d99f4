"""Row f-3 (training): python wrappers over the C ABI's backward / statistics entries -- weight gradients, BatchNorm in training mode, the detection
loss, the ConvGRU gates, the affine warp and the upsample + concat with their adjoints.  Re-exported by ops.py (`ops.bn_train_forward` ...)."""
import torch

from . import _lib
from ._launch import _Prof, _dev, _dev_opt, _stream  # noqa: F401


def conv3x3_wgrad(x, dy, cin_out=None):
    """Weight gradient of a 3x3 stride-1 pad-1 conv.  x (N, H, W, Cin), dy (N, H, W, Cout) bf16 NHWC -> dW (Cout, cin_out or Cin, 3, 3) fp32
    (the parameter's own layout; cin_out < Cin: the input was stored zero-padded).  MFMA kernel contracting over pixels + a fixed-order
    sum of the per-block partials written straight in OIHW order (v2x_conv3x3_wgrad_reduce)."""
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    if tuple(dy.shape[:3]) != (N, H, W):
        raise ValueError("x %s and dy %s disagree" % (tuple(x.shape), tuple(dy.shape)))
    ns = lib.v2x_conv3x3_wgrad_splits(N, H, W, Cin, Cout)
    if ns == 0:
        raise ValueError("v2x_conv3x3_wgrad needs H % 8 == 0, W % 32 == 0, Cin % 32 == 0 and Cout % 32 == 0")
    ws = torch.empty((ns, Cout, 3, 3, Cin), dtype=torch.float32, device=x.device)
    prof = _Prof("conv3x3_wgrad_kernel", 2.0 * N * H * W * Cout * 9 * Cin, (x.numel() + dy.numel()) * 2 + ws.numel() * 4)
    rc = lib.v2x_conv3x3_wgrad(_dev(x, torch.bfloat16, "x"), _dev(dy, torch.bfloat16, "dy"), N, H, W, Cin, Cout,
                               _dev(ws, torch.float32, "workspace"), ns, _stream())
    prof.done()
    _lib.check(rc, "v2x_conv3x3_wgrad")
    cin_out = Cin if cin_out is None else cin_out
    dw = torch.empty((Cout, cin_out, 3, 3), dtype=torch.float32, device=x.device)
    _lib.check(lib.v2x_conv3x3_wgrad_reduce(_dev(ws, torch.float32, "workspace"), ns, Cout, Cin, cin_out, _dev(dw, torch.float32, "dw"), _stream()),
               "v2x_conv3x3_wgrad_reduce")
    return dw


def gru_gates(gi, bias_hh):
    """v2x_gru_gates_f32: gi (P, 3C, H, W) fp32 contiguous, bias_hh (3C,) fp32 -> h (P, C, H, W) fp32 (h0 = 0: h = n - z n)."""
    lib = _lib.load()
    P, C3, H, W = gi.shape
    h = torch.empty((P, C3 // 3, H, W), dtype=torch.float32, device=gi.device)
    _lib.check(lib.v2x_gru_gates_f32(_dev(gi, torch.float32, "gi"), _dev(bias_hh, torch.float32, "bias_hh"), P, C3 // 3, H * W,
                                     _dev(h, torch.float32, "h"), _stream()), "v2x_gru_gates_f32")
    return h


def gru_gates_backward(gi, bias_hh, dh):
    """v2x_gru_gates_bwd_f32: -> (dgi like gi, dn_r (P, C, H, W)): d bias_hh = cat(dgi[:, :2C].sum((0, 2, 3)), dn_r.sum((0, 2, 3)))."""
    lib = _lib.load()
    P, C3, H, W = gi.shape
    dgi = torch.empty_like(gi)
    dn_r = torch.empty((P, C3 // 3, H, W), dtype=torch.float32, device=gi.device)
    _lib.check(lib.v2x_gru_gates_bwd_f32(_dev(gi, torch.float32, "gi"), _dev(bias_hh, torch.float32, "bias_hh"), _dev(dh, torch.float32, "dh"),
                                         P, C3 // 3, H * W, _dev(dgi, torch.float32, "dgi"), _dev(dn_r, torch.float32, "dn_r"), _stream()),
               "v2x_gru_gates_bwd_f32")
    return dgi, dn_r


def v2v_message(cur, base, trans, plan):
    """v2x_v2v_message_bf16: cur, base (N, H, W, C) bf16 NHWC (base None: the same maps), trans fp32 (..., 4, 4) contiguous, plan = the int32 device tables of
    train/hip_graph.py::_v2v_plan -> conv_in (M, H, W, 2C) bf16 = [cur[rows[m]] | mean over item m's K neighbours of the twice-warped base maps]."""
    lib = _lib.load()
    N, H, W, Cc = cur.shape
    M, K = plan["M"], plan["K"]
    out = torch.empty((M, H, W, 2 * Cc), dtype=torch.bfloat16, device=cur.device)
    b = cur if base is None else base
    _lib.check(lib.v2x_v2v_message_bf16(_dev(cur, torch.bfloat16, "cur"), _dev(b, torch.bfloat16, "base"), _dev(trans, torch.float32, "trans"),
                                        _dev(plan["src"], torch.int32, "src"), _dev(plan["tsel"], torch.int32, "tsel"), _dev(plan["rows"], torch.int32, "rows"),
                                        M, K, N, Cc, H, W, _dev(out, torch.bfloat16, "conv_in"), _stream()), "v2x_v2v_message_bf16")
    return out


def v2v_message_backward(dconv_in, trans, plan, N, separate_cur):
    """v2x_v2v_message_bwd_bf16: dconv_in (M, H, W, 2C) bf16 -> dbase (N, H, W, C) bf16 (the exact transpose of v2v_message w.r.t. base), with the ego half
    added in (separate_cur False: base and cur were the same maps) or returned beside it as dcur."""
    lib = _lib.load()
    M, H, W, C2 = dconv_in.shape
    Cc = C2 // 2
    dbase = torch.empty((N, H, W, Cc), dtype=torch.bfloat16, device=dconv_in.device)
    dcur = torch.empty_like(dbase) if separate_cur else None
    _lib.check(lib.v2x_v2v_message_bwd_bf16(_dev(dconv_in, torch.bfloat16, "dconv_in"), _dev(trans, torch.float32, "trans"), _dev(plan["inv"], torch.int32, "inv"),
                                            _dev(plan["tsel"], torch.int32, "tsel"), _dev(plan["item_of_row"], torch.int32, "item_of_row"), M, plan["K"], N, Cc, H, W,
                                            _dev(dbase, torch.bfloat16, "dbase"), None if dcur is None else _dev(dcur, torch.bfloat16, "dcur"), _stream()),
               "v2x_v2v_message_bwd_bf16")
    return dbase, dcur


def gru_gates_nhwc_ok(P, Cc):
    return _lib.load().v2x_gru_gates_nhwc_workspace_size(P, Cc) > 0


def gru_gates_nhwc(gi, bias_hh):
    """v2x_gru_gates_nhwc_bf16: gi (..., 3C) bf16 NHWC, bias_hh (3C,) fp32 -> h (..., C) bf16 (h0 = 0: h = n - z n)."""
    lib = _lib.load()
    C3 = gi.shape[-1]
    P = gi.numel() // C3
    h = torch.empty(tuple(gi.shape[:-1]) + (C3 // 3,), dtype=torch.bfloat16, device=gi.device)
    _lib.check(lib.v2x_gru_gates_nhwc_bf16(_dev(gi, torch.bfloat16, "gi"), _dev(bias_hh, torch.float32, "bias_hh"), P, C3 // 3, _dev(h, torch.bfloat16, "h"), _stream()),
               "v2x_gru_gates_nhwc_bf16")
    return h


def gru_gates_nhwc_backward(gi, bias_hh, dh):
    """v2x_gru_gates_nhwc_bwd_bf16: -> (dgi like gi, sums (6C,) fp32 = channel sums of dgi as stored (3C: d bias_ih) | d bias_hh (3C)); fixed order."""
    lib = _lib.load()
    C3 = gi.shape[-1]
    Cc = C3 // 3
    P = gi.numel() // C3
    nbytes = lib.v2x_gru_gates_nhwc_workspace_size(P, Cc)
    if nbytes == 0:
        raise ValueError("gru_gates_nhwc_backward: C / 8 must divide 256, got C=%d" % Cc)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=gi.device)
    dgi = torch.empty_like(gi)
    sums = torch.empty((6 * Cc,), dtype=torch.float32, device=gi.device)
    _lib.check(lib.v2x_gru_gates_nhwc_bwd_bf16(_dev(gi, torch.bfloat16, "gi"), _dev(bias_hh, torch.float32, "bias_hh"), _dev(dh, torch.bfloat16, "dh"), P, Cc,
                                               _dev(dgi, torch.bfloat16, "dgi"), _dev(sums, torch.float32, "sums6c"), _dev(ws, torch.float32, "workspace"), _stream()),
               "v2x_gru_gates_nhwc_bwd_bf16")
    return dgi, sums


def det_loss_forward(cls, labels, loc, targets, mask, alpha, beta):
    """v2x_det_loss_forward: fp32 contiguous device tensors cls / labels (n, 2), loc / targets (n, 6), mask (n,) bool or uint8 ->
    out4 (4,) fp32 = (loss, cls_loss, loc_loss, n_pos clamped to >= 1)."""
    lib = _lib.load()
    n = cls.numel() // 2
    ws = torch.empty((lib.v2x_det_loss_workspace_size(n) // 4,), dtype=torch.float32, device=cls.device)
    out = torch.empty((4,), dtype=torch.float32, device=cls.device)
    m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    _lib.check(lib.v2x_det_loss_forward(_dev(cls, torch.float32, "cls"), _dev(labels, torch.float32, "labels"), _dev(loc, torch.float32, "loc"),
                                        _dev(targets, torch.float32, "targets"), _dev(m8, torch.uint8, "mask"), n, alpha, beta,
                                        _dev(out, torch.float32, "out4"), _dev(ws, torch.float32, "workspace"), _stream()), "v2x_det_loss_forward")
    return out


def det_loss_backward(cls, labels, loc, targets, mask, alpha, beta, out4, g_loss, g_cls, g_loc):
    """v2x_det_loss_backward: -> (dcls like cls, dloc like loc) for the incoming gradients of (loss, cls_loss, loc_loss) (fp32 device scalars or None)."""
    lib = _lib.load()
    n = cls.numel() // 2
    dcls, dloc = torch.empty_like(cls), torch.empty_like(loc)
    m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    gs = [None if g is None else _dev(g, torch.float32, "grad") for g in (g_loss, g_cls, g_loc)]
    _lib.check(lib.v2x_det_loss_backward(_dev(cls, torch.float32, "cls"), _dev(labels, torch.float32, "labels"), _dev(loc, torch.float32, "loc"),
                                         _dev(targets, torch.float32, "targets"), _dev(m8, torch.uint8, "mask"), n, alpha, beta,
                                         _dev(out4, torch.float32, "out4"), gs[0], gs[1], gs[2], _dev(dcls, torch.float32, "dcls"),
                                         _dev(dloc, torch.float32, "dloc"), _stream()), "v2x_det_loss_backward")
    return dcls, dloc


def channel_sum(x):
    """x (..., C) bf16 NHWC -> (C,) fp32 = the sum over every other axis, in a fixed order (the bias gradient of a convolution)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    nbytes = lib.v2x_channel_sum_workspace_size(M, Cc)
    if nbytes == 0:
        return x.float().reshape(M, Cc).sum(0)          # channel counts the kernel does not tile (C / 8 must divide 256)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    _lib.check(lib.v2x_channel_sum_bf16(_dev(x, torch.bfloat16, "x"), M, Cc, _dev(out, torch.float32, "out"), _dev(ws, torch.float32, "workspace"),
                                        _stream()), "v2x_channel_sum_bf16")
    return out


def cast_pad_chsum(x, c_pad):
    """x (..., C) fp32 -> (bf16 (..., c_pad) with channels C.. zero, (C,) fp32 per-channel sums of x) in ONE pass + a finish launch (v2x_cast_pad_chsum_f32):
    the logit gradients of a 1x1 head made ready for its data- and weight-gradient kernels, with the bias gradient on the side.  None when the shape is not
    one the kernel takes (the caller falls back to pad / cast / sum)."""
    lib = _lib.load()
    Cc = x.shape[-1]
    M = x.numel() // Cc
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or Cc % 4 or c_pad < Cc:
        return None
    nbytes = lib.v2x_cast_pad_chsum_workspace_size(M, c_pad)
    if nbytes == 0:
        return None
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    out = torch.empty(x.shape[:-1] + (c_pad,), dtype=torch.bfloat16, device=x.device)
    sums = torch.empty((c_pad,), dtype=torch.float32, device=x.device)
    _lib.check(lib.v2x_cast_pad_chsum_f32(_dev(x, torch.float32, "x"), M, Cc, c_pad, _dev(out, torch.bfloat16, "out"), _dev(sums, torch.float32, "sums"),
                                          _dev(ws, torch.float32, "workspace"), _stream()), "v2x_cast_pad_chsum_f32")
    return out, sums[:Cc]


def warp_affine(x, theta, backward=False):
    """F.grid_sample(x, F.affine_grid(theta, x.shape, align_corners=False), "bilinear", "zeros", align_corners=False) on the HIP kernel
    (warp_train.hip), or -- backward=True -- its exact transpose applied to an output gradient x (deterministic gather).
    x (P, C, H, W) fp32 contiguous, theta (P, 2, 3) fp32 on the same device -> (P, C, H, W) fp32."""
    lib = _lib.load()
    P, Cc, H, W = x.shape
    if theta.shape != (P, 2, 3):
        raise ValueError("warp_affine: theta must be (%d, 2, 3), got %s" % (P, tuple(theta.shape)))
    theta = theta.contiguous()
    out = torch.empty_like(x)
    fn = lib.v2x_warp_affine_bwd_f32 if backward else lib.v2x_warp_affine_f32
    _lib.check(fn(_dev(x, torch.float32, "x"), _dev(theta, torch.float32, "theta"), P, Cc, H, W, _dev(out, torch.float32, "out"), _stream()),
               "v2x_warp_affine_bwd_f32" if backward else "v2x_warp_affine_f32")
    return out


def upcat(lo, skip):
    """cat(nearest x2 upsample of lo, skip) along the channels: lo (N, H, W, C0), skip (N, 2H, 2W, C1) bf16 NHWC -> (N, 2H, 2W, C0 + C1)."""
    lib = _lib.load()
    N, H, W, C0 = lo.shape
    if skip.shape[:3] != (N, 2 * H, 2 * W):
        raise ValueError("upcat: skip %s does not match twice the extent of lo %s" % (tuple(skip.shape), tuple(lo.shape)))
    C1 = skip.shape[3]
    out = torch.empty((N, 2 * H, 2 * W, C0 + C1), dtype=torch.bfloat16, device=lo.device)
    _lib.check(lib.v2x_upcat_bf16(_dev(lo, torch.bfloat16, "lo"), _dev(skip, torch.bfloat16, "skip"), N, H, W, C0, C1, _dev(out, torch.bfloat16, "out"),
                                  _stream()), "v2x_upcat_bf16")
    return out


def upcat_backward(dcat, C0):
    """Backward of upcat: dcat (N, 2H, 2W, C0 + C1) bf16 -> (d_lo (N, H, W, C0) = the 2x2 sums, d_skip (N, 2H, 2W, C1))."""
    lib = _lib.load()
    N, H2, W2, Ct = dcat.shape
    H, W, C1 = H2 // 2, W2 // 2, Ct - C0
    d_lo = torch.empty((N, H, W, C0), dtype=torch.bfloat16, device=dcat.device)
    d_skip = torch.empty((N, H2, W2, C1), dtype=torch.bfloat16, device=dcat.device)
    _lib.check(lib.v2x_upcat_bwd_bf16(_dev(dcat, torch.bfloat16, "dcat"), N, H, W, C0, C1, _dev(d_lo, torch.bfloat16, "d_lo"),
                                      _dev(d_skip, torch.bfloat16, "d_skip"), _stream()), "v2x_upcat_bwd_bf16")
    return d_lo, d_skip


def zero_insert(dy):
    """dy (N, Ho, Wo, C) bf16 of a stride-2 layer -> (N, 2Ho, 2Wo, C) with dy at the even positions, zeros elsewhere (one launch)."""
    lib = _lib.load()
    N, Ho, Wo, Cc = dy.shape
    out = torch.empty((N, 2 * Ho, 2 * Wo, Cc), dtype=torch.bfloat16, device=dy.device)
    _lib.check(lib.v2x_zero_insert_bf16(_dev(dy, torch.bfloat16, "dy"), N, Ho, Wo, Cc, _dev(out, torch.bfloat16, "out"), _stream()), "v2x_zero_insert_bf16")
    return out


def bn_train_forward(x, gamma, beta, running_mean, running_var, eps, momentum, relu=True):
    """Batch-statistics BN (+ ReLU) of a bf16 NHWC map on the HIP kernels (bn_train.hip).  x (..., C) bf16; gamma / beta (C,) fp32;
    running_mean / running_var fp32 (updated in place) or None.  -> (y bf16 like x, save_mean, save_invstd)."""
    lib = _lib.load()
    C = x.shape[-1]
    M = x.numel() // C
    nbytes = lib.v2x_bn_train_workspace_size(M, C)
    if nbytes == 0:
        raise ValueError("v2x_bn_train_forward: unsupported shape M=%d C=%d (C / 8 must divide 256)" % (M, C))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    prof = _Prof("bn_train_forward", 0.0, x.numel() * 2 * 3)
    rc = lib.v2x_bn_train_forward(_dev(x, torch.bfloat16, "x"), M, C, _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"),
                                  float(eps), float(momentum), _dev_opt(running_mean, torch.float32, "running_mean"),
                                  _dev_opt(running_var, torch.float32, "running_var"), 1 if relu else 0, _dev(y, torch.bfloat16, "y"),
                                  _dev(mean, torch.float32, "save_mean"), _dev(invstd, torch.float32, "save_invstd"),
                                  _dev(ws, torch.float32, "workspace"), _stream())
    prof.done()
    _lib.check(rc, "v2x_bn_train_forward")
    return y, mean, invstd


def bn_train_backward(x, dy, gamma, beta, mean, invstd, relu=True, dx_sum=False):
    """Backward of bn_train_forward: -> (dx bf16 like x, dgamma (C,), dbeta (C,)) fp32; dx_sum=True: also the per-channel sum of dx as stored
    (fp32 (C,): the bias gradient of the convolution that produced x), accumulated by the kernel that writes dx."""
    lib = _lib.load()
    C = x.shape[-1]
    M = x.numel() // C
    nbytes = lib.v2x_bn_train_workspace_size(M, C)
    if nbytes == 0:
        raise ValueError("v2x_bn_train_backward: unsupported shape M=%d C=%d" % (M, C))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    prof = _Prof("bn_train_backward", 0.0, x.numel() * 2 * 5)
    args = [_dev(x, torch.bfloat16, "x"), _dev(dy, torch.bfloat16, "dy"), M, C, _dev(gamma, torch.float32, "gamma"),
            _dev(beta, torch.float32, "beta"), _dev(mean, torch.float32, "save_mean"), _dev(invstd, torch.float32, "save_invstd"),
            1 if relu else 0, _dev(dx, torch.bfloat16, "dx"), _dev(dgamma, torch.float32, "dgamma"), _dev(dbeta, torch.float32, "dbeta")]
    if dx_sum:
        ws2 = torch.empty(lib.v2x_bn_dxsum_workspace_size(M, C) // 4, dtype=torch.float32, device=x.device)
        dsum = torch.empty(C, dtype=torch.float32, device=x.device)
        rc = lib.v2x_bn_train_backward_dxsum(*args, _dev(dsum, torch.float32, "dx_sum"), _dev(ws, torch.float32, "workspace"),
                                             _dev(ws2, torch.float32, "sum_workspace"), _stream())
    else:
        rc = lib.v2x_bn_train_backward(*args, _dev(ws, torch.float32, "workspace"), _stream())
    prof.done()
    _lib.check(rc, "v2x_bn_train_backward")
    return (dx, dgamma, dbeta, dsum) if dx_sum else (dx, dgamma, dbeta)
