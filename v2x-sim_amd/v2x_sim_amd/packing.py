"""Host-side weight packing: torch parameters -> the device layouts libv2x_amd.so consumes.

  * conv weight [Cout, Cin, k, k] fp32  ->  bf16 [w_rows][w_kpad], k = (ky*k + kx)*Cin + c
    (K-contiguous rows = MFMA "A" operand), rows padded to the kernel's channel tile,
    K padded to a multiple of 64 (zero weights, so the padded taps contribute nothing);
  * eval-mode BatchNorm folded to an fp32 (scale, shift) pair applied AFTER the fp32
    accumulation:  y = acc*scale + shift,  scale = gamma/sqrt(var+eps),
    shift = beta + scale*(conv_bias - mean)   (DESIGN.md section 3.2);
  * ConvGRU (h0 = 0): W_ih rows regrouped as (r,z,n) triples of 16 hidden channels so one
    wave owns all three gates of its channels; biases packed float4 per hidden channel.
"""

import torch

from . import _lib, tuning
from ._lib import V2X_EPI_DET
from .ops import PackedConv, V2X_EPI_BF16, V2X_EPI_F32, V2X_EPI_GRU



# ---- "has this parameter changed?" for the packed-weight caches ---------------------------------------------------------------------------
# Tensor._version counts in-place writes made through autograd-visible ops, and that is what the caches keyed on.  FUSED optimizers
# (torch.optim.Adam(fused=True): one multi-tensor kernel) and hipGraph replays update the parameters WITHOUT bumping it: a cache keyed on
# _version alone keeps serving the packed weights of step 0 and nothing learns.  watch_optimizer() registers a post-step hook that stamps
# every parameter the optimizer owns; param_version() is what the caches compare.
def param_version(t):
    return (t._version, getattr(t, "_v2x_epoch", 0))


def note_params_changed(params):
    for q in params:
        q._v2x_epoch = getattr(q, "_v2x_epoch", 0) + 1


def stepped(opt):
    """Call after opt.step() when the optimizer could not be hooked (watch_optimizer found no register_step_post_hook): stamps its parameters.
    A no-op for a watched optimizer (its hook has already done it)."""
    if opt is not None and not opt.__dict__.get("_v2x_watched"):
        for g in opt.param_groups:
            note_params_changed(g["params"])


def watch_optimizer(opt):
    """Idempotent: after every opt.step() the packed-weight caches see the optimizer's parameters as changed."""
    if opt is None or opt.__dict__.get("_v2x_watched") or not hasattr(opt, "register_step_post_hook"):
        return opt
    from .train.optim import use_hip_adam
    use_hip_adam(opt)                    # a plain torch.optim.Adam on CUDA fp32 parameters steps on the library's kernel from here on (same state, same hooks)

    def _hook(o, *_a, **_k):
        for g in o.param_groups:
            note_params_changed(g["params"])
    opt.register_step_post_hook(_hook)
    opt.__dict__["_v2x_watched"] = True
    return opt


# The per-optimizer hook above needs the caller's cooperation (make_optimizer, FaFModule, SegModule, GraphedTrainStep install it).  A caller
# who builds torch.optim.Adam(fused=True) himself and calls train_forward / backward / opt.step() directly would train on the packed weights of
# step 0 -- silently.  torch.optim.optimizer.register_optimizer_step_post_hook is a GLOBAL post-step hook for every optimizer of
# the process: installed once, at import, it stamps the parameters of whatever optimizer stepped (one attribute write per parameter per step).
def _install_global_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:      # (older torch: the per-optimizer hook and packing.stepped remain the way)
        return False

    def _global_hook(o, *_a, **_k):
        if not o.__dict__.get("_v2x_watched"):          # (a watched optimizer has stamped its parameters already)
            for g in o.param_groups:
                note_params_changed(g["params"])
    register_optimizer_step_post_hook(_global_hook)
    return True


GLOBAL_OPTIMIZER_HOOK = _install_global_optimizer_hook()


def _ceil_to(x, m):
    return (x + m - 1) // m * m


# Where the re-ordering runs.  None: on the host (inference packs once per checkpoint).  A device: the training path re-packs every
# layer after every optimizer step, so the permutes / pads / casts run on the GPU and nothing crosses PCIe (train/hip_graph.py).
PACK_DEVICE = None


class on_device:
    """with packing.on_device(dev): ... -- pack_conv / pack_conv_halo / pack_conv_stream build their buffers on `dev`."""

    def __init__(self, device):
        self.device, self.prev = device, None

    def __enter__(self):
        global PACK_DEVICE
        self.prev, PACK_DEVICE = PACK_DEVICE, self.device

    def __exit__(self, *exc):
        global PACK_DEVICE
        PACK_DEVICE = self.prev


def _home(t):
    return t.detach().float().cpu() if PACK_DEVICE is None else t.detach().float().to(PACK_DEVICE)


def _zeros(*shape):
    return torch.zeros(shape, dtype=torch.float32, device="cpu" if PACK_DEVICE is None else PACK_DEVICE)


def fold_bn(conv_bias, bn, cout):
    """-> (scale, shift) fp32 on CPU for y = acc*scale + shift: eval-mode BatchNorm folded behind the fp32 accumulation,
    scale = gamma / sqrt(var + eps), shift = beta + scale * (conv_bias - mean).  ONE definition for every host: the arithmetic is the library's
    v2x_fold_bn (IEEE sqrtf / division, separately rounded product and sum) -- torch's vectorised CPU sqrt is not correctly rounded and not
    even the same from one CPU model to the next (1 ulp on ~1-15 % of the channels), which made a C host's packed parameters differ from this
    one's on the GPU box (tests/test_c_abi.py compares the two hosts bit for bit)."""
    if bn is None:
        scale = _zeros(cout) + 1.0
        shift = _home(conv_bias) if conv_bias is not None else _zeros(cout)
        return scale, shift
    import ctypes as C
    import numpy as np
    f = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())   # noqa: E731
    g, b, mu, var = f(bn.weight), f(bn.bias), f(bn.running_mean), f(bn.running_var)
    cb = f(conv_bias) if conv_bias is not None else np.zeros(cout, np.float32)
    scale, shift = np.empty(cout, np.float32), np.empty(cout, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _lib.check(_lib.load().v2x_fold_bn(cout, cout, p(cb), p(g), p(b), p(mu), p(var), C.c_float(bn.eps), p(scale), p(shift)), "v2x_fold_bn")
    return torch.from_numpy(scale), torch.from_numpy(shift)


def pack_conv(name, weight, scale, shift, *, stride=1, pad=None, C0=None, C1=0, up0=0, relu=True,
              epilogue=V2X_EPI_BF16, cin_pad=None, device="cuda"):
    """weight [Cout, Cin, k, k] (any float dtype, CPU or device) -> PackedConv on `device`."""
    lib = _lib.load()
    w = _home(weight)
    cout, cin, k, _ = w.shape
    if pad is None:
        pad = (k - 1) // 2
    cin_p = cin if cin_pad is None else cin_pad
    if C0 is None:
        C0 = cin_p
    if C0 + C1 != cin_p:
        raise ValueError("%s: C0+C1=%d != Cin=%d" % (name, C0 + C1, cin_p))
    w = w.permute(0, 2, 3, 1)  # [Cout, ky, kx, Cin]
    if cin_p != cin:
        w = torch.nn.functional.pad(w, (0, cin_p - cin))
    K = k * k * cin_p
    tile = lib.v2x_conv_tile_rows(cout, epilogue)
    rows, kpad = _ceil_to(cout, tile), _ceil_to(K, 64)
    wp = _zeros(rows, kpad)
    wp[:cout, :K] = w.reshape(cout, K)
    sc = _zeros(rows)
    sf = _zeros(rows)
    sc[:cout] = scale
    sf[:cout] = shift
    return PackedConv(name=name, weight=wp.to(torch.bfloat16).to(device).contiguous(), scale=sc.to(device),
                      shift=sf.to(device), C0=C0, C1=C1, Cout=cout, ksize=k, stride=stride, pad=pad, up0=up0,
                      epilogue=epilogue, relu=relu, w_rows=rows, w_kpad=kpad)


def pack_conv_bn(name, conv, bn, *, relu=True, epilogue=V2X_EPI_BF16, device="cuda", **kw):
    scale, shift = fold_bn(conv.bias, bn, conv.out_channels)
    w = conv.weight
    if w.dim() == 5:  # Conv3d 1x1x1 over a length-1 sequence
        w = w[:, :, 0]
    return pack_conv(name, w, scale, shift, stride=conv.stride[-1], pad=conv.padding[-1], relu=relu,
                     epilogue=epilogue, device=device, **kw)


def pack_linear(name, lin, *, relu, epilogue=V2X_EPI_BF16, col_perm=None, device="cuda"):
    """nn.Linear as a 1x1 conv on a 1x1 map.  col_perm re-orders input features (NCHW -> NHWC flatten)."""
    w = lin.weight.detach().float().cpu()
    if col_perm is not None:
        w = w[:, col_perm]
    scale, shift = fold_bn(lin.bias, None, w.shape[0])
    return pack_conv(name, w[:, :, None, None], scale, shift, stride=1, pad=0, relu=relu, epilogue=epilogue,
                     device=device)


def pack_heads(name, cls_conv1, cls_bn1, cls_conv2, reg_conv1, reg_bn1, reg_conv2, device="cuda"):
    """Fuse the two det heads: one 3x3 conv 32 -> 64 (cls | reg hidden), one block-diagonal 1x1
    conv 64 -> (n_cls + n_reg) writing cls and loc to two contiguous fp32 NHWC tensors."""
    s1, t1 = fold_bn(cls_conv1.bias, cls_bn1, cls_conv1.out_channels)
    s2, t2 = fold_bn(reg_conv1.bias, reg_bn1, reg_conv1.out_channels)
    w1 = torch.cat([cls_conv1.weight.detach().float().cpu(), reg_conv1.weight.detach().float().cpu()], 0)
    hidden = pack_conv(name + ".hidden", w1, torch.cat([s1, s2]), torch.cat([t1, t2]), stride=1, pad=1, relu=True,
                       device=device)
    hc, hr = cls_conv1.out_channels, reg_conv1.out_channels
    ncls, nreg = cls_conv2.out_channels, reg_conv2.out_channels
    w2 = torch.zeros((ncls + nreg, hc + hr, 1, 1), dtype=torch.float32)
    w2[:ncls, :hc] = cls_conv2.weight.detach().float().cpu()
    w2[ncls:, hc:] = reg_conv2.weight.detach().float().cpu()
    b2 = torch.cat([cls_conv2.bias.detach().float().cpu(), reg_conv2.bias.detach().float().cpu()])
    final = pack_conv(name + ".final", w2, torch.ones(ncls + nreg), b2, stride=1, pad=0, relu=False,
                      epilogue=V2X_EPI_F32, device=device)
    return hidden, final, ncls


_CONSTS = {}


def _const(n, value, device, n_real=None):
    key = (n, value, str(device), n_real)
    t = _CONSTS.get(key)
    if t is None:
        t = torch.full((n,), value, dtype=torch.float32, device=device)
        if n_real is not None and n_real < n:
            t[n_real:] = 0.0
        _CONSTS[key] = t
    return t


def train_layout(cin_p, cout, stride):
    """Which packed layout the training graph's single-source 3x3 layers take (the rule of layer_conv_bn): 1 = halo kernel, 2 = streamed
    kernels (stride 1 and 2), 0 = gather kernel."""
    if stride == 1 and (cin_p, cout) in ((32, 32), (64, 64), (64, 32)):   # (64 -> 32: the data gradient of conv1_1)
        return 1
    if stride == 1 and cin_p >= (64 if STREAM_64 else 128) and cin_p % 32 == 0 and cout % 64 == 0 and STREAM_KERNEL:
        return 2
    if stride == 2 and cin_p % 32 == 0 and cout % 64 == 0 and STREAM_KERNEL:
        return 2
    return 0


def _repack_info(spec, weight, w, dgrad, buf, shift, bias, cout):
    """What RepackPlan needs to rebuild this packing in place after the parameter changed; None when the packer read a converted COPY of the
    parameter (not fp32 / not contiguous: its pointer is not the parameter's)."""
    import weakref
    if w.data_ptr() != weight.data_ptr():
        return None
    copy_bias = bias is not None and shift.data_ptr() != bias.data_ptr()     # shift is a padded copy of the bias (else: the parameter itself)
    return {"spec": spec, "weight": weakref.ref(weight), "ptr": weight.data_ptr(), "transform": 1 if dgrad else 0, "buf": buf,
            "bias": weakref.ref(bias) if copy_bias else None, "shift": shift if copy_bias else None, "cout": cout}


class RepackPlan:
    """ONE launch that rebuilds the packed weights of many layers in place (v2x_pack_conv_device_batch) -- a training step re-packs every layer
    after its optimizer step (train/hip_graph.py), one small launch each before this.  Built once from the PackedConv objects of a step (their
    `repack` records); valid while every parameter is alive and where it was.  The job table is copied to the device at construction (not
    inside a stream capture); launch() itself issues the one kernel (+ one small copy per layer whose shift vector is a padded copy of the bias)."""

    def __init__(self, packs):
        import ctypes as C
        from ._lib import PackJob
        lib = _lib.load()
        infos = [pc.repack for pc in packs]
        if not infos or any(i is None for i in infos):
            raise ValueError("RepackPlan: every packing must come from pack_conv_device / pack_conv1x1_device on the parameter itself")
        self.packs = list(packs)
        jobs = (PackJob * len(infos))()
        begin = 0
        for k, i in enumerate(infos):
            nb = C.c_int64(0)
            _lib.check(lib.v2x_pack_conv_device_job(C.byref(i["spec"]), C.c_void_p(i["ptr"]), i["transform"], C.c_void_p(i["buf"].data_ptr()), begin,
                                                    C.byref(jobs[k]), C.byref(nb)), "v2x_pack_conv_device_job")
            begin += nb.value
        self.total_blocks = begin
        dev = infos[0]["buf"].device
        self.jobs = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.n_jobs = len(infos)

    def valid(self):
        for pc in self.packs:
            i = pc.repack
            w = i["weight"]()
            if w is None or w.data_ptr() != i["ptr"] or (i["bias"] is not None and i["bias"]() is None):
                return False
        return True

    def launch(self):
        import ctypes as C
        lib = _lib.load()
        _lib.check(lib.v2x_pack_conv_device_batch(C.c_void_p(self.jobs.data_ptr()), self.n_jobs, self.total_blocks,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "v2x_pack_conv_device_batch")
        for pc in self.packs:
            i = pc.repack
            if i["bias"] is not None:
                i["shift"][:i["cout"]].copy_(i["bias"]().detach())


def pack_conv_device(name, weight, bias, *, stride=1, cin_pad=None, dgrad=False):
    """One-launch device packing of a plain 3x3 layer for the training graph (v2x_pack_conv_device): weight = the fp32 parameter ON THE
    DEVICE, [Cout, Cin, 3, 3].  dgrad: the layer that computes the convolution's data gradient (stride-1 convolution of dy with the flipped,
    transposed weights, no bias).  -> ops.Layer carrying exactly ONE packing (the kernel family train_layout picks; the gather packing
    as the layer's `fallback`, the others as its `halo`), scale = 1, shift = bias."""
    import ctypes as C
    from ._lib import PackSpec
    from .ops import Layer
    lib = _lib.load()
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
        w = w.float().contiguous()
    co_w, ci_w = w.shape[0], w.shape[1]
    cout, cin = (ci_w, co_w) if dgrad else (co_w, ci_w)
    cin_p = cin if cin_pad is None else cin_pad
    layout = train_layout(cin_p, cout, 1 if dgrad else stride)
    spec = PackSpec(Cout=cout, Cin=cin, ksize=3, cin_pad=cin_p, w_layout=layout, epilogue=V2X_EPI_BF16, chain=0)
    rows, kpad = C.c_int32(0), C.c_int32(0)
    nbytes = lib.v2x_pack_conv_size(C.byref(spec), C.byref(rows), C.byref(kpad))
    if nbytes == 0:
        raise ValueError("pack_conv_device(%s): %s" % (name, lib.v2x_last_error().decode()))
    buf = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.v2x_pack_conv_device(C.byref(spec), C.c_void_p(w.data_ptr()), 1 if dgrad else 0, C.c_void_p(buf.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "v2x_pack_conv_device(%s)" % name)
    n_par = rows.value
    scale = _const(n_par, 1.0, w.device, cout)        # shared read-only vectors: no fill per packing (ones for the real rows, zeros for the padding)
    if bias is not None and not dgrad:
        if n_par == cout and bias.dtype == torch.float32:
            shift = bias.detach()                      # the parameter itself (the packing lives until its next update)
        else:
            shift = torch.zeros((n_par,), dtype=torch.float32, device=w.device)
            shift[:cout] = bias.detach().float()
    else:
        shift = _const(n_par, 0.0, w.device)
    pc = PackedConv(name=name, weight=buf, scale=scale, shift=shift, C0=cin_p, C1=0, Cout=cout, ksize=3, stride=1 if dgrad else stride, pad=1,
                    up0=0, epilogue=V2X_EPI_BF16, relu=False, w_rows=n_par, w_kpad=kpad.value, w_layout=layout or None, Cout2=0)
    pc.repack = _repack_info(spec, weight, w, dgrad, buf, shift, bias, cout)
    return Layer([pc], None, name=name) if layout == 0 else Layer([], pc, name=name)


def pack_conv_device_halo(name, weight, bias, *, c_up=0, dgrad_rows=None):
    """Round 6, the decoder's full-resolution `_1` layer (conv8_1) of the training graph on the resident-weights halo kernels (w_layout 1), one launch per
    packing like pack_conv_device.  weight = the fp32 parameter ON THE DEVICE, [Cout, Cin, 3, 3].
      * dgrad_rows=None: the FORWARD layer on cat(up(lo), skip) with the two sources read in place -- C0 = c_up channels of the half-resolution map (up0 = 1),
        C1 = Cin - c_up of the skip (conv_halo.hip: conv3x3_halo_pp_kernel<64, 32, 32>, the inference kernel of the 9-tap form); scale = 1, shift = bias.
      * dgrad_rows=(r0, n): the DATA-GRADIENT layer of the input channels r0 .. r0 + n - 1 alone (Cout -> n channels; v2x_pack_spec.src_rows / src_row0): the
        32 -> 96 data gradient has no halo kernel (conv_halo.hip: it spills), 32 -> 64 and 32 -> 32 into one 96-channel map have.
    -> PackedConv."""
    import ctypes as C
    from ._lib import PackSpec
    lib = _lib.load()
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
        w = w.float().contiguous()
    co_w, ci_w = w.shape[0], w.shape[1]
    if dgrad_rows is None:
        spec = PackSpec(Cout=co_w, Cin=ci_w, ksize=3, cin_pad=ci_w, w_layout=1, epilogue=V2X_EPI_BF16, chain=0)
        cout, c0, c1, up0 = co_w, c_up, ci_w - c_up, 1 if c_up else 0
    else:
        r0, n = dgrad_rows
        spec = PackSpec(Cout=n, Cin=co_w, ksize=3, cin_pad=co_w, w_layout=1, epilogue=V2X_EPI_BF16, chain=0, src_rows=ci_w, src_row0=r0)
        cout, c0, c1, up0 = n, co_w, 0, 0
    rows, kpad = C.c_int32(0), C.c_int32(0)
    nbytes = lib.v2x_pack_conv_size(C.byref(spec), C.byref(rows), C.byref(kpad))
    if nbytes == 0:
        raise ValueError("pack_conv_device_halo(%s): %s" % (name, lib.v2x_last_error().decode()))
    buf = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=w.device)
    dgrad = dgrad_rows is not None
    _lib.check(lib.v2x_pack_conv_device(C.byref(spec), C.c_void_p(w.data_ptr()), 1 if dgrad else 0, C.c_void_p(buf.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "v2x_pack_conv_device(%s)" % name)
    n_par = rows.value
    scale = _const(n_par, 1.0, w.device, cout)
    if bias is not None and not dgrad:
        if n_par == cout and bias.dtype == torch.float32:
            shift = bias.detach()
        else:
            shift = torch.zeros((n_par,), dtype=torch.float32, device=w.device)
            shift[:cout] = bias.detach().float()
    else:
        shift = _const(n_par, 0.0, w.device)
    pc = PackedConv(name=name, weight=buf, scale=scale, shift=shift, C0=c0, C1=c1, Cout=cout, ksize=3, stride=1, pad=1, up0=up0,
                    epilogue=V2X_EPI_BF16, relu=False, w_rows=n_par, w_kpad=kpad.value, w_layout=1, Cout2=0)
    pc.repack = _repack_info(spec, weight, w, dgrad, buf, shift, bias if not dgrad else None, cout)
    return pc


def pack_conv1x1_device(name, weight, bias, *, dgrad=False, cout_pad=0, f32_out=False):
    """One-launch device packing of a 1x1 layer for the training graph (v2x_pack_conv_device, gather layout): weight = the fp32 parameter ON THE
    DEVICE, [Cout, Cin, 1, 1] (or [Cout, Cin]).  dgrad: the layer dx = dy . W (the transposed weights; dy's channels zero-padded to cout_pad, no
    bias, bf16 output).  -> PackedConv with scale = 1, shift = bias -- the bytes of the torch-op packing it replaces (pad / transpose / cast /
    scatter into a zeroed buffer: ~8 small launches per packing, 8 packings per training step)."""
    import ctypes as C
    from ._lib import PackSpec
    from ._lib import V2X_EPI_F32
    lib = _lib.load()
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
        w = w.float().contiguous()
    co_w, ci_w = w.shape[0], w.shape[1]
    if dgrad:
        cout, cin, cin_p = ci_w, co_w, max(cout_pad, co_w)
    else:
        cout, cin, cin_p = co_w, ci_w, ci_w
    epi = V2X_EPI_F32 if (f32_out and not dgrad) else V2X_EPI_BF16
    spec = PackSpec(Cout=cout, Cin=cin, ksize=1, cin_pad=cin_p, w_layout=0, epilogue=epi, chain=0)
    rows, kpad = C.c_int32(0), C.c_int32(0)
    nbytes = lib.v2x_pack_conv_size(C.byref(spec), C.byref(rows), C.byref(kpad))
    if nbytes == 0:
        raise ValueError("pack_conv1x1_device(%s): %s" % (name, lib.v2x_last_error().decode()))
    buf = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.v2x_pack_conv_device(C.byref(spec), C.c_void_p(w.data_ptr()), 1 if dgrad else 0, C.c_void_p(buf.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "v2x_pack_conv_device(%s)" % name)
    n_par = rows.value
    scale = _const(n_par, 1.0, w.device, cout)
    if bias is not None and not dgrad:
        if n_par == cout and bias.dtype == torch.float32:
            shift = bias.detach()
        else:
            shift = torch.zeros((n_par,), dtype=torch.float32, device=w.device)
            shift[:cout] = bias.detach().float()
    else:
        shift = _const(n_par, 0.0, w.device)
    pc = PackedConv(name=name, weight=buf.view(n_par, kpad.value), scale=scale, shift=shift, C0=cin_p, C1=0, Cout=cout, ksize=1, stride=1, pad=0,
                    up0=0, epilogue=epi, relu=False, w_rows=n_par, w_kpad=kpad.value)
    pc.repack = _repack_info(spec, weight, w, dgrad, buf, shift, bias if not dgrad else None, cout)
    return pc


def det_row_order(n_anchor=6, n_cls=2, n_code=6):
    """Row order of the chained 1x1 of the fused DETECTION heads (include/v2x_amd.h, V2X_EPI_DET): packed row 16 t + 4 q + r, q < 3,
    belongs to anchors a0 = 2 q, a1 = 2 q + 1.  -> list of 64 entries: ("cls", anchor, class) | ("loc", anchor, code) | None (zero row)."""
    if (n_anchor, n_cls, n_code) != (6, 2, 6):
        raise ValueError("the fused detection heads cover 6 anchors x (2 class logits + 6 box codes)")
    rows = [None] * 64
    for q in range(3):
        a0, a1 = 2 * q, 2 * q + 1
        tiles = [[("cls", a0, 0), ("cls", a0, 1), ("cls", a1, 0), ("cls", a1, 1)],
                 [("loc", a0, c) for c in range(4)],
                 [("loc", a0, 4), ("loc", a0, 5), ("loc", a1, 0), ("loc", a1, 1)],
                 [("loc", a1, c) for c in range(2, 6)]]
        for t in range(4):
            for r in range(4):
                rows[16 * t + 4 * q + r] = tiles[t][r]
    return rows


def pack_heads_det(name, cls_conv1, cls_bn1, cls_conv2, reg_conv1, reg_bn1, reg_conv2, device="cuda"):
    """The two det heads with the score threshold fused into the epilogue (conv_halo.hip, V2X_EPI_DET): the 3x3 hidden layer 32 -> 64
    (cls | reg) chained with the block-diagonal 1x1 whose 48 rows are scattered into 64 in det order.  Same products and sums per logit
    as pack_heads' fused fp32 layer -- only the row a logit is computed in changes."""
    s1, t1 = fold_bn(cls_conv1.bias, cls_bn1, cls_conv1.out_channels)
    s2, t2 = fold_bn(reg_conv1.bias, reg_bn1, reg_conv1.out_channels)
    hc, hr = cls_conv1.out_channels, reg_conv1.out_channels
    if (hc, hr, cls_conv2.out_channels, reg_conv2.out_channels) != (32, 32, 12, 36):
        return None
    w1 = torch.cat([cls_conv1.weight.detach().float().cpu(), reg_conv1.weight.detach().float().cpu()], 0)
    wc, wr = cls_conv2.weight.detach().float().cpu().reshape(12, hc), reg_conv2.weight.detach().float().cpu().reshape(36, hr)
    bc, br = cls_conv2.bias.detach().float().cpu(), reg_conv2.bias.detach().float().cpu()
    w2, b2 = torch.zeros((64, hc + hr, 1, 1)), torch.zeros(64)
    for rho, what in enumerate(det_row_order()):
        if what is None:
            continue
        kind, a, c = what
        if kind == "cls":
            w2[rho, :hc, 0, 0], b2[rho] = wc[a * 2 + c], bc[a * 2 + c]
        else:
            w2[rho, hc:, 0, 0], b2[rho] = wr[a * 6 + c], br[a * 6 + c]
    return pack_conv_halo(name, w1, torch.cat([s1, s2]), torch.cat([t1, t2]), relu=True, chain=(w2, torch.ones(64), b2, False),
                          epilogue=V2X_EPI_DET, device=device)


def pack_gru(name, weight_ih, bias_ih, bias_hh, *, C0, C1, device="cuda"):
    """ConvGRU cell step with h0 = 0 (the only way upstream V2VNet calls it: convgru(x, None)).
    W_hh * 0 contributes exactly b_hh, so only W_ih is packed."""
    w = weight_ih.detach().float().cpu()
    three_h, cin, k, _ = w.shape
    hid = three_h // 3
    if hid % 32 != 0 or cin != C0 + C1:
        raise ValueError("%s: hidden %d must be a multiple of 32 and Cin %d == C0+C1" % (name, hid, cin))
    K = k * k * cin
    kpad = _ceil_to(K, 64)
    wk = w.permute(0, 2, 3, 1).reshape(three_h, K)
    groups = hid // 16
    wp = torch.zeros((groups * 48, kpad), dtype=torch.float32)
    # packed row g*48 + gate*16 + e  <-  gate row gate*hid + g*16 + e
    src = wk.view(3, groups, 16, K).permute(1, 0, 2, 3).reshape(groups * 48, K)
    wp[:, :K] = src
    bi, bh = bias_ih.detach().float().cpu().view(3, hid), bias_hh.detach().float().cpu().view(3, hid)
    bias4 = torch.stack([bi[0] + bh[0], bi[1] + bh[1], bi[2], bh[2]], dim=1).contiguous()  # [hid][4]
    return PackedConv(name=name, weight=wp.to(torch.bfloat16).to(device).contiguous(), scale=bias4.to(device),
                      shift=None, C0=C0, C1=C1, Cout=hid, ksize=k, stride=1, pad=(k - 1) // 2, up0=0,
                      epilogue=V2X_EPI_GRU, relu=False, w_rows=groups * 48, w_kpad=kpad)


# ------------------------------------------------------------------ halo-tile kernel layouts (conv_halo.hip)
def _chain_row_order(cout):
    """packed row rho = 16*i + 4*q + r  computes hidden channel  kappa = 32*(i>>1) + 8*q + 4*(i&1) + r."""
    rho = torch.arange(cout)
    i, q, r = rho >> 4, (rho >> 2) & 3, rho & 3
    return 32 * (i >> 1) + 8 * q + 4 * (i & 1) + r


def pack_conv_halo(name, weight, scale, shift, *, C0=None, C1=0, relu=True, cin_pad=None, chain=None,
                   epilogue=V2X_EPI_BF16, device="cuda"):
    """3x3 stride-1 conv for the halo kernel: weights k-slot-major [9*Cin/8][Cout][8].
    chain = (weight2 [Cout2, Cout, 1, 1], scale2, shift2, relu2): 1x1 conv fused in the epilogue; the
    hidden rows are then stored in the kernel's chain order (scale/shift stay in natural order)."""
    w = _home(weight)
    cout, cin, k, _ = w.shape
    if k != 3:
        raise ValueError("halo kernel is 3x3 only")
    cin_p = cin if cin_pad is None else cin_pad
    if C0 is None:
        C0 = cin_p
    if C0 + C1 != cin_p or cin_p % 32 or cout % 32:
        raise ValueError("%s: halo kernel needs C0+C1 == Cin (multiple of 32) and Cout %% 32 == 0" % name)
    w = w.permute(0, 2, 3, 1)
    if cin_p != cin:
        w = torch.nn.functional.pad(w, (0, cin_p - cin))
    K = 9 * cin_p
    wk = w.reshape(cout, K)
    if chain is not None:
        wk = wk[_chain_row_order(cout)]
    wp = wk.view(cout, K // 8, 8).permute(1, 0, 2).contiguous()  # [kslot][cout][8]
    pc = PackedConv(name=name, weight=wp.to(torch.bfloat16).to(device).contiguous(),
                    scale=scale.detach().float().to(device).contiguous(),
                    shift=shift.detach().float().to(device).contiguous(), C0=C0, C1=C1, Cout=cout, ksize=3,
                    stride=1, pad=1, up0=1 if C1 else 0, epilogue=epilogue, relu=relu, w_rows=cout, w_kpad=K,
                    w_layout=1, Cout2=0)
    if chain is not None:
        w2, s2, t2, relu2 = chain
        w2 = w2.detach().float().cpu().reshape(w2.shape[0], cout)
        c2 = w2.shape[0]
        c2p = _ceil_to(c2, 16)
        w2p = torch.zeros((c2p, cout), dtype=torch.float32)
        w2p[:c2] = w2
        s2p, t2p = torch.zeros(c2p), torch.zeros(c2p)
        s2p[:c2], t2p[:c2] = s2, t2
        pc.Cout2, pc.relu2 = c2, relu2
        pc.weight2 = w2p.to(torch.bfloat16).to(device).contiguous()
        pc.scale2, pc.shift2 = s2p.to(device), t2p.to(device)
    return pc


# taps of the 3x3 that fall on the same half-resolution source pixel, per output parity p and class tap t: PARITY_TAPS[p][t]
PARITY_TAPS = (((0,), (1, 2)), ((0, 1), (2,)))


def parity_class_weights(w_up):
    """The x2 nearest-upsampled half of a decoder layer as four 2x2-tap convolutions on the half-resolution map (conv_halo.hip,
    conv3x3_halo_ppc_kernel): w_up fp32 [Cout][C0][3][3] -> fp32 [4 classes (py, px)][4 taps (a, b)][Cout][C0] with
    W'[py][px][a][b] = sum over ky in PARITY_TAPS[py][a], kx in PARITY_TAPS[px][b] of W[ky][kx], added in fp32 in (ky, kx) ascending order
    (the C packer v2x_pack_conv adds in the same order: bit-identical buffers).  Exact in real arithmetic; the caller rounds to bf16 ONCE."""
    cout, c0 = w_up.shape[0], w_up.shape[1]
    out = torch.empty((4, 4, cout, c0), dtype=w_up.dtype, device=w_up.device)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    acc = None
                    for ky in PARITY_TAPS[py][a]:
                        for kx in PARITY_TAPS[px][b]:
                            acc = w_up[:, :, ky, kx].clone() if acc is None else acc + w_up[:, :, ky, kx]
                    out[py * 2 + px, a * 2 + b] = acc
    return out


def pack_conv_halo_parity(name, weight, scale, shift, *, C0, C1, relu=True, device="cuda"):
    """Decoder layer cat(up(in0) [C0], in1 [C1]) -> 3x3 in the parity-class form (w_layout 3): up half = [class][tap][C0 / 8][Cout][8] with the
    pre-summed weights above, skip half = [tap ky*3+kx][C1 / 8][Cout][8]; bf16, rounded once from the fp32 sums."""
    w = _home(weight)
    cout, cin, k, _ = w.shape
    if k != 3 or C0 + C1 != cin or C0 % 32 or C1 % 32 or not C0 or not C1 or cout % 32:
        raise ValueError("%s: the parity-class form needs a 3x3 layer on cat(up(C0), C1) with C0, C1, Cout multiples of 32" % name)
    up = parity_class_weights(w[:, :C0])                                            # [4][4][cout][C0]
    up = up.view(16, cout, C0 // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)     # [class*4 + tap][kslot][cout][8]
    sk = w[:, C0:].permute(2, 3, 0, 1).contiguous()                                 # [ky][kx][cout][C1]
    sk = sk.view(9, cout, C1 // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)      # [tap][kslot][cout][8]
    wp = torch.cat([up, sk])
    return PackedConv(name=name, weight=wp.to(torch.bfloat16).to(device).contiguous(),
                      scale=scale.detach().float().to(device).contiguous(), shift=shift.detach().float().to(device).contiguous(),
                      C0=C0, C1=C1, Cout=cout, ksize=3, stride=1, pad=1, up0=1, epilogue=V2X_EPI_BF16, relu=relu, w_rows=cout,
                      w_kpad=16 * C0 + 9 * C1, w_layout=3, Cout2=0)


def pack_conv_stream_parity(name, weight, scale, shift, *, C0, C1, relu=True, device="cuda"):
    """Decoder layer cat(up(in0) [C0], in1 [C1]) -> 3x3 with Cout % 128 == 0 in the parity-class form of the STREAMED kernel (w_layout 4,
    conv_stream_pc.hip: conv5_1, conv6_1).  Per 128-row channel tile: [C0 / 32][class tap 2a + b][class 2py + px][4 k-slots][128][8] with the
    pre-summed weights of parity_class_weights, then [C1 / 32][kx][ky][4 k-slots][128][8]; + 64 B of zeros (the kernel's zero page) at the end."""
    w = _home(weight)
    cout, cin, k, _ = w.shape
    if k != 3 or C0 + C1 != cin or C0 % 32 or C1 % 32 or not C0 or not C1 or (cout % 128 and cout != 64):
        raise ValueError("%s: the streamed parity-class form needs a 3x3 layer on cat(up(C0), C1), C0 and C1 multiples of 32, Cout of 128 (or 64)" % name)
    T = 64 if cout == 64 else 128           # rows per channel tile (64: the two-tiles-per-workgroup kernel of conv7_1's shape)
    nt, n_up, n_sk = cout // T, C0 // 32, C1 // 32
    up = parity_class_weights(w[:, :C0])                                                   # [cls][tap][cout][C0]
    up = up.view(4, 4, nt, T, n_up, 4, 8).permute(2, 4, 1, 0, 5, 3, 6).contiguous()        # [tile][chunk][tap][cls][k-slot][row][8]
    sk = w[:, C0:].reshape(nt, T, n_sk, 4, 8, 3, 3).permute(0, 2, 6, 5, 3, 1, 4).contiguous()     # [tile][chunk][kx][ky][k-slot][row][8]
    flat = torch.cat([up.view(nt, -1), sk.view(nt, -1)], 1).reshape(-1)
    flat = torch.cat([flat, torch.zeros(32, dtype=flat.dtype, device=flat.device)])
    return PackedConv(name=name, weight=flat.to(torch.bfloat16).to(device).contiguous(),
                      scale=scale.detach().float().to(device).contiguous(), shift=shift.detach().float().to(device).contiguous(),
                      C0=C0, C1=C1, Cout=cout, ksize=3, stride=1, pad=1, up0=1, epilogue=V2X_EPI_BF16, relu=relu, w_rows=cout,
                      w_kpad=16 * C0 + 9 * C1, w_layout=4, Cout2=0)


def layer_conv_bn(name, conv, bn, *, device, relu=True, halo=True, **kw):
    """conv+BN(+ReLU) as an ops.Layer: gather-kernel packing always, plus the packing of the patch-based kernel that
    covers the layer -- halo (3x3 stride 1, <= 96 input channels), streamed (3x3 stride 1, >= 128 input channels),
    or stride-2 streamed (3x3 stride 2, one source, Cin % 32 == 0, Cout % 64 == 0)."""
    from .ops import Layer
    fb = pack_conv_bn(name, conv, bn, relu=relu, device=device, **kw)
    h = None
    cin_p = fb.C0 + fb.C1
    k3 = conv.kernel_size[-1] == 3
    if halo and k3 and conv.stride[-1] == 1 and conv.out_channels % 32 == 0:
        scale, shift = fold_bn(conv.bias, bn, conv.out_channels)
        cin_h = _ceil_to(cin_p, 32)
        key = (fb.C0 if fb.C1 else 0, fb.C1 if fb.C1 else cin_h, conv.out_channels)
        if key == (64, 32, 32) and fb.up0 == 1 and tuning.get("PARITY_CLASS") != 0:   # conv8_1: parity-class form
            h = pack_conv_halo_parity(name, conv.weight, scale, shift, C0=fb.C0, C1=fb.C1, relu=relu, device=device)
        elif key in ((0, 32, 32), (64, 32, 32), (0, 64, 32)) or key == (0, 64, 64):   # (0, 64, 32): no model layer; the data gradient of conv1_1 (train_layout)
            h = pack_conv_halo(name, conv.weight, scale, shift, C0=fb.C0 if fb.C1 else cin_h, C1=fb.C1, relu=relu,
                               cin_pad=cin_h if not fb.C1 else None, device=device)
        elif (fb.C1 and fb.up0 == 1 and fb.C0 % 32 == 0 and fb.C1 % 32 == 0 and STREAM_KERNEL and
              ((conv.out_channels % 128 == 0 and tuning.get("PARITY_CLASS") >= 2) or (conv.out_channels == 64 and tuning.get("PARITY_CLASS") >= 3))):
            # conv5_1, conv6_1 (PARITY_CLASS >= 2) and conv7_1 (>= 3): streamed parity-class forms
            h = pack_conv_stream_parity(name, conv.weight, scale, shift, C0=fb.C0, C1=fb.C1, relu=relu, device=device)
            # declared latency launches (a handful of maps) split K over several workgroups per tile (ops.small_batch_splitk): that form exists for
            # the 9-tap streamed layout only, so the layer carries it as well
            lat = pack_conv_stream(name, conv.weight, scale, shift, C0=fb.C0, C1=fb.C1, up0=fb.up0, relu=relu, device=device)
            return Layer([fb], h, name=name, latency=lat)
        elif (cin_p >= (64 if STREAM_64 else 128) and fb.C0 % 32 == 0 and fb.C1 % 32 == 0 and conv.out_channels % 64 == 0
              and STREAM_KERNEL):
            h = pack_conv_stream(name, conv.weight, scale, shift, C0=fb.C0, C1=fb.C1, up0=fb.up0, relu=relu,
                                 device=device)
    elif (halo and STREAM_S2 and STREAM_KERNEL and k3 and conv.stride[-1] == 2 and not fb.C1 and cin_p % 32 == 0
          and conv.out_channels % 64 == 0):
        scale, shift = fold_bn(conv.bias, bn, conv.out_channels)
        h = pack_conv_stream(name, conv.weight, scale, shift, C0=cin_p, relu=relu, stride=2, device=device)
    return Layer([fb], h, name=name)


STREAM_KERNEL = True  # tools flip this to A/B the streamed-weights kernel against the gather kernel
CHAIN_STREAM = True   # conv1_2 -> conv3d_1 and conv2_2 -> conv3d_2: 1x1 chained in the streamed kernel's epilogue (False: separate launches for conv3d_2)
STREAM_64 = True      # 64 -> 64 layers (conv7_2): streamed (wide 4-wave) kernel instead of the resident-weights halo kernel (471 vs 495 us)
# (round 6: the switch PP_64 = 0 -- the 64 -> 64 full-resolution layers on the wide streamed kernel instead of the resident-weights ping-pong halo kernel -- was retired: measured
#  slower in round 3, exercised by no test or tool since)
STREAM_S2 = True      # stride-2 3x3 layers (conv1_1, conv2_1, conv3_1): patch-based stride-2 kernel instead of the gather kernel


# ------------------------------------------------------------------ streamed-weights kernel layout (conv_stream.hip)
def _stream_layout(wk, rows_tile, cin):
    """wk [rows][9*cin] with k = tap*cin + c  ->  [co_tile][chunk][tap][slot][row][8] (flattened)."""
    rows = wk.shape[0]
    n_tiles, n_chunks = rows // rows_tile, cin // 32
    w = wk.view(n_tiles, rows_tile, 9, n_chunks, 4, 8)            # [tile][row][tap][chunk][slot][8]
    flat = w.permute(0, 3, 2, 4, 1, 5).contiguous().view(-1)      # [tile][chunk][tap][slot][row][8]
    return torch.cat([flat, torch.zeros(32, dtype=flat.dtype, device=flat.device)])   # + 64 B of zeros: the kernel's zero page


def pack_conv_stream(name, weight, scale, shift, *, C0=None, C1=0, up0=0, relu=True, chain=None, stride=1, device="cuda"):
    """3x3 conv with C0, C1 multiples of 32 and Cout a multiple of 64, for conv_stream.hip (stride 1) or
    conv_stream_s2.hip (stride 2: one source, no chain) -- same weight layout for both.
    chain = (weight2 [Cout, Cout, 1, 1], scale2, shift2, relu2): 1x1 conv fused in the epilogue (Cout == 64 or 128: all
    channels in one workgroup); the hidden rows are stored in the chain order of conv_halo.hip."""
    lib = _lib.load()
    w = _home(weight)
    cout, cin, k, _ = w.shape
    if C0 is None:
        C0 = cin
    tile = lib.v2x_conv_stream_tile_rows(cout, V2X_EPI_BF16)
    if k != 3 or C0 + C1 != cin or C0 % 32 or C1 % 32 or tile == 0:
        raise ValueError("%s: not a shape the streamed kernel covers" % name)
    wk = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
    if chain is not None:
        if cout not in (64, 128) or chain[0].shape[0] != cout:
            raise ValueError("%s: the streamed kernel chains only 64 -> 64 -> 64 and 128 -> 128 -> 128" % name)
        wk = wk[_chain_row_order(cout)]
    pc = PackedConv(name=name, weight=_stream_layout(wk, tile, cin).to(torch.bfloat16).to(device).contiguous(),
                    scale=scale.detach().float().to(device).contiguous(),
                    shift=shift.detach().float().to(device).contiguous(), C0=C0, C1=C1, Cout=cout, ksize=3,
                    stride=stride, pad=1, up0=up0, epilogue=V2X_EPI_BF16, relu=relu, w_rows=cout, w_kpad=9 * cin,
                    w_layout=2, Cout2=0)
    if stride == 2 and (C1 or up0 or chain is not None):
        raise ValueError("%s: the stride-2 streamed kernel takes one plain source" % name)
    if chain is not None:
        w2, s2, t2, relu2 = chain
        pc.Cout2, pc.relu2 = cout, relu2
        pc.weight2 = w2.detach().float().cpu().reshape(cout, cout).to(torch.bfloat16).to(device).contiguous()
        pc.scale2, pc.shift2 = s2.detach().float().to(device).contiguous(), t2.detach().float().to(device).contiguous()
    return pc


def pack_gru_stream(name, weight_ih, bias_ih, bias_hh, *, C0, C1, device="cuda"):
    """ConvGRU (h0 = 0) for conv_stream.hip: (r,z,n) row triples as pack_gru, streamed-slice layout."""
    w = weight_ih.detach().float().cpu()
    three_h, cin, k, _ = w.shape
    hid = three_h // 3
    if hid % 32 != 0 or cin != C0 + C1 or C0 % 32 or C1 % 32 or k != 3:
        raise ValueError("%s: not a shape the streamed GRU kernel covers" % name)
    K = 9 * cin
    wk = w.permute(0, 2, 3, 1).reshape(three_h, K)
    groups = hid // 16
    wk = wk.view(3, groups, 16, K).permute(1, 0, 2, 3).reshape(groups * 48, K)
    bi, bh = bias_ih.detach().float().cpu().view(3, hid), bias_hh.detach().float().cpu().view(3, hid)
    bias4 = torch.stack([bi[0] + bh[0], bi[1] + bh[1], bi[2], bh[2]], dim=1).contiguous()
    return PackedConv(name=name, weight=_stream_layout(wk, 96, cin).to(torch.bfloat16).to(device).contiguous(),
                      scale=bias4.to(device), shift=None, C0=C0, C1=C1, Cout=hid, ksize=3, stride=1, pad=1, up0=0,
                      epilogue=V2X_EPI_GRU, relu=False, w_rows=three_h, w_kpad=K, w_layout=2, Cout2=0)
