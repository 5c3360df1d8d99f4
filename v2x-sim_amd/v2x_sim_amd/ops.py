"""Stage-level python wrappers over the C ABI (one per SURVEY.md section 8 row).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic
happens inside libv2x_amd.so.  Every wrapper refuses non-device tensors -- there
is deliberately no eager/CPU fallback.
"""
import ctypes as C

import torch

from . import _lib, tuning
from ._lib import ConvDesc, V2X_EPI_BF16, V2X_EPI_F32, V2X_EPI_GRU, V2X_FUSE_MAX, V2X_FUSE_MEAN, V2X_FUSE_WSUM  # noqa: F401


from . import _launch
from ._launch import _Prof, _dev, _dev_opt, _stream  # noqa: F401


# Latency dispatch.  Kernel forms whose choice depends on the NUMBER of maps in a launch (split-K of the streamed layers, the 1-tap stride-2
# kernel below four tiles per CU) change the fp32 summation order, so they are never taken silently: the caller declares a latency launch.
# tuning SMALL_BATCH: 0 never; 1 every launch of the process (the explicit pin, also for the sharded runners); 2 (default) only inside a
# `with ops.latency_dispatch():` block -- the plain single-GPU model classes enter one in forward_nhwc() (@ops.latency_entry: the common entry
# of forward(), forward_points() and the Seg / Det modules), the sharded runners (R-rank == 1-rank bitwise) never do.
# NOT only a summation order: the streamed parity-class kernel (conv5_1, conv6_1: w_layout 4) has no split-K form, so a declared latency
# launch that splits takes `Layer.latency` -- the layer's 9-TAP bf16 weights -- while a throughput launch of the same model multiplies the PRE-SUMMED
# parity-class weights (fp32 sums of 1, 2 or 4 taps rounded to bf16 once).  The two forms agree with the fp32 oracle layer to the same bound
# (tests/test_gpu_parity_class.py) but differ from each other by that weight rounding, not just by summation order; one-frame inference mixes them
# (conv7_1 / conv8_1 parity-class, conv5_1 / conv6_1 9-tap).  Bounded end to end: tests/test_gpu_bits_input.py (latency vs throughput logits of one frame
# within 2e-2 of max|ref|) and tests/test_gpu_models.py (either dispatch within TOL of the oracle).
import threading as _threading

_latency = _threading.local()   # per THREAD: a latency forward in one thread must not switch a sharded runner or a training step in another
                                # thread to the split-K / 1-tap forms (that would break their R-rank == 1-rank bitwise guarantee)


class latency_dispatch:
    """Context manager: launches of THIS thread inside the block are declared latency launches (see above)."""

    def __init__(self, target=0):
        self.target = target      # > 0: the number of workgroups small_batch_splitk aims at inside this block (default 320; the training graph's sweep)

    def __enter__(self):
        _latency.depth = getattr(_latency, "depth", 0) + 1
        self.prev, _latency.target = getattr(_latency, "target", 0), self.target
        return self

    def __exit__(self, *exc):
        _latency.depth -= 1
        _latency.target = self.prev
        return False


def latency_entry(fn):
    """Decorator of the plain single-GPU model entries (forward_nhwc of every model class): launches inside are declared latency launches."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with latency_dispatch():
            return fn(*a, **k)
    return wrapped


def latency_launches():
    sb = tuning.get("SMALL_BATCH")
    return sb == 1 or (sb == 2 and getattr(_latency, "depth", 0) > 0)


_CONV_TILES = {32: (32, 256, 1, 4), 48: (48, 256, 1, 4), 64: (64, 128, 2, 2), 128: (128, 128, 2, 2), 96: (96, 128, 2, 2)}


def conv_kernel_name(pc, H=0, W=0, bits=False, N=0):
    """Name of the template instantiation v2x_conv2d dispatches to (as rocprofv3 prints it)."""
    if pc.w_layout == 2 and pc.stride == 2:
        if pc.Cout == 64 and pc.C0 == 32:
            return "conv3x3_s2_resident_kernel<64>"
        rows = 128 if pc.Cout % 128 == 0 else 64
        s2g = tuning.get("S2_G")
        if rows == 128 and pc.C0 >= 64 and s2g != 0:       # conv_stream_s2.hip: v2x_conv_stream_s2_dispatch
            Ho, Wo = H // 2, W // 2
            g32 = Ho % 8 == 0 and Wo % 32 == 0
            g16 = not g32 and Ho % 16 == 0 and Wo % 16 == 0
            tiles = N * (Ho * Wo // 256) * (pc.Cout // 128)
            if (g32 or g16) and not (latency_launches() and tiles < 4 * torch.cuda.get_device_properties(0).multi_processor_count):
                return "conv3x3_s2g_kernel<%s>" % ("8, 32" if g32 else "16, 16")
        if not (H % 8 == 0 and W % 64 == 0):
            return "conv3x3_s2_stream_kernel<%d, 8, 16>" % rows    # 16 x 16 outputs (conv4_1)
        return "conv3x3_s2_stream_kernel<%d, 4, 32>" % rows
    if pc.w_layout == 2:
        rows = _lib.load().v2x_conv_stream_tile_rows(pc.Cout, pc.epilogue)
        th, tw = (8, 32) if W % 32 == 0 else (16, 16)
        epi = 2 if pc.epilogue == V2X_EPI_GRU else (1 if pc.Cout2 else 0)
        if W % 32 == 0 and H % 16 == 0 and rows == 64 and pc.epilogue == V2X_EPI_BF16 and (pc.C0 + pc.C1) >= 64 \
                and tuning.get("STREAM_WIDE") != 0:
            if epi == 0 and (pc.C0 + pc.C1) >= 96 and tuning.get("WIDE3") != 0:
                return "conv3x3_wide3_kernel<64>"          # 128 pixels per wave, three taps per synchronisation
            return "conv3x3_wide_kernel<64, %d>" % epi   # 128 pixels per wave (conv_stream.hip)
        if W % 32 == 0 and H % 16 == 0 and rows in (96, 128) \
                and tuning.get("STREAM_WAVES") != 4:
            if epi != 1 and tuning.get("STREAM_G") != 0:
                wt = tuning.get("STREAM_WT")   # wave tiling: half the channels x 128 pixels per wave
                tiled = (wt >= 1 and epi == 0) or (wt >= 2 and epi == 2)
                return "conv3x3_stream8g_kernel<%d, %d, %s>" % (rows, epi, "true" if tiled else "false")  # 8 waves, three taps per synchronisation
            return "conv3x3_stream8_kernel<%d, %d>" % (rows, epi)  # 8-wave ping-pong form (conv_stream.hip)
        return "conv3x3_stream_kernel<%d, %d, %d, %d, false>" % (rows, th, tw, epi)
    if pc.w_layout == 4:
        return "conv3x3_stream8q_kernel" if pc.Cout == 64 else "conv3x3_stream8p_kernel"   # streamed parity-class forms (conv7_1 | conv5_1, conv6_1)
    if pc.w_layout == 3:
        return "conv3x3_halo_ppc_kernel<%d, %d, %d>" % (pc.C0, pc.C1, pc.Cout)   # parity-class form (pre-summed 2x2-tap weights for the upsampled source)
    if pc.w_layout == 1:
        c0, c1 = (pc.C0, pc.C1) if pc.C1 else (0, pc.C0)
        co2 = (pc.Cout2 + 15) // 16 * 16 if pc.Cout2 else 0
        e2 = 0 if not pc.Cout2 else (2 if pc.epilogue == V2X_EPI_F32 else 1)
        if (c0, c1, pc.Cout, co2) == (0, 32, 32, 0):  # HBM-bound layers: single-buffer form (+ bit-grid input)
            return "conv3x3_halo_sb_kernel<0, 32, 32, 0, 0, %s>" % ("true" if bits else "false")
        if (c0, c1, pc.Cout, co2, e2) == (0, 64, 64, 64, 1):
            return "conv3x3_halo_pp_kernel<0, 64, 64, 64>"   # conv1_2 -> conv3d_1 chained: ping-pong form (odd tile counts: the 4-wave form)
        if (c0, c1, pc.Cout, co2) in ((64, 32, 32, 0), (0, 64, 64, 0)) and tuning.get("HALO_PP") != 0:
            return "conv3x3_halo_pp_kernel<%d, %d, %d, 0>" % (c0, c1, pc.Cout)   # conv8_1 / conv7_2: 8-wave ping-pong form
        return "conv3x3_halo_kernel<%d, %d, %d, %d, %d>" % (c0, c1, pc.Cout, co2, e2)
    if (pc.ksize == 1 and pc.stride == 1 and not pc.C1 and not pc.Cout2 and pc.epilogue in (V2X_EPI_BF16, V2X_EPI_F32) and pc.C0 % 32 == 0 and pc.C0 <= 128
            and pc.C0 // 32 != 3 and 4 <= pc.Cout <= 128 and pc.Cout % 4 == 0 and (pc.Cout + 15) // 16 in (1, 2, 3, 4, 6, 8) and tuning.get("CONV1X1") != 0):
        ks, ct = pc.C0 // 32, (pc.Cout + 15) // 16      # conv1x1.hip: the streaming 1x1 kernel (round 6; same bits as the gather kernel)
        return "conv1x1_stream_kernel<%d, %d, %s, %d>" % (ks, ct, "true" if pc.epilogue == V2X_EPI_F32 else "false", 2 if ks * ct >= 16 else 4)
    rows = _lib.load().v2x_conv_tile_rows(pc.Cout, pc.epilogue)
    return "conv_igemm_kernel<%d, %d, %d, %d, %d>" % (_CONV_TILES[rows] + (pc.epilogue,))


# ------------------------------------------------------------------ a1
class VoxelGrid:
    """Geometry of the BEV grid (upstream Config.voxel_size / area_extents)."""

    def __init__(self, voxel_size=(0.25, 0.25, 0.4), area_extents=((-32.0, 32.0), (-32.0, 32.0), (-3.0, 2.0))):
        import math
        self.voxel = tuple(float(v) for v in voxel_size)
        self.extents = tuple((float(lo), float(hi)) for lo, hi in area_extents)
        self.dims = tuple(int(math.ceil(hi / v) - 1 - math.floor(lo / v) + 1)
                          for (lo, hi), v in zip(self.extents, self.voxel))
        self._ext = (C.c_double * 6)(*[x for lohi in self.extents for x in lohi])
        self._vox = (C.c_double * 3)(*self.voxel)
        self._dims = (C.c_int32 * 3)(*self.dims)


def voxelize_bits(points, n_pts, grid, out=None):
    """points (n_clouds, max_pts, stride>=3) fp32, n_pts (n_clouds,) int32 -> bits (n_clouds, X, Y) int32."""
    lib = _lib.load()
    if points.dim() != 3:
        raise ValueError("points must be (n_clouds, max_pts, stride)")
    n, mp, st = points.shape
    if tuple(n_pts.shape) != (n,):
        raise ValueError("n_pts must hold one count per cloud: expected shape (%d,), got %s" % (n, tuple(n_pts.shape)))
    X, Y, Z = grid.dims
    if out is None:
        out = torch.empty((n, X, Y), dtype=torch.int32, device=points.device)
    vl = tuning.get("VOXELIZE_LDS")
    lds = Z <= 16 and X * Y * 2 <= 128 * 1024 and (X * Y) % 8 == 0 and mp > 0 and vl != 0 and (vl == 2 or n > 48)
    prof = (_Prof("voxelize_lds_kernel", 0, points.numel() * 4 + out.numel() * 4) if lds else
            _Prof("voxelize_scatter_kernel", 0, points.numel() * 4 + 2 * out.numel() * 4))
    rc = lib.v2x_voxelize_bits(_dev(points, torch.float32, "points"), _dev(n_pts, torch.int32, "n_pts"), n, mp, st,
                               grid._ext, grid._vox, grid._dims, _dev(out, torch.int32, "bits"), _stream())
    prof.done()
    _lib.check(rc, "v2x_voxelize_bits")
    return out


def voxelize_fused_bits(points, n_pts, xform, src_cloud, dst_grid, n_grids, grid):
    """Early fusion: job j scatters cloud src_cloud[j], moved by xform[j] (n_jobs, 3, 4) fp32, into grid dst_grid[j].
    -> bits (n_grids, X, Y) int32."""
    lib = _lib.load()
    n, mp, st = points.shape
    n_jobs = xform.shape[0]
    if tuple(n_pts.shape) != (n,):
        raise ValueError("n_pts must hold one count per cloud: expected shape (%d,), got %s" % (n, tuple(n_pts.shape)))
    if tuple(xform.shape) != (n_jobs, 3, 4) or src_cloud.shape[0] != n_jobs or dst_grid.shape[0] != n_jobs:
        raise ValueError("xform (n_jobs, 3, 4), src_cloud (n_jobs,), dst_grid (n_jobs,) expected")
    X, Y, Z = grid.dims
    out = torch.empty((n_grids, X, Y), dtype=torch.int32, device=points.device)
    rc = lib.v2x_voxelize_fused_bits(_dev(points, torch.float32, "points"), _dev(n_pts, torch.int32, "n_pts"), n, mp, st,
                                     _dev(xform, torch.float32, "xform"), _dev(src_cloud, torch.int32, "src_cloud"),
                                     _dev(dst_grid, torch.int32, "dst_grid"), n_jobs, n_grids, grid._ext, grid._vox,
                                     grid._dims, _dev(out, torch.int32, "bits"), _stream())
    _lib.check(rc, "v2x_voxelize_fused_bits")
    return out


def bits_to_dense(bits, Z):
    lib = _lib.load()
    n, X, Y = bits.shape
    out = torch.empty((n, X, Y, Z), dtype=torch.float32, device=bits.device)
    _lib.check(lib.v2x_bits_to_dense_f32(_dev(bits, torch.int32, "bits"), n, X, Y, Z, _dev(out, torch.float32, "out"),
                                         _stream()), "v2x_bits_to_dense_f32")
    return out


def bits_to_nhwc(bits, Z, c_pad=16, out=None):
    lib = _lib.load()
    n, X, Y = bits.shape
    if out is None:
        out = torch.empty((n, X, Y, c_pad), dtype=torch.bfloat16, device=bits.device)
    prof = _Prof("bits_to_nhwc_bf16_kernel", 0, bits.numel() * 4 + out.numel() * 2)
    rc = lib.v2x_bits_to_nhwc_bf16(_dev(bits, torch.int32, "bits"), n, X, Y, Z, c_pad,
                                   _dev(out, torch.bfloat16, "out"), _stream())
    prof.done()
    _lib.check(rc, "v2x_bits_to_nhwc_bf16")
    return out


def dense_to_nhwc(bev, c_pad=16, out=None):
    """bev (n, X, Y, Z) fp32 -> (n, X, Y, c_pad) bf16."""
    lib = _lib.load()
    n, X, Y, Z = bev.shape
    if out is None:
        out = torch.empty((n, X, Y, c_pad), dtype=torch.bfloat16, device=bev.device)
    _lib.check(lib.v2x_dense_f32_to_nhwc_bf16(_dev(bev, torch.float32, "bev"), n, X, Y, Z, c_pad,
                                              _dev(out, torch.bfloat16, "out"), _stream()),
               "v2x_dense_f32_to_nhwc_bf16")
    return out


def bits_to_indices(bits, Z, cap):
    """-> (idx (n, cap, 3) int32 sorted lexicographically, counts (n,) int32)."""
    lib = _lib.load()
    n, X, Y = bits.shape
    idx = torch.zeros((n, cap, 3), dtype=torch.int32, device=bits.device)
    counts = torch.zeros((n,), dtype=torch.int32, device=bits.device)
    scratch = torch.empty((n, X), dtype=torch.int32, device=bits.device)
    _lib.check(lib.v2x_bits_to_indices(_dev(bits, torch.int32, "bits"), n, X, Y, Z, _dev(idx, torch.int32, "idx"), cap,
                                       _dev(counts, torch.int32, "counts"), _dev(scratch, torch.int32, "scratch"),
                                       _stream()), "v2x_bits_to_indices")
    return idx, counts


def indices_to_bits(idx, counts, grid):
    """idx (n, cap, 3) int32 sparse voxel indices, counts (n,) int32 -> bits (n, X, Y) int32 (the densify
    scatter of the parsed-dataset reader, on the GPU)."""
    lib = _lib.load()
    n, cap, three = idx.shape
    if three != 3:
        raise ValueError("idx must be (n, cap, 3)")
    X, Y, Z = grid.dims
    out = torch.empty((n, X, Y), dtype=torch.int32, device=idx.device)
    _lib.check(lib.v2x_indices_to_bits(_dev(idx, torch.int32, "idx"), _dev(counts, torch.int32, "counts"), n, cap,
                                       X, Y, Z, _dev(out, torch.int32, "bits"), _stream()), "v2x_indices_to_bits")
    return out


# ------------------------------------------------------------------ a2/a4/a6/a7/a8
class PackedConv:
    """Device-resident packed parameters of one conv layer (see packing.py)."""

    __slots__ = ("weight", "scale", "shift", "C0", "C1", "Cout", "ksize", "stride", "pad", "up0", "epilogue", "relu",
                 "w_rows", "w_kpad", "name", "w_layout", "Cout2", "weight2", "scale2", "shift2", "relu2",
                 "repack")   # packing.pack_conv_device: what packing.RepackPlan needs to rebuild `weight` in place (None otherwise)

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


def conv_out_hw(pc, H, W):
    Ho = (H + 2 * pc.pad - pc.ksize) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.ksize) // pc.stride + 1
    return Ho, Wo


def conv2d(pc, in0, in1=None, out=None, out_coff=0, split=0, zbits=0, splitk=0):
    """in0 (N, H>>up0, W>>up0, C0) bf16 NHWC [, in1 (N, H, W, C1)] -> out (N, Ho, Wo, Cout) NHWC.

    split > 0: returns (out[..., :split], out2[..., Cout-split]) as two contiguous tensors.
    splitk > 1 (streamed stride-1 layers without a chained 1x1): the small-batch form, see small_batch_splitk()."""
    lib = _lib.load()
    # the gather kernel does its row arithmetic in 24 bits (N*H*W < 2^24): larger batches go through in slices of whole
    # maps (contiguous NHWC views, same stream) -- the only kernel with that limit, and only 256x256 inputs reach it
    if not pc.w_layout and in0.dtype != torch.int32:
        Hx, Wx = (in1.shape[1], in1.shape[2]) if in1 is not None else (in0.shape[1] << pc.up0, in0.shape[2] << pc.up0)
        N0 = in0.shape[0]
        if N0 * Hx * Wx >= (1 << 24) and N0 > 1:
            Ho, Wo = conv_out_hw(pc, Hx, Wx)
            odt = torch.float32 if pc.epilogue == V2X_EPI_F32 else torch.bfloat16
            cfin = pc.Cout2 if pc.Cout2 else pc.Cout
            o1 = o2 = None
            if split:
                o1 = torch.empty((N0, Ho, Wo, split), dtype=odt, device=in0.device)
                o2 = torch.empty((N0, Ho, Wo, cfin - split), dtype=odt, device=in0.device)
            elif out is None:
                out = torch.empty((N0, Ho, Wo, cfin), dtype=odt, device=in0.device)
            step = max(1, ((1 << 24) - 1) // (Hx * Wx))
            for a0 in range(0, N0, step):
                sl = slice(a0, min(a0 + step, N0))
                r = conv2d(pc, in0[sl], None if in1 is None else in1[sl], None if split else out[sl], out_coff, split, zbits)
                if split:
                    o1[sl].copy_(r[0])
                    o2[sl].copy_(r[1])
            return (o1, o2) if split else out
    d = ConvDesc()
    from_bits = in0.dtype == torch.int32  # the voxelizer's bit grid (N, H, W): first layer, halo kernel only
    if from_bits:
        if pc.w_layout != 1 or in1 is not None or in0.dim() != 3 or not (1 <= zbits <= 32):
            raise ValueError("bit-grid input needs a halo-packed single-source layer and 1 <= zbits <= 32")
        d.in0 = _dev(in0, torch.int32, "in0").value
        d.in_format, d.in_zbits = 1, zbits
        in0_shape = (in0.shape[0], in0.shape[1], in0.shape[2], pc.C0)
    else:
        d.in0 = _dev(in0, torch.bfloat16, "in0").value
        in0_shape = tuple(in0.shape)
    N = in0.shape[0]
    if in1 is not None:
        d.in1 = _dev(in1, torch.bfloat16, "in1").value
        H, W = in1.shape[1], in1.shape[2]
        if in1.shape[3] != pc.C1 or in1.shape[0] != N:
            raise ValueError("in1 shape %s does not match C1=%d" % (tuple(in1.shape), pc.C1))
        if (in0.shape[1] << pc.up0, in0.shape[2] << pc.up0) != (H, W):
            raise ValueError("in0 %s (up0=%d) does not match in1 %s" % (tuple(in0.shape), pc.up0, tuple(in1.shape)))
    else:
        if pc.C1:
            raise ValueError("layer %s expects a second source" % pc.name)
        d.in1 = None
        H, W = in0.shape[1] << pc.up0, in0.shape[2] << pc.up0
    if in0_shape[3] != pc.C0:
        raise ValueError("in0 has %d channels, layer %s expects %d" % (in0_shape[3], pc.name, pc.C0))
    Ho, Wo = conv_out_hw(pc, H, W)
    odt = torch.float32 if pc.epilogue == V2X_EPI_F32 else torch.bfloat16
    cfin = pc.Cout2 if pc.Cout2 else pc.Cout  # channels of the tensor that is actually written
    out2 = None
    if split:
        out = torch.empty((N, Ho, Wo, split), dtype=odt, device=in0.device)
        out2 = torch.empty((N, Ho, Wo, cfin - split), dtype=odt, device=in0.device)
    elif out is None:
        out = torch.empty((N, Ho, Wo, cfin), dtype=odt, device=in0.device)
    elif tuple(out.shape[:3]) != (N, Ho, Wo):
        raise ValueError("out shape %s != %s" % (tuple(out.shape), (N, Ho, Wo, "*")))
    d.C0, d.C1, d.up0 = pc.C0, pc.C1, pc.up0
    d.N, d.H, d.W = N, H, W
    d.ksize, d.stride, d.pad = pc.ksize, pc.stride, pc.pad
    d.Cout, d.w_rows, d.w_kpad = pc.Cout, pc.w_rows, pc.w_kpad
    d.weight = pc.weight.data_ptr()
    d.scale = pc.scale.data_ptr()
    d.shift = pc.shift.data_ptr() if pc.shift is not None else None
    d.epilogue, d.relu = pc.epilogue, int(bool(pc.relu))
    d.out = _dev(out, odt, "out").value
    d.out_cstride, d.out_coff = out.shape[3], out_coff
    if split:
        d.out2, d.split, d.out2_cstride = out2.data_ptr(), split, out2.shape[3]
    d.w_layout = pc.w_layout or 0
    if pc.Cout2:
        d.Cout2, d.relu2 = pc.Cout2, int(bool(pc.relu2))
        d.weight2, d.scale2, d.shift2 = pc.weight2.data_ptr(), pc.scale2.data_ptr(), pc.shift2.data_ptr()
    ws = None
    if splitk > 1:
        ws = torch.empty((splitk, N * Ho * Wo, pc.w_rows), dtype=torch.float32, device=in0.device)
        d.splitk, d.splitk_ws = splitk, ws.data_ptr()
    d.small_batch = 1 if latency_launches() else 0
    prof = None
    if _launch.PROFILE is not None:
        rows_logical = 3 * pc.Cout if pc.epilogue == V2X_EPI_GRU else pc.Cout
        k_logical = pc.ksize * pc.ksize * (zbits if from_bits else pc.C0 + pc.C1)   # bit-grid input: zbits real channels, the rest padding
        M = N * Ho * Wo
        nbytes = in0.numel() * (4 if from_bits else 2) + (in1.numel() * 2 if in1 is not None else 0) + pc.weight.numel() * 2 \
            + M * cfin * (4 if pc.epilogue == V2X_EPI_F32 else 2)
        flops = 2.0 * M * (rows_logical * k_logical + (pc.Cout2 or 0) * pc.Cout)
        if pc.w_layout in (3, 4):   # parity-class forms: the FLOPs the kernel EXECUTES (4 taps on the upsampled source); the reference's 9-tap count is
            flops = 2.0 * M * pc.Cout * (4 * pc.C0 + 9 * pc.C1)   # reported separately (bench.py: reference_flops)
        prof = _Prof(conv_kernel_name(pc, H, W, from_bits, N) if splitk <= 1 else
                     "conv3x3_%sstream_kernel<%d, split-K %d> + splitk_reduce" % ("s2_" if pc.stride == 2 else "", lib.v2x_conv_stream_tile_rows(pc.Cout, pc.epilogue), splitk),
                     flops, nbytes, pc.name)
    rc = lib.v2x_conv2d(C.byref(d), _stream())
    if prof is not None:
        prof.done()
    _lib.check(rc, "v2x_conv2d(%s)" % pc.name)
    return (out, out2) if split else out


def _pair_desc(pc, N, H, W):
    d = ConvDesc()
    d.C0, d.C1, d.up0 = pc.C0, pc.C1, pc.up0
    d.N, d.H, d.W = N, H, W
    d.ksize, d.stride, d.pad = pc.ksize, pc.stride, pc.pad
    d.Cout, d.w_rows, d.w_kpad = pc.Cout, pc.w_rows, pc.w_kpad
    d.weight, d.scale, d.shift = pc.weight.data_ptr(), pc.scale.data_ptr(), pc.shift.data_ptr()
    d.epilogue, d.relu = pc.epilogue, int(bool(pc.relu))
    d.w_layout = pc.w_layout or 0
    return d


def pair_eligible(pa, pb, in0, zbits):
    """conv_halo_pair.hip covers: bit-grid input, two halo-packed 3x3 s1 32 -> 32 bf16 layers, H % 8 == 0, W % 32 == 0."""
    ok = lambda pc: (pc is not None and pc.w_layout == 1 and pc.ksize == 3 and pc.stride == 1 and pc.C0 == 32 and not pc.C1  # noqa: E731
                     and pc.Cout == 32 and not pc.Cout2 and pc.epilogue == V2X_EPI_BF16)
    return (ok(pa) and ok(pb) and in0.dtype == torch.int32 and in0.dim() == 3 and 1 <= zbits <= 16
            and in0.shape[1] % 8 == 0 and in0.shape[2] % 32 == 0 and tuning.get("CONV_PAIR") != 0)


def conv2d_pair(pa, pb, bits, zbits, out=None):
    """pb(pa(bits)) in one launch, the intermediate map never stored (conv_pre_1 -> conv_pre_2).  bits (N, H, W) int32."""
    lib = _lib.load()
    N, H, W = bits.shape
    if out is None:
        out = torch.empty((N, H, W, pb.Cout), dtype=torch.bfloat16, device=bits.device)
    da, db = _pair_desc(pa, N, H, W), _pair_desc(pb, N, H, W)
    da.in0 = _dev(bits, torch.int32, "bits").value
    da.in_format, da.in_zbits = 1, zbits
    db.out = _dev(out, torch.bfloat16, "out").value
    db.out_cstride, db.out_coff = out.shape[3], 0
    prof = None
    if _launch.PROFILE is not None:
        M = N * H * W
        # algorithmic FLOPs: the first layer has `zbits` (13) real input channels -- the 32 of the packed layout are zero padding
        prof = _Prof("conv3x3_pair_bits_kernel<%s>" % ("true" if tuning.get("STORE_X4") != 0 else "false"), 2.0 * M * 9 * (pa.Cout * zbits + pb.Cout * pb.C0),
                     bits.numel() * 4 + out.numel() * 2 + (pa.weight.numel() + pb.weight.numel()) * 2, pa.name + "+" + pb.name)
    rc = lib.v2x_conv2d_pair(C.byref(da), C.byref(db), _stream())
    if prof is not None:
        prof.done()
    _lib.check(rc, "v2x_conv2d_pair(%s, %s)" % (pa.name, pb.name))
    return out


def tail_eligible(pa, pb, x, split=12):
    """conv_tail.hip covers: conv8_2 as a halo-packed 3x3 s1 32 -> 32 bf16 layer, the fused heads (halo-packed 3x3 32 -> 64 chained with the
    1x1 -> split + (48 - split) without ReLU, fp32, split in {4, 8, 12}), a bf16 NHWC input with H % 8 == 0, W % 32 == 0 and N H W < 2^27."""
    return (pa is not None and pb is not None and split in (4, 8, 12) and pa.w_layout == 1 and pa.ksize == 3 and pa.stride == 1 and pa.C0 == 32 and not pa.C1 and pa.Cout == 32
            and not pa.Cout2 and pa.epilogue == V2X_EPI_BF16 and pb.w_layout == 1 and pb.ksize == 3 and pb.stride == 1 and pb.C0 == 32 and not pb.C1
            and pb.Cout == 64 and pb.Cout2 == 48 and not pb.relu2 and pb.epilogue == V2X_EPI_F32 and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == 32
            and x.shape[1] % 8 == 0 and x.shape[2] % 32 == 0 and x.shape[0] * x.shape[1] * x.shape[2] < (1 << 27) and tuning.get("TAIL_FUSE") != 0)


def conv2d_tail(pa, pb, x, split):
    """heads(conv8_2(x)) in one launch (conv_tail.hip through v2x_conv2d_pair's second form): x (N, H, W, 32) bf16 = conv8_1's output ->
    (cls (N, H, W, split), loc (N, H, W, 48 - split)) fp32, bit-identical to conv2d(pa) followed by conv2d(pb, split=split)."""
    lib = _lib.load()
    N, H, W, _ = x.shape
    out = torch.empty((N, H, W, split), dtype=torch.float32, device=x.device)
    out2 = torch.empty((N, H, W, pb.Cout2 - split), dtype=torch.float32, device=x.device)
    da, db = _pair_desc(pa, N, H, W), _pair_desc(pb, N, H, W)
    da.in0 = _dev(x, torch.bfloat16, "x").value
    db.Cout2, db.relu2 = pb.Cout2, int(bool(pb.relu2))
    db.weight2, db.scale2, db.shift2 = pb.weight2.data_ptr(), pb.scale2.data_ptr(), pb.shift2.data_ptr()
    db.out, db.out_cstride, db.out_coff = out.data_ptr(), split, 0
    db.out2, db.split, db.out2_cstride = out2.data_ptr(), split, out2.shape[3]
    prof = None
    if _launch.PROFILE is not None:
        M = N * H * W
        prof = _Prof("conv3x3_tail_kernel", 2.0 * M * (9 * 32 * 32 + 9 * 32 * 64 + 64 * 48), x.numel() * 2 + (out.numel() + out2.numel()) * 4 +
                     (pa.weight.numel() + pb.weight.numel() + pb.weight2.numel()) * 2, pa.name + "+" + pb.name)
    rc = lib.v2x_conv2d_pair(C.byref(da), C.byref(db), _stream())
    if prof is not None:
        prof.done()
    _lib.check(rc, "v2x_conv2d_pair(%s, %s)" % (pa.name, pb.name))
    return out, out2


def warp_fuse(feat, A, Bt, trans, items, coef, mode, out=None):
    """feat (A*Bt, H, W, C) bf16; trans (Bt, A, A, 4, 4) fp32; items (n_out, 2) int32; coef (n_out, A) fp32."""
    lib = _lib.load()
    n_items, H, W, Cc = feat.shape
    if n_items != A * Bt:
        raise ValueError("feat holds %d maps, expected A*Bt=%d" % (n_items, A * Bt))
    if tuple(trans.shape) != (Bt, A, A, 4, 4):
        raise ValueError("trans must be (Bt, A, A, 4, 4), got %s" % (tuple(trans.shape),))
    n_out = items.shape[0]
    if tuple(coef.shape) != (n_out, A):
        raise ValueError("coef must be (n_out, A)")
    if out is None:
        out = torch.empty((n_out, H, W, Cc), dtype=torch.bfloat16, device=feat.device)
    lds = H % 8 == 0 and W % 8 == 0 and Cc % 128 == 0 and tuning.get("WARP_LDS") != 0
    prof = _Prof(("warp_fuse_lds2_kernel<%d>" % mode if tuning.get("WARP_LDS") >= 2 else "warp_fuse_lds_kernel") if lds else "warp_fuse_kernel", 0,
                 (n_out * (A - 1) + n_out) * H * W * Cc * 2)
    order = _warp_frame_order(items, A, Bt) if (lds and tuning.get("WARP_XCD") != 0) else None
    if order is not None:
        rc = lib.v2x_warp_fuse_ordered(_dev(feat, torch.bfloat16, "feat"), A, Bt, H, W, Cc, _dev(trans, torch.float32, "trans"),
                                       _dev(items, torch.int32, "items"), n_out, _dev(coef, torch.float32, "coef"), mode,
                                       _dev(out, torch.bfloat16, "out"), _dev(order, torch.int32, "order"), A, order.numel(), _stream())
    else:
        rc = lib.v2x_warp_fuse(_dev(feat, torch.bfloat16, "feat"), A, Bt, H, W, Cc, _dev(trans, torch.float32, "trans"),
                               _dev(items, torch.int32, "items"), n_out, _dev(coef, torch.float32, "coef"), mode,
                               _dev(out, torch.bfloat16, "out"), _stream())
    prof.done()
    _lib.check(rc, "v2x_warp_fuse")
    return out


def items_tensor(items, A, Bt, device):
    """(n_out, 2) int32 device tensor of the (ego, frame) pairs of a fusion plan, with the frame-ordered table v2x_warp_fuse_ordered walks
    attached (built here on the host, where the list is: no device round trip later)."""
    t = torch.tensor(items, dtype=torch.int32, device=device).view(-1, 2)
    order = [-1] * (A * Bt)
    ok = True
    for m, (a, f) in enumerate(items):
        if not (0 <= a < A and 0 <= f < Bt) or order[f * A + a] >= 0:
            ok = False                       # outside the table, or listed twice: the plain grid
            break
        order[f * A + a] = m
    t._v2x_frame_order = (t._version, torch.tensor(order, dtype=torch.int32, device=device).view(Bt, A) if ok and len(items) else False)
    return t


def _warp_frame_order(items, A, Bt):
    """(Bt, A) int32 table: the output map of (frame, ego), -1 where there is none -- what v2x_warp_fuse_ordered walks so that the output maps
    of one frame (which all read the same A source maps) are computed by one XCD and share its L2.  Built once per items tensor (the fusion plan
    keeps it); never inside a hipGraph capture (the table would live in the graph's private pool): the caller then takes the unordered launch."""
    tag = getattr(items, "_v2x_frame_order", None)
    if tag is None or tag[0] != items._version:            # (a table built for other contents of this tensor is not trusted)
        if torch.cuda.is_current_stream_capturing():
            return None
        order = torch.full((Bt, A), -1, dtype=torch.int32, device=items.device)
        order[items[:, 1].long(), items[:, 0].long()] = torch.arange(items.shape[0], dtype=torch.int32, device=items.device)
        if int((order >= 0).sum()) != items.shape[0]:
            order = False                        # an (ego, frame) pair listed twice: the table cannot hold both -> the plain grid, always
        tag = (items._version, order)
        items._v2x_frame_order = tag
    return None if tag[1] is False else tag[1]


# ------------------------------------------------------------------ a5
ATTN_MODES = {"softmax": 0, "activated": 1, "argmax_test": 2}


def attn_handshake(keys, querys, w_lin, b_lin, A, Bt, mode, thres=0.2):
    """keys (A*Bt, K) fp32, querys (A*Bt, Q) fp32 -> prob, coef (Bt, A_key, A_query) fp32."""
    lib = _lib.load()
    K, Q = keys.shape[1], querys.shape[1]
    prob = torch.empty((Bt, A, A), dtype=torch.float32, device=keys.device)
    coef = torch.empty_like(prob)
    _lib.check(lib.v2x_attn_handshake(_dev(keys, torch.float32, "keys"), _dev(querys, torch.float32, "querys"),
                                      _dev(w_lin, torch.float32, "w_lin"), _dev(b_lin, torch.float32, "b_lin"), A, Bt,
                                      K, Q, ATTN_MODES[mode], C.c_float(thres), _dev(prob, torch.float32, "prob"),
                                      _dev(coef, torch.float32, "coef"), _stream()), "v2x_attn_handshake")
    return prob, coef


# ------------------------------------------------------------------ f-4 (DiscoNet)
def pixel_weighted_fuse(scores, valid, maps):
    """scores (n, A, H, W, S) fp32 (channel 0 used), valid (n, A) fp32, maps (n, A, H, W, C) bf16 -> (n, H, W, C) bf16."""
    lib = _lib.load()
    n, A, H, W, Cc = maps.shape
    out = torch.empty((n, H, W, Cc), dtype=torch.bfloat16, device=maps.device)
    _lib.check(lib.v2x_pixel_weighted_fuse(_dev(scores, torch.float32, "scores"), scores.shape[-1],
                                           _dev(valid, torch.float32, "valid"), _dev(maps, torch.bfloat16, "maps"), n, A, H, W,
                                           Cc, _dev(out, torch.bfloat16, "out"), _stream()), "v2x_pixel_weighted_fuse")
    return out


# ------------------------------------------------------------------ a8
def seg_argmax_confusion(logits, label=None, want_pred=True):
    """logits (n, H, W, n_cls) fp32 NHWC; label (n, H, W) uint8 -> (pred uint8, conf int64 [n_cls, n_cls])."""
    lib = _lib.load()
    n, H, W, ncls = logits.shape
    pred = torch.empty((n, H, W), dtype=torch.uint8, device=logits.device) if want_pred else None
    conf = torch.zeros((ncls, ncls), dtype=torch.int64, device=logits.device) if label is not None else None
    _lib.check(lib.v2x_seg_argmax_confusion(
        _dev(logits, torch.float32, "logits"), _dev(label, torch.uint8, "label") if label is not None else None,
        n, H, W, ncls, _dev(pred, torch.uint8, "pred") if pred is not None else None,
        _dev(conf, torch.int64, "conf") if conf is not None else None, _stream()), "v2x_seg_argmax_confusion")
    return pred, conf


# ------------------------------------------------------------------ layer = halo kernel when eligible, else gather kernel(s)
class Layer:
    """One logical layer (possibly a fused pair).  `halo` is the k-slot-major packing for
    conv_halo.hip (3x3 stride 1, H % 8 == 0, W % 32 == 0); `fallback` is the list of gather-kernel
    convs computing the same thing for any other extent.  `split` > 0: two output tensors."""

    __slots__ = ("halo", "fallback", "split", "name", "det", "latency")

    def __init__(self, fallback, halo=None, split=0, name=None, latency=None):
        self.fallback, self.halo, self.split = list(fallback), halo, split
        self.latency = latency   # a 9-tap streamed packing for DECLARED latency launches when `halo` is a form without split-K (the streamed parity-class kernel)
        self.det = None      # det heads only: the packing with the score threshold in the epilogue (packing.pack_heads_det)
        self.name = name or self.fallback[0].name


def _splitk_gate():
    """Launches with at least this many tiles are never split: 200 (measured, one-frame inference), or 5/8 of a latency_dispatch(target=...) block's target."""
    return max(200, (getattr(_latency, "target", 0) or tuning.get("SPLITK_TARGET")) * 5 // 8)


def small_batch_splitk(pc, N, H, W):
    """Latency mode (a declared latency launch: latency_launches() above): how many chunk ranges a streamed stride-1 layer is split into so
    that a launch has about one workgroup per CU.  One collaborative frame is 5 maps: a 32x32 layer with 256 output channels is then 40
    tiles of 256 pixels x 128 channels on 256 CUs, each walking all its chunks (conv5_1: 216 steps) -- 117 us for 18 GFLOP.  With the
    chunks divided over `splitk` workgroups per tile (partial sums added by splitk_reduce_kernel in range order) the same layer runs on
    ~240.  The mode is DECLARED by the caller, never derived from the batch inside a sharded runner: the split changes the fp32 summation
    order (one bf16 rounding of the output), and kernel selection that followed the item count would break the R-rank == 1-rank bitwise
    equality."""
    if pc.w_layout != 2 or pc.stride not in (1, 2) or not latency_launches():
        return 0
    rows = _lib.load().v2x_conv_stream_tile_rows(pc.Cout, pc.epilogue)
    if rows not in (64, 96, 128):
        return 0
    if pc.stride == 2:
        # the 1-tap stride-2 kernel (128 output pixels per workgroup): conv4_1 at one frame is 20-40 workgroups walking 8 chunks each (34 us);
        # conv1_1 (one chunk, resident form) and conv2_1 (two chunks) are never split
        if pc.C1 or pc.up0 or pc.Cout2 or pc.epilogue != V2X_EPI_BF16 or pc.w_rows != pc.Cout:
            return 0
        if not ((H % 8 == 0 and W % 64 == 0) or (H % 16 == 0 and W % 32 == 0)):
            return 0
        tiles = N * ((H // 2) * (W // 2) // 128) * (pc.Cout // rows)
        chunks = pc.C0 // 32
        if tiles >= _splitk_gate() or chunks < 8:                 # (conv3_1, 4 chunks: two ranges + the reduce launch 21 us against 20 unsplit)
            return 0
        want = min(chunks // 2, -(-(getattr(_latency, "target", 0) or tuning.get("SPLITK_TARGET")) // tiles))
        while want > 1 and -(-chunks // want) * (want - 1) >= chunks:
            want -= 1
        return want if want > 1 else 0
    t16 = W % 32 != 0
    if (t16 and (W % 16 or H % 16)) or (not t16 and H % 8):
        return 0
    tiles = N * (H * W // 256) * (pc.w_rows // rows)       # workgroups of the 4-wave form (256-pixel tiles)
    chunks = (pc.C0 + pc.C1) // 32
    # Measured (one frame = 5 maps / eight = 40, tools/layer_profile.py): a split pays when the unsplit launch has fewer than ~200 tiles
    # AND every range keeps >= 2 chunks (conv6_2, 4 chunks: 4 ranges of one 35 us vs 32 unsplit); the 4-wave kernel WITHOUT a split is slower
    # than the 8-wave forms even at 160 workgroups (conv5_1 at 40 maps: 163 vs 137 us), so there is no "more, smaller tiles" mode.
    if pc.Cout2 or tiles >= _splitk_gate() or chunks < 4:
        return 0
    want = min(chunks // 2, -(-(getattr(_latency, "target", 0) or tuning.get("SPLITK_TARGET")) // tiles))
    while want > 1 and -(-chunks // want) * (want - 1) >= chunks:     # no empty range
        want -= 1
    return want if want > 1 else 0


def halo_eligible(H, W, w_layout=1, cmax=0, cout=0):
    """cmax: the widest source's channel count -- the halo kernel's packed DMA tables hold a lane's element offset inside the patch rows
    in 20 bits (conv_halo.hip: (10 W + 34) cmax < 2^20, i.e. W < 1 635 at 64 channels); wider maps take the layer's fallback."""
    if w_layout in (1, 3) and (10 * W + 34) * cmax >= (1 << 20):
        return False
    if w_layout == 4:
        return H % 16 == 0 and W % (64 if cout == 64 else 32) == 0      # the streamed parity-class kernels: 16 x 32 tiles (pairs of them at 64 rows)
    if H % 8 == 0 and W % 32 == 0:
        return True
    return w_layout == 2 and H % 16 == 0 and W % 16 == 0  # the streamed kernel also has 16x16 tiles


def run_layer(layer, in0, in1=None, zbits=0):
    """zbits > 0: in0 is the voxelizer's int32 bit grid (N, H, W) -- only the first layer's halo packing reads it."""
    if in0.dtype == torch.int32:
        h = layer.halo
        if h is None or not halo_eligible(in0.shape[1], in0.shape[2], 1) or h.w_layout != 1:
            in0 = bits_to_nhwc(in0, zbits, layer.fallback[0].C0)  # odd extent: expand, then the gather kernel
        else:
            return conv2d(h, in0, zbits=zbits)
    if in1 is not None:
        H, W = in1.shape[1], in1.shape[2]
    else:
        H, W = in0.shape[1], in0.shape[2]
    h = layer.halo
    if layer.latency is not None and latency_launches():
        sk = small_batch_splitk(layer.latency, in0.shape[0], H, W)
        if sk > 1:
            return conv2d(layer.latency, in0, in1, split=layer.split, splitk=sk)
    if h is not None and h.stride == 2:
        # stride-2 streamed kernel: 4x32 output tiles, or 8x16 ones for narrow maps (conv4_1: 16x16 outputs)
        if (H % 8 == 0 and W % 64 == 0) or (H % 16 == 0 and W % 32 == 0):
            return conv2d(h, in0, in1, split=layer.split, splitk=small_batch_splitk(h, in0.shape[0], H, W))
    elif h is not None and halo_eligible(H, W, h.w_layout, max(h.C0, h.C1 or 0), h.Cout):
        use = True
        if h.w_layout in (2, 4):
            # streamed kernel = one 256-pixel x <=128-channel tile per workgroup.  The choice looks at the map extent
            # only, never at the batch: kernel selection must not change with the number of items a rank owns, or
            # R-rank results would stop being bitwise equal to 1-rank results.  16x16 maps give Cout/128 workgroups
            # per map, enough to fill 256 CUs from ~13 frames x 5 agents on; below 16x16 the gather kernel is used.
            use = H * W >= 256
        if use:
            return conv2d(h, in0, in1, split=layer.split, splitk=small_batch_splitk(h, in0.shape[0], H, W))
    y = conv2d(layer.fallback[0], in0, in1, split=layer.split if len(layer.fallback) == 1 else 0)
    for i, pc in enumerate(layer.fallback[1:], 1):
        y = conv2d(pc, y, split=layer.split if i == len(layer.fallback) - 1 else 0)
    return y


# ------------------------------------------------------------------ rows f-1 and f-3 live in their own modules; `ops.<name>` keeps working
from .ops_post import conv2d_det, det_nms_candidates, det_postprocess, match_detections, rotated_iou  # noqa: E402,F401
from .ops_train import (bn_train_backward, bn_train_forward, cast_pad_chsum, channel_sum, conv3x3_wgrad, det_loss_backward, det_loss_forward,  # noqa: E402,F401
                        gru_gates, gru_gates_backward, gru_gates_nhwc, gru_gates_nhwc_backward, gru_gates_nhwc_ok, upcat, upcat_backward, v2v_message,
                        v2v_message_backward, warp_affine, zero_insert)


# `ops.PROFILE = []` / `ops.PROFILE` (bench.py, tools/): one list for every wrapper module, kept in _launch
import sys as _sys  # noqa: E402
import types as _types  # noqa: E402


class _OpsModule(_types.ModuleType):
    @property
    def PROFILE(self):
        return _launch.PROFILE

    @PROFILE.setter
    def PROFILE(self, value):
        _launch.PROFILE = value


_sys.modules[__name__].__class__ = _OpsModule
