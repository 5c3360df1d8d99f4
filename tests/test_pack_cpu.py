"""The C-ABI weight packers (v2x_pack_conv & co., include/v2x_amd.h "weight layouts") against the torch packers of
v2x_sim_amd/packing.py: bit-for-bit the same buffers for every layout / shape the models use.  A non-Python host builds
its w_layout 0 / 1 / 2 buffers with the C entry points; these run on the host, no GPU needed."""
import ctypes as C

import numpy as np
import pytest
import torch

from v2x_sim_amd import _lib, packing
from v2x_sim_amd._lib import PackSpec, V2X_EPI_BF16, V2X_EPI_F32, V2X_EPI_GRU


def c_pack(w, *, layout, epilogue=V2X_EPI_BF16, cin_pad=0, chain=0, gru=False, c_up=0):
    lib = _lib.load()
    w = np.ascontiguousarray(w.detach().float().numpy())
    rows, cin, k, _ = w.shape
    spec = PackSpec(Cout=rows // 3 if gru else rows, Cin=cin, ksize=k, cin_pad=cin_pad, w_layout=layout,
                    epilogue=V2X_EPI_GRU if gru else epilogue, chain=chain, c_up=c_up)
    w_rows, w_kpad = C.c_int32(), C.c_int32()
    nbytes = lib.v2x_pack_conv_size(C.byref(spec), C.byref(w_rows), C.byref(w_kpad))
    assert nbytes > 0, lib.v2x_last_error()
    dst = np.full(nbytes // 2, 0x7fff, np.uint16)          # poisoned: the packer must write every element
    rc = lib.v2x_pack_conv(C.byref(spec), w.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.v2x_last_error()
    return dst, w_rows.value, w_kpad.value


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1)


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(*shape, generator=g) * 0.2
    w.view(-1)[::7] *= 1e-3                                 # small magnitudes: bf16 rounding of subnormal-ish tails too
    return w


@pytest.mark.parametrize("cout,cin,k,cin_pad", [(32, 13, 3, 32), (64, 32, 3, 0), (512, 256, 3, 0), (12, 32, 1, 0),
                                                (36, 32, 1, 0), (256, 4096, 1, 0), (48, 64, 1, 0), (8, 32, 1, 0)])
def test_layout0_gather(cout, cin, k, cin_pad):
    w = rnd(cout, cin, k, k, seed=cout + cin)
    pc = packing.pack_conv("t", w, torch.ones(cout), torch.zeros(cout), cin_pad=cin_pad or None, device="cpu",
                           epilogue=V2X_EPI_F32 if cout in (12, 36) else V2X_EPI_BF16)
    got, rows, kpad = c_pack(w, layout=0, cin_pad=cin_pad, epilogue=pc.epilogue)
    assert (rows, kpad) == (pc.w_rows, pc.w_kpad)
    assert np.array_equal(got, bits(pc.weight))


@pytest.mark.parametrize("cout,cin,cin_pad,chain", [(32, 13, 32, 0), (32, 32, 0, 0), (32, 96, 0, 0), (64, 32, 0, 1), (64, 64, 0, 0)])
def test_layout1_halo(cout, cin, cin_pad, chain):
    w = rnd(cout, cin, 3, 3, seed=3 * cout + cin)
    ch = (rnd(48, cout, 1, 1, seed=5), torch.ones(48), torch.zeros(48), False) if chain else None
    pc = packing.pack_conv_halo("t", w, torch.ones(cout), torch.zeros(cout), cin_pad=cin_pad or None, chain=ch, device="cpu")
    got, rows, kpad = c_pack(w, layout=1, cin_pad=cin_pad, chain=chain)
    assert (rows, kpad) == (pc.w_rows, pc.w_kpad)
    assert np.array_equal(got, bits(pc.weight))
    if chain:
        lib = _lib.load()
        w2 = np.ascontiguousarray(ch[0].reshape(48, cout).numpy())
        dw = np.zeros(48 * cout, np.uint16)
        ds, dt = np.zeros(48, np.float32), np.zeros(48, np.float32)
        assert lib.v2x_pack_chain_1x1(48, cout, w2.ctypes.data, None, None, dw.ctypes.data, ds.ctypes.data, dt.ctypes.data) == 0
        assert np.array_equal(dw, bits(pc.weight2)) and np.array_equal(ds, pc.scale2.numpy()) and np.array_equal(dt, pc.shift2.numpy())


@pytest.mark.parametrize("cout,cin,chain", [(128, 128, 0), (256, 768, 0), (64, 192, 0), (64, 64, 1), (128, 128, 1), (512, 512, 0)])
def test_layout2_stream(cout, cin, chain):
    w = rnd(cout, cin, 3, 3, seed=cout + 2 * cin)
    ch = (rnd(cout, cout, 1, 1, seed=9), torch.ones(cout), torch.zeros(cout), True) if chain else None
    pc = packing.pack_conv_stream("t", w, torch.ones(cout), torch.zeros(cout), chain=ch, device="cpu")
    got, rows, kpad = c_pack(w, layout=2, chain=chain)
    assert (rows, kpad) == (pc.w_rows, pc.w_kpad)
    assert np.array_equal(got, bits(pc.weight))


@pytest.mark.parametrize("cout,c_up,c1", [(32, 64, 32), (64, 128, 64), (32, 32, 32)])
def test_layout3_parity_class(cout, c_up, c1):
    """Parity-class packing (pre-summed 2x2-tap weights for the upsampled source): C packer == torch packer bit for bit -- the fp32 sums are
    added in the same (ky, kx) order and rounded once -- and the class sums reproduce the 9-tap layer on a nearest-upsampled operand."""
    w = rnd(cout, c_up + c1, 3, 3, seed=cout + c_up)
    pc = packing.pack_conv_halo_parity("t", w, torch.ones(cout), torch.zeros(cout), C0=c_up, C1=c1, device="cpu")
    got, rows, kpad = c_pack(w, layout=3, c_up=c_up)
    assert (rows, kpad) == (pc.w_rows, pc.w_kpad) == (cout, 16 * c_up + 9 * c1)
    assert np.array_equal(got, bits(pc.weight))
    # the identity the layout rests on, in fp64: four 2x2-tap class convolutions on the half-resolution map == 3x3 on the upsampled map
    import torch.nn.functional as F
    wc = packing.parity_class_weights(w[:, :c_up].double())
    lo = torch.randn(2, c_up, 5, 7, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    ref = F.conv2d(F.interpolate(lo, scale_factor=(2, 2)), w[:, :c_up].double(), None, 1, 1)
    out = torch.zeros_like(ref)
    lop = F.pad(lo, (1, 1, 1, 1))
    for py in range(2):
        for px in range(2):
            k = wc[py * 2 + px].view(2, 2, cout, c_up).permute(2, 3, 0, 1).contiguous()
            out[:, :, py::2, px::2] = F.conv2d(lop, k)[:, :, py:py + 5, px:px + 7]
    assert float((out - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    # unsupported specs are refused with a message
    lib = _lib.load()
    bad = PackSpec(Cout=32, Cin=96, ksize=3, cin_pad=0, w_layout=3, epilogue=V2X_EPI_BF16, chain=0, c_up=0)
    assert lib.v2x_pack_conv_size(C.byref(bad), None, None) == 0 and b"layout 3" in lib.v2x_last_error()


@pytest.mark.parametrize("cout,c_up,c1", [(256, 512, 256), (128, 256, 128), (128, 64, 32), (64, 128, 64)])
def test_layout4_streamed_parity_class(cout, c_up, c1):
    """The streamed parity-class layout (conv_stream_pc.hip): C packer == torch packer bit for bit, zero page included."""
    w = rnd(cout, c_up + c1, 3, 3, seed=cout + c_up + 1)
    pc = packing.pack_conv_stream_parity("t", w, torch.ones(cout), torch.zeros(cout), C0=c_up, C1=c1, device="cpu")
    got, rows, kpad = c_pack(w, layout=4, c_up=c_up)
    assert (rows, kpad) == (pc.w_rows, pc.w_kpad) == (cout, 16 * c_up + 9 * c1)
    assert got.size == cout * kpad + 32 and not got[-32:].any()
    assert np.array_equal(got, bits(pc.weight))
    lib = _lib.load()
    bad = PackSpec(Cout=96, Cin=96, ksize=3, cin_pad=0, w_layout=4, epilogue=V2X_EPI_BF16, chain=0, c_up=64)
    assert lib.v2x_pack_conv_size(C.byref(bad), None, None) == 0 and b"layout 4" in lib.v2x_last_error()


def test_gru_layouts_and_bias():
    hid, cin = 256, 512
    w = rnd(3 * hid, cin, 3, 3, seed=1)
    bi, bh = rnd(3 * hid, seed=2), rnd(3 * hid, seed=3)
    p0 = packing.pack_gru("g", w, bi, bh, C0=hid, C1=hid, device="cpu")
    p2 = packing.pack_gru_stream("g", w, bi, bh, C0=hid, C1=hid, device="cpu")
    g0, r0, k0 = c_pack(w, layout=0, gru=True)
    g2, r2, k2 = c_pack(w, layout=2, gru=True)
    assert (r0, k0) == (p0.w_rows, p0.w_kpad) and np.array_equal(g0, bits(p0.weight))
    assert (r2, k2) == (p2.w_rows, p2.w_kpad) and np.array_equal(g2, bits(p2.weight))
    lib = _lib.load()
    dst = np.zeros(4 * hid, np.float32)
    assert lib.v2x_pack_gru_bias(hid, bi.numpy().ctypes.data, bh.numpy().ctypes.data, dst.ctypes.data) == 0
    assert np.array_equal(dst, p0.scale.numpy().reshape(-1)) and np.array_equal(dst, p2.scale.numpy().reshape(-1))


def test_fold_bn_correctly_rounded():
    """scale = gamma / sqrt(var + eps), shift = beta + scale*(bias - mean), every op rounded to fp32 separately.  The C packer
    is IEEE-exact (checked op by op against float64).  torch's vectorised CPU sqrt is off by one ulp in 1-15 % of the elements depending on the
    CPU model, so packing.fold_bn CALLS the library (round 5: one definition for every host -- a C host's packed parameters are then
    bit-identical to the Python host's on any machine, tests/test_c_abi.py) and is held to bit equality; the plain torch expression stays
    within 2 ulp."""
    C_ = 96
    conv = torch.nn.Conv2d(8, C_, 3)
    bn = torch.nn.BatchNorm2d(C_)
    with torch.no_grad():
        for t in (conv.bias, bn.weight, bn.bias, bn.running_mean):
            t.copy_(rnd(C_, seed=int(t.numel()) + 4))
        bn.running_var.copy_(rnd(C_, seed=8).abs() + 0.05)
    s, t = packing.fold_bn(conv.bias, bn, C_)
    lib = _lib.load()
    ds, dt = np.full(128, 9.0, np.float32), np.full(128, 9.0, np.float32)
    a = [x.detach().numpy() for x in (conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    assert lib.v2x_fold_bn(C_, 128, *[x.ctypes.data for x in a], C.c_float(bn.eps), ds.ctypes.data, dt.ctypes.data) == 0
    f32, f64 = np.float32, np.float64
    g_, b_, mu_, var_, cb_ = a[1], a[2], a[3], a[4], a[0]
    v = (var_.astype(f64) + f64(f32(bn.eps))).astype(f32)
    want_s = (g_.astype(f64) / np.sqrt(v.astype(f64)).astype(f32).astype(f64)).astype(f32)
    want_t = (b_.astype(f64) + (want_s.astype(f64) * (cb_.astype(f64) - mu_.astype(f64)).astype(f32).astype(f64)).astype(f32).astype(f64)).astype(f32)
    assert np.array_equal(ds[:C_], want_s) and np.array_equal(dt[:C_], want_t)
    assert np.array_equal(ds[:C_], s.numpy()) and np.array_equal(dt[:C_], t.numpy())
    s_torch = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    ulp = np.abs(ds[:C_].view(np.int32) - s_torch.numpy().view(np.int32))
    assert ulp.max() <= 2   # a 1-ulp sqrt error can double through the division
    assert not ds[C_:].any() and not dt[C_:].any()
    assert lib.v2x_fold_bn(C_, 128, a[0].ctypes.data, None, None, None, None, C.c_float(0), ds.ctypes.data, dt.ctypes.data) == 0
    assert np.array_equal(ds[:C_], np.ones(C_, np.float32)) and np.array_equal(dt[:C_], a[0])


def test_unsupported_specs_are_refused():
    lib = _lib.load()
    for kw in (dict(Cout=48, Cin=32, ksize=3, w_layout=1), dict(Cout=64, Cin=48, ksize=3, w_layout=2),
               dict(Cout=64, Cin=64, ksize=1, w_layout=2), dict(Cout=64, Cin=64, ksize=3, w_layout=7),
               dict(Cout=64, Cin=64, ksize=3, w_layout=0, chain=1), dict(Cout=40, Cin=64, ksize=3, w_layout=0, epilogue=V2X_EPI_GRU)):
        spec = PackSpec(**kw)
        assert lib.v2x_pack_conv_size(C.byref(spec), None, None) == 0 and lib.v2x_last_error()
