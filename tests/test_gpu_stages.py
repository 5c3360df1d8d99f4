"""GPU parity tests, one block per SURVEY.md section 8 row, calling the HIP kernels through the
C ABI (v2x_sim_amd.ops -> libv2x_amd.so) and checking against the build-owned CPU oracle and
the committed golden vectors.  PARITY UNPINNED w.r.t. the reference (no code in /root/reference).

Bars: bit-exact for voxel indices / occupancy / confusion matrices; for floating point the
tolerance is written next to each assert (bf16 storage, fp32 accumulation)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import coperception_ref as R
from oracle import voxelize_ref as VR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def to_nhwc_bf16(x_nchw, dev):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------- a1
def _voxel_gpu(pts_list, dev, grid=None):
    from v2x_sim_amd import ops
    grid = grid or ops.VoxelGrid()
    n = len(pts_list)
    mp = max(1, max(p.shape[0] for p in pts_list))
    buf = np.zeros((n, mp, 4), np.float32)
    cnt = np.zeros((n,), np.int32)
    for i, p in enumerate(pts_list):
        buf[i, :p.shape[0]] = p
        cnt[i] = p.shape[0]
    bits = ops.voxelize_bits(torch.from_numpy(buf).to(dev), torch.from_numpy(cnt).to(dev), grid)
    return bits, grid


@pytest.mark.parametrize("voxel", [(0.25, 0.25, 0.4), (0.3, 0.5, 0.5), (0.4, 0.125, 0.25), (1.0, 0.7, 2.0)])
def test_voxelize_power_of_two_voxels_take_the_exact_reciprocal(device, voxel, tune):
    """Per axis the quotient fp64(p) / fp64(voxel) is formed by the exact reciprocal when the voxel size is a power of two (one multiply instead of an
    IEEE division) and by the division otherwise: both are the correctly rounded quotient, so every mix of axes is bit-exact against the oracle's
    numpy fp64 division -- LDS form and global-atomic form, points on and next to voxel boundaries included."""
    from v2x_sim_amd import ops
    ext = ((-16.0, 16.0), (-16.0, 16.0), (-3.0, 2.0))
    grid = ops.VoxelGrid(voxel_size=voxel, area_extents=ext)
    rng = np.random.default_rng(int(sum(voxel) * 1000))
    pts = np.zeros((3, 20000, 4), np.float32)
    pts[..., :2] = rng.uniform(-17, 17, (3, 20000, 2))
    pts[..., 2] = rng.uniform(-3.5, 2.5, (3, 20000))
    # exact multiples of the voxel size and their float32 neighbours: the floor must fall on the same side as numpy's
    k = rng.integers(-40, 40, (3, 4000, 3)).astype(np.float64) * np.asarray(voxel)
    edge = k.astype(np.float32)
    pts[:, :4000, :3] = edge
    pts[:, 4000:8000, :3] = np.nextafter(edge, np.float32(np.inf))
    pts[:, 8000:12000, :3] = np.nextafter(edge, np.float32(-np.inf))
    for mode in (2, 0):                                     # the LDS form at any cloud count, the global-atomic form
        tune("VOXELIZE_LDS", mode)
        bits, _ = _voxel_gpu([p for p in pts], device, grid)
        got = ops.bits_to_dense(bits, grid.dims[2]).cpu().numpy()
        for i in range(3):
            ref = VR.voxelize_occupy(pts[i], np.asarray(voxel), np.asarray(ext))
            assert got[i].shape == ref.shape and np.array_equal(got[i], ref), (voxel, mode, i)


@pytest.mark.parametrize("sizes", [(65536, 65536, 65536, 65536, 65536), (4096, 0, 17, 300), (2048,)])
def test_voxelize_bit_exact(device, sizes):
    from v2x_sim_amd import ops
    clouds = [VR.synthetic_points(n, seed=100 + i, n_edge=min(64, n // 4)) if n else np.zeros((0, 4), np.float32)
              for i, n in enumerate(sizes)]
    bits, grid = _voxel_gpu(clouds, device)
    X, Y, Z = grid.dims
    assert (X, Y, Z) == (256, 256, 13)
    dense = ops.bits_to_dense(bits, Z).cpu().numpy()
    cap = 65536
    idx, counts = ops.bits_to_indices(bits, Z, cap)
    idx, counts = idx.cpu().numpy(), counts.cpu().numpy()
    nhwc = ops.bits_to_nhwc(bits, Z, 16).float().cpu().numpy()
    for i, pts in enumerate(clouds):
        ref_grid, ref_idx = VR.voxelize_occupy(pts, return_indices=True)
        assert np.array_equal(dense[i], ref_grid)                      # bit-exact occupancy
        assert counts[i] == ref_idx.shape[0]
        assert np.array_equal(idx[i, :counts[i]], ref_idx.astype(np.int32))  # bit-exact, same order
        assert np.array_equal(nhwc[i, :, :, :13], ref_grid) and nhwc[i, :, :, 13:].sum() == 0


@pytest.mark.parametrize("mode", [2, 0])
def test_voxelize_ring_sweep_bit_exact(device, tune, mode):
    """a REALISTIC sweep instead of uniform points -- 32 beams x 2 048 azimuth steps (utils/synthetic.py::synthetic_ring_sweep): the steep
    inner rings pile tens to hundreds of returns into one 0.25 m cell (duplicates colliding on one LDS word / one global atomic), lost returns are parked
    outside the extents.  Bit-exact occupancy and indices against the numpy oracle for the LDS-binned form (2) and the global-atomic form (0), and the
    two forms agree bit for bit."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils.synthetic import synthetic_ring_sweep
    tune("VOXELIZE_LDS", mode)
    clouds = list(synthetic_ring_sweep(5, 65536, seed=31))
    bits, grid = _voxel_gpu(clouds, device)
    Z = grid.dims[2]
    dense = ops.bits_to_dense(bits, Z).cpu().numpy()
    idx, counts = ops.bits_to_indices(bits, Z, 65536)
    idx, counts = idx.cpu().numpy(), counts.cpu().numpy()
    for i, pts in enumerate(clouds):
        ref_grid, ref_idx = VR.voxelize_occupy(pts, return_indices=True)
        cells = np.unique(np.floor(pts[:, :2].astype(np.float64) / 0.25), axis=0, return_counts=True)[1]
        assert cells.max() >= 50, "the sweep has no near-range duplicates"          # the property this test exists for
        assert np.array_equal(dense[i], ref_grid)
        assert counts[i] == ref_idx.shape[0] and np.array_equal(idx[i, :counts[i]], ref_idx.astype(np.int32))


def test_voxelize_golden_and_idempotent(device):
    from v2x_sim_amd import ops
    g = np.load(os.path.join(GOLD, "voxel_2048.npz"))
    bits, grid = _voxel_gpu([g["points"]], device)
    idx, counts = ops.bits_to_indices(bits, 13, 4096)
    assert np.array_equal(idx[0, :int(counts[0])].cpu().numpy(), g["indices"])
    # scattering the same cloud twice (duplicates of every point) changes nothing
    twice = np.concatenate([g["points"], g["points"]])
    bits2, _ = _voxel_gpu([twice], device)
    assert torch.equal(bits, bits2)


def test_voxelize_edges_and_cross_road_extents(device):
    from v2x_sim_amd import ops
    on = np.array([[-32, 0, 0, 0], [32, 0, 0, 0], [0, -32, 0, 0], [0, 32, 0, 0], [0, 0, -3, 0], [0, 0, 2, 0],
                   [0.25, -0.25, np.float32(0.4), 0], [0, 0, np.float32(1.2), 0]], np.float32)
    bits, _ = _voxel_gpu([on], device)
    idx, counts = ops.bits_to_indices(bits, 13, 16)
    assert int(counts[0]) == 2
    assert idx[0, :2].cpu().tolist() == [[128, 128, 11], [129, 127, 9]]
    grid = ops.VoxelGrid(area_extents=((-32, 32), (-32, 32), (-8, -3)))
    pts = VR.synthetic_points(8192, seed=3)
    pts[:, 2] -= 5.0
    bits, _ = _voxel_gpu([pts], device, grid)
    ext = np.array([[-32.0, 32.0], [-32.0, 32.0], [-8.0, -3.0]])
    ref = VR.voxelize_occupy(pts, VR.VOXEL_SIZE, ext)
    assert np.array_equal(ops.bits_to_dense(bits, 13)[0].cpu().numpy(), ref)


def test_voxelize_lds_form_equals_atomic_form(device, tune):
    """The LDS-binned kernel (default) and the global-atomic scatter kernel are bit-identical, for the packed float4
    cloud, a stride-5 cloud (scalar loads), ragged counts incl. 0 and a grid that is too big for the LDS (fallback)."""
    from v2x_sim_amd import ops
    rng = np.random.default_rng(5)
    sizes = (65536, 0, 1, 4097, 30000)
    clouds = [VR.synthetic_points(n, seed=200 + i, n_edge=min(64, n // 4)) if n else np.zeros((0, 4), np.float32)
              for i, n in enumerate(sizes)]
    # adversarial cloud: every voxel boundary of every axis, and its two fp32 neighbours (strict bounds, floor of the
    # fp64 quotient: where any arithmetic shortcut would show)
    kx = np.arange(-130, 131)
    bx = (kx * 0.25).astype(np.float32)
    bz = (np.arange(-9, 7) * 0.4).astype(np.float32)
    edge = []
    for b, axis in ((bx, 0), (bx, 1), (bz, 2)):
        for v in (b, np.nextafter(b, np.float32(np.inf)), np.nextafter(b, np.float32(-np.inf))):
            e = np.zeros((v.size, 4), np.float32)
            e[:, 0], e[:, 1], e[:, 2] = 0.1, -0.1, 0.1
            e[:, axis] = v
            edge.append(e)
    clouds[2] = np.concatenate(edge)
    sizes = tuple(c.shape[0] for c in clouds)
    mp = max(sizes)
    cnt = torch.tensor(sizes, dtype=torch.int32, device=device)
    for stride in (4, 5):
        buf = rng.uniform(-40, 40, (len(sizes), mp, stride)).astype(np.float32)   # garbage past n_pts must be ignored
        for i, c in enumerate(clouds):
            buf[i, :c.shape[0], :4] = c
        pts = torch.from_numpy(buf).to(device)
        grid = ops.VoxelGrid()
        tune("VOXELIZE_LDS", 2)      # 2 = the LDS-binned form at any cloud count (1 = default: the global-atomic form below 49 clouds)
        lds = ops.voxelize_bits(pts, cnt, grid).clone()
        tune("VOXELIZE_LDS", 0)
        atm = ops.voxelize_bits(pts, cnt, grid).clone()
        tune.reset("VOXELIZE_LDS")
        assert torch.equal(lds, atm)
        dense = ops.bits_to_dense(lds, 13).cpu().numpy()
        for i, c in enumerate(clouds):
            assert np.array_equal(dense[i], VR.voxelize_occupy(c))
    # 0.125 m voxels: 512 x 512 grid = 512 KiB of 16-bit words, does not fit the LDS -> atomic form, same spec
    big = ops.VoxelGrid(voxel_size=(0.125, 0.125, 0.4))
    pts = torch.from_numpy(np.stack([clouds[0]])).to(device)
    b = ops.voxelize_bits(pts, cnt[:1], big)
    ref = VR.voxelize_occupy(clouds[0], np.array([0.125, 0.125, 0.4]))
    assert np.array_equal(ops.bits_to_dense(b, 13)[0].cpu().numpy(), ref)


def test_dense_to_nhwc(device):
    from v2x_sim_amd import ops
    bev = (torch.rand(3, 32, 48, 13) < 0.1).float()
    out = ops.dense_to_nhwc(bev.to(device), 16).float().cpu()
    assert torch.equal(out[..., :13], bev) and out[..., 13:].abs().sum() == 0


# ------------------------------------------------------------------------------------- a2/a6/a7
def _run_conv(dev, x, w, scale, shift, *, stride=1, relu=True, x_skip=None, up0=0, f32=False, cin_pad=None):
    """x (N,C0,h,w) [+ x_skip (N,C1,H,W)] NCHW fp32 (bf16-representable) -> NCHW fp32 result."""
    from v2x_sim_amd import ops, packing
    C0 = x.shape[1] if cin_pad is None else cin_pad
    C1 = 0 if x_skip is None else x_skip.shape[1]
    pc = packing.pack_conv("t", w, scale, shift, stride=stride, C0=C0, C1=C1, up0=up0, relu=relu,
                           epilogue=ops.V2X_EPI_F32 if f32 else ops.V2X_EPI_BF16, cin_pad=cin_pad, device=dev)
    xin = x if cin_pad is None else F.pad(x, (0, 0, 0, 0, 0, cin_pad - x.shape[1]))
    y = ops.conv2d(pc, to_nhwc_bf16(xin, dev), None if x_skip is None else to_nhwc_bf16(x_skip, dev))
    return from_nhwc(y)


def _ref_conv(x, w, scale, shift, stride, relu, x_skip=None, up0=0):
    if x_skip is not None:
        up = F.interpolate(x, scale_factor=(2, 2)) if up0 else x
        x = torch.cat((up, x_skip), 1)
    y = F.conv2d(x, bf16r(w), None, stride, (w.shape[-1] - 1) // 2)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    return F.relu(y) if relu else y


CONV_CASES = [
    # (N, Cin, Cout, H, W, k, stride)  -- covers every tile config (32/48/64/128 rows) and K tails
    (2, 16, 32, 24, 40, 3, 1),    # first-layer shape class (Cin=16 -> 4 taps per K chunk)
    (1, 32, 32, 33, 19, 3, 1),    # odd extents, M tail
    (2, 32, 64, 32, 32, 3, 2),    # stride 2
    (1, 64, 64, 20, 20, 1, 1),    # 1x1 ("Conv3D")
    (1, 64, 128, 16, 16, 3, 2),
    (3, 128, 256, 8, 8, 3, 1),    # 128-row tiles, 2 channel tiles
    (1, 256, 48, 8, 12, 1, 1),    # 48-row tile
    (1, 512, 512, 4, 4, 3, 1),    # deep K (72 chunks), tiny M
    (5, 32, 12, 16, 16, 1, 1),    # Cout not a multiple of 16
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_vs_torch(device, case):
    N, Cin, Cout, H, W, k, stride = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf16r(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.2
    ref = _ref_conv(x, w, scale, shift, stride, True)
    got32 = _run_conv(device, x, w, scale, shift, stride=stride, relu=True, f32=True)
    # fp32 epilogue: only the accumulation order differs -> 1e-4 abs on O(1) values
    assert got32.shape == ref.shape
    assert torch.allclose(got32, ref, atol=2e-4, rtol=1e-4), float((got32 - ref).abs().max())
    got = _run_conv(device, x, w, scale, shift, stride=stride, relu=True)
    # bf16 epilogue: one bf16 rounding of the same value (rel 2^-8) on top
    assert torch.allclose(got, ref, atol=2e-3, rtol=2 ** -7), float((got - ref).abs().max())


def test_conv_transpose_detecting(device):
    """Asymmetric check (guide rule 16): single hot input pixel/channel -> the weight slice appears,
    spatially flipped, at the right place and channel order."""
    N, Cin, Cout, H, W = 1, 32, 64, 16, 16
    x = torch.zeros(N, Cin, H, W)
    x[0, 5, 7, 3] = 1.0
    w = torch.arange(Cout * Cin * 9, dtype=torch.float32).view(Cout, Cin, 3, 3) % 251 / 64.0
    ref = _ref_conv(x, w, torch.ones(Cout), torch.zeros(Cout), 1, False)
    got = _run_conv(device, x, w, torch.ones(Cout), torch.zeros(Cout), relu=False, f32=True)
    assert torch.equal(got, ref)


def test_conv_first_layer_channel_pad(device):
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(2, 13, 32, 32, generator=g) < 0.1).float()
    w = torch.randn(32, 13, 3, 3, generator=g) * 0.2
    ref = _ref_conv(x, w, torch.ones(32), torch.zeros(32), 1, True)
    got = _run_conv(device, x, w, torch.ones(32), torch.zeros(32), cin_pad=16, f32=True)
    assert torch.allclose(got, ref, atol=1e-4)


@pytest.mark.parametrize("cup,cskip,cout", [(64, 32, 32), (128, 64, 64), (512, 256, 256)])
def test_conv_upsample_concat(device, cup, cskip, cout):
    g = torch.Generator().manual_seed(cup)
    H, W = (8, 8) if cup == 512 else (16, 24)
    x_up = bf16r(torch.randn(2, cup, H // 2, W // 2, generator=g))
    x_sk = bf16r(torch.randn(2, cskip, H, W, generator=g))
    w = torch.randn(cout, cup + cskip, 3, 3, generator=g) * (2.0 / ((cup + cskip) * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = _ref_conv(x_up, w, scale, shift, 1, True, x_skip=x_sk, up0=1)
    got = _run_conv(device, x_up, w, scale, shift, x_skip=x_sk, up0=1, f32=True)
    assert torch.allclose(got, ref, atol=3e-4, rtol=1e-4), float((got - ref).abs().max())


def test_conv_golden(device):
    g = np.load(os.path.join(GOLD, "conv_small.npz"))
    x, w = torch.from_numpy(g["x"]), torch.from_numpy(g["w"])
    scale, shift = torch.from_numpy(g["scale"]), torch.from_numpy(g["shift"])
    for s in (1, 2):
        got = _run_conv(device, x, w, scale, shift, stride=s, f32=True)
        assert torch.allclose(got, torch.from_numpy(g["y_s%d" % s]), atol=2e-4)
    got = _run_conv(device, torch.from_numpy(g["x_up"]), torch.from_numpy(g["w2"]), scale, shift,
                    x_skip=x, up0=1, f32=True)
    assert torch.allclose(got, torch.from_numpy(g["y_upcat"]), atol=2e-4)


def test_conv_split_outputs(device):
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(9)
    x = bf16r(torch.randn(2, 64, 16, 16, generator=g))
    w = torch.randn(48, 64, 1, 1, generator=g) * 0.2
    b = torch.randn(48, generator=g)
    pc = packing.pack_conv("t", w, torch.ones(48), b, relu=False, epilogue=ops.V2X_EPI_F32, device=device)
    a, c = ops.conv2d(pc, to_nhwc_bf16(x, device), split=12)
    ref = _ref_conv(x, w, torch.ones(48), b, 1, False)
    assert a.shape == (2, 16, 16, 12) and c.shape == (2, 16, 16, 36) and a.is_contiguous() and c.is_contiguous()
    assert torch.allclose(from_nhwc(a), ref[:, :12], atol=2e-4) and torch.allclose(from_nhwc(c), ref[:, 12:], atol=2e-4)


# ------------------------------------------------------------------------------------- a4
def test_gru_golden_and_random(device):
    from v2x_sim_amd import ops, packing
    g = np.load(os.path.join(GOLD, "gru_32.npz"))
    x = torch.from_numpy(g["x"])  # (1, 64, 8, 8): first 32 = ego, last 32 = neighbour mean
    pc = packing.pack_gru("gru", torch.from_numpy(g["w_ih"]), torch.from_numpy(g["b_ih"]),
                          torch.from_numpy(g["b_hh"]), C0=32, C1=32, device=device)
    h = ops.conv2d(pc, to_nhwc_bf16(x[:, :32], device), to_nhwc_bf16(x[:, 32:], device))
    got = from_nhwc(h)
    # same bf16-rounded operands, fp32 accumulate, bf16 output: <= 1 bf16 ulp of an O(1) value
    assert torch.allclose(got, torch.from_numpy(g["h"]), atol=2 ** -8, rtol=2 ** -7)
    # fp32 spec (un-rounded weights): bf16 weight rounding adds ~1e-2 on |h| <= 1
    assert torch.allclose(got, torch.from_numpy(g["h_fp32"]), atol=3e-2)
    # full-size cell, 256 hidden, several maps
    torch.manual_seed(5)
    cell = R.Conv2dGRUCell(512, 256, 3)
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for p in cell.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
        xx = bf16r(torch.randn(3, 512, 32, 32, generator=gen))
        ref = cell(xx, None, emulate=True)
    pc = packing.pack_gru("gru", cell.weight_ih_l0, cell.bias_ih_l0, cell.bias_hh_l0, C0=256, C1=256, device=device)
    h = ops.conv2d(pc, to_nhwc_bf16(xx[:, :256], device), to_nhwc_bf16(xx[:, 256:], device))
    assert torch.allclose(from_nhwc(h), ref, atol=2 ** -7, rtol=2 ** -7), float((from_nhwc(h) - ref).abs().max())


# ------------------------------------------------------------------------------------- a3
def _warp_ref(feat, T, items, coef, A, Bt, mode):
    n_out = len(items)
    C, H, W = feat.shape[1:]
    out = torch.zeros(n_out, C, H, W)
    for m, (ego, f) in enumerate(items):
        acc = torch.zeros(C, H, W)
        cnt = 0
        for j in range(A):
            c = float(coef[m, j])
            if c == 0:
                continue
            cnt += 1
            v = feat[j * Bt + f] if j == ego else R.feature_transformation(feat[j * Bt + f], T[f, ego, j], (1, C, H, W))
            if mode == 2:
                acc = v if cnt == 1 else torch.maximum(acc, v)
            else:
                acc = acc + (v if mode == 1 else c * v)
        out[m] = acc / cnt if (mode == 1 and cnt) else acc
    return out


def test_warp_golden_and_identity(device):
    from v2x_sim_amd import ops
    g = np.load(os.path.join(GOLD, "warp_2agent.npz"))
    feat = torch.from_numpy(g["feat"])
    T = torch.zeros(1, 2, 2, 4, 4)
    T[0, 0, 1] = torch.from_numpy(g["T"])
    T[0, 1, 0] = torch.eye(4)
    items = torch.tensor([[0, 0], [1, 0]], dtype=torch.int32, device=device)
    coef = torch.tensor([[0.0, 1.0], [1.0, 0.0]], device=device)
    out = ops.warp_fuse(to_nhwc_bf16(feat, device), 2, 1, T.to(device), items, coef, ops.V2X_FUSE_MEAN)
    got = from_nhwc(out)
    # fp32 interpolation of bf16 maps, output rounded to bf16: 1 ulp (2^-8 rel) + 1e-3 abs
    assert torch.allclose(got[0], bf16r(torch.from_numpy(g["warped"])), atol=2e-3, rtol=2 ** -7)
    # identity pose reproduces the neighbour map exactly (bilinear weights are exactly 1/0)
    assert torch.allclose(got[1], feat[0], atol=1e-6)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_warp_fuse_vs_oracle(device, mode):
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    A, Bt, C, H, W = 5, 2, 64, 32, 32
    g = torch.Generator().manual_seed(21 + mode)
    feat = bf16r(torch.randn(A * Bt, C, H, W, generator=g))
    T = torch.from_numpy(synthetic_poses(Bt, A, seed=4))
    items = [(a, f) for a in range(A) for f in range(Bt)]
    if mode == 2:       # MaxFusion: ego (unwarped) and neighbours, the warped maps' zero padding takes part in the max
        coef = torch.ones(len(items), A)
        coef[3, 4] = 0
    elif mode == 1:
        coef = torch.ones(len(items), A)
        for m, (a, f) in enumerate(items):
            coef[m, a] = 0
        coef[3, 4] = 0  # a frame with fewer neighbours
    else:
        coef = torch.rand(len(items), A, generator=g)
        coef[coef < 0.3] = 0
    ref = _warp_ref(feat, T, items, coef, A, Bt, mode)
    out = ops.warp_fuse(to_nhwc_bf16(feat, device), A, Bt, T.to(device),
                        torch.tensor(items, dtype=torch.int32, device=device), coef.to(device), mode)
    got = from_nhwc(out)
    assert torch.allclose(got, bf16r(ref), atol=4e-3, rtol=2 ** -7), float((got - ref).abs().max())


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_warp_fuse_lds_form_bitwise_and_oracle(device, mode, tune):
    """C = 256 (the fusion layer of the benchmark config) takes the LDS-staged kernel: bit-identical to the direct kernel
    and within the usual tolerance of the oracle -- incl. poses that leave the map, sub-pixel and exactly-integer-pixel
    translations (the window-origin rounding corner) and pure rotations."""
    import math
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    A, Bt, C, H, W = 5, 3, 256, 32, 32
    g = torch.Generator().manual_seed(31 + mode)
    feat = bf16r(torch.randn(A * Bt, C, H, W, generator=g))
    T = torch.from_numpy(synthetic_poses(Bt, A, seed=6))
    # frame 1: hand-made poses.  translation of k pixels = k * 2 (the warp shifts by 4*T/128 in normalised units = T/2 px)
    eye = torch.eye(4)
    def pose(yaw, tx, ty):
        M = eye.clone()
        M[0, 0], M[0, 1], M[1, 0], M[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
        M[0, 3], M[1, 3] = tx, ty
        return M
    special = [pose(0.0, 2.0, -4.0), pose(0.0, 1e-3, -1e-3), pose(0.7, 0.0, 0.0), pose(0.0, 80.0, 3.0), pose(3.1, -31.0, 62.0),
               pose(0.0, 6.0, 6.0), pose(-1.2, 15.9999, -16.0001)]
    k = 0
    for i in range(A):
        for j in range(A):
            if i != j:
                T[1, i, j] = special[k % len(special)]
                k += 1
    items = [(a, f) for a in range(A) for f in range(Bt)]
    if mode == 2:
        coef = torch.ones(len(items), A)
        coef[3, 4] = 0
    elif mode == 1:
        coef = torch.ones(len(items), A)
        for m, (a, f) in enumerate(items):
            coef[m, a] = 0
        coef[3, 4] = 0
    else:
        coef = torch.rand(len(items), A, generator=g)
        coef[coef < 0.3] = 0
    x = to_nhwc_bf16(feat, device)
    it = torch.tensor(items, dtype=torch.int32, device=device)
    tune("WARP_LDS", 1)
    lds = ops.warp_fuse(x, A, Bt, T.to(device), it, coef.to(device), mode).clone()
    tune("WARP_LDS", 2)     # (round 5, the default) the rotate set-ups of a window shared through an LDS table: the same arithmetic
    lds2 = ops.warp_fuse(x, A, Bt, T.to(device), it, coef.to(device), mode).clone()
    tune("WARP_LDS", 0)
    direct = ops.warp_fuse(x, A, Bt, T.to(device), it, coef.to(device), mode).clone()
    tune.reset("WARP_LDS")
    assert torch.equal(lds.view(torch.int16), direct.view(torch.int16))
    assert torch.equal(lds2.view(torch.int16), direct.view(torch.int16))
    ref = _warp_ref(feat, T, items, coef, A, Bt, mode)
    got = from_nhwc(lds)
    assert torch.allclose(got, bf16r(ref), atol=4e-3, rtol=2 ** -7), float((got - ref).abs().max())


@pytest.mark.parametrize("ragged", [False, True])
def test_warp_fuse_frame_ordered_launch_is_bit_identical(device, ragged, tune):
    """WARP_XCD: v2x_warp_fuse_ordered computes the output maps of one frame on one XCD (a permutation of which workgroup computes which tile):
    identical bits to the plain grid, with every (ego, frame) present and with a ragged item list (frames with fewer agents, shuffled order,
    19 frames = not a multiple of the 8 XCDs)."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    A, Bt, C, H, W = 5, 19, 256, 32, 32
    g = torch.Generator().manual_seed(77)
    x = torch.randn(A * Bt, H, W, C, generator=g).to(torch.bfloat16).to(device)
    T = torch.from_numpy(synthetic_poses(Bt, A, seed=3)).to(device)
    items = [(a, f) for a in range(A) for f in range(Bt)]
    if ragged:
        items = [(a, f) for (a, f) in items if not (f % 3 == 1 and a >= 3)]
        perm = torch.randperm(len(items), generator=g).tolist()
        items = [items[i] for i in perm]
    coef = torch.rand(len(items), A, generator=g)
    coef[coef < 0.25] = 0
    it = torch.tensor(items, dtype=torch.int32, device=device)
    for mode in (0, 1):
        tune("WARP_XCD", 0)
        ref = ops.warp_fuse(x, A, Bt, T, it, coef.to(device), mode).clone()
        tune("WARP_XCD", 1)
        got = ops.warp_fuse(x, A, Bt, T, it, coef.to(device), mode)
        assert hasattr(it, "_v2x_frame_order")
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # an (ego, frame) pair listed twice cannot be ordered by the table: the plain grid is taken and both outputs are computed
    dup = torch.tensor(items[:7] + items[:3], dtype=torch.int32, device=device)
    cd = torch.rand(10, A, generator=g).to(device)
    tune("WARP_XCD", 1)
    got = ops.warp_fuse(x, A, Bt, T, dup, cd, 0)
    assert dup._v2x_frame_order[1] is False
    tune("WARP_XCD", 0)
    assert torch.equal(ops.warp_fuse(x, A, Bt, T, dup, cd, 0), got)


# ------------------------------------------------------------------------------------- a5
def test_attention_golden(device):
    from v2x_sim_amd import ops
    g = np.load(os.path.join(GOLD, "attn_5x5.npz"))
    d = lambda k: torch.from_numpy(g[k]).to(device)
    prob, coef = ops.attn_handshake(d("keys"), d("querys"), d("w"), d("b"), 5, 2, "activated")
    assert torch.allclose(prob.cpu(), torch.from_numpy(g["prob"]), atol=1e-5)   # fp32, order of summation only
    assert torch.allclose(coef.cpu(), torch.from_numpy(g["coef_activated"]), atol=1e-5)
    _, coef = ops.attn_handshake(d("keys"), d("querys"), d("w"), d("b"), 5, 2, "argmax_test")
    assert torch.equal(coef.cpu(), torch.from_numpy(g["coef_argmax"]))
    p2, c2 = ops.attn_handshake(d("keys"), d("querys"), d("w"), d("b"), 5, 2, "softmax")
    assert torch.equal(p2, c2)


# ------------------------------------------------------------------------------------- a8
def test_seg_argmax_confusion_exact(device):
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(3, 64, 64, 8, generator=g)
    label = torch.randint(0, 8, (3, 64, 64), generator=g).to(torch.uint8)
    pred, conf = ops.seg_argmax_confusion(logits.to(device), label.to(device))
    ref_pred = logits.argmax(-1)
    assert torch.equal(pred.cpu().long(), ref_pred)
    assert torch.equal(conf.cpu(), R.confusion_matrix(ref_pred, label))   # integer-exact
    assert int(conf.sum()) == 3 * 64 * 64
    # the generic kernel (other class counts; also labels >= n_cls are ignored) and the 8-class form agree with torch on ties and ignore labels
    lg = torch.round(torch.randn(2, 16, 32, 8, generator=g) * 2) / 2             # many exact ties: the FIRST maximum wins
    lb = torch.randint(0, 10, (2, 16, 32), generator=g).to(torch.uint8)          # 8, 9 = ignore
    p8, c8 = ops.seg_argmax_confusion(lg.to(device), lb.to(device))
    assert torch.equal(p8.cpu().long(), lg.argmax(-1))
    keep = lb < 8
    assert torch.equal(c8.cpu(), R.confusion_matrix(lg.argmax(-1)[keep], lb[keep])) and int(c8.sum()) == int(keep.sum())
    lg5 = torch.randn(2, 16, 32, 5, generator=g)
    lb5 = torch.randint(0, 5, (2, 16, 32), generator=g).to(torch.uint8)
    p5, c5 = ops.seg_argmax_confusion(lg5.to(device), lb5.to(device))
    assert torch.equal(p5.cpu().long(), lg5.argmax(-1)) and int(c5.sum()) == 2 * 16 * 32
    p_only, none = ops.seg_argmax_confusion(lg.to(device), None)
    assert none is None and torch.equal(p_only, p8)


# ------------------------------------------------------------------------------------- halo-tile kernel
def _halo_ref(x, w, scale, shift, relu, x_up=None):
    if x_up is not None:
        x = torch.cat((F.interpolate(x_up, scale_factor=(2, 2)), x), 1)
    y = F.conv2d(x, bf16r(w), None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    return F.relu(y) if relu else y


@pytest.mark.parametrize("cfg", [
    # (C_up, C, Cout, N, H, W)
    (0, 32, 32, 2, 16, 64),     # conv_pre_2 / conv8_2 class; several tiles in x, image borders
    (0, 32, 32, 1, 8, 32),      # a single tile: every border is zero padding
    (64, 32, 32, 2, 16, 32),    # conv8_1: nearest-x2 upsampled source + skip
    (0, 64, 64, 3, 24, 32),     # conv7_2 class
])
def test_halo_conv_vs_torch(device, cfg):
    from v2x_sim_amd import ops, packing
    cup, c, cout, N, H, W = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = bf16r(torch.randn(N, c, H, W, generator=g))
    x_up = bf16r(torch.randn(N, cup, H // 2, W // 2, generator=g)) if cup else None
    w = torch.randn(cout, cup + c, 3, 3, generator=g) * (2.0 / ((cup + c) * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    ref = _halo_ref(x, w, scale, shift, True, x_up)
    pc = packing.pack_conv_halo("t", w, scale, shift, C0=cup if cup else c, C1=c if cup else 0, relu=True, device=device)
    if cup:
        y = ops.conv2d(pc, to_nhwc_bf16(x_up, device), to_nhwc_bf16(x, device))
    else:
        y = ops.conv2d(pc, to_nhwc_bf16(x, device))
    got = from_nhwc(y)
    assert got.shape == ref.shape
    # bf16 output: one rounding of the fp32 result
    assert torch.allclose(got, ref, atol=2e-3, rtol=2 ** -7), float((got - ref).abs().max())


@pytest.mark.parametrize("cup,c,cout,N,H,W", [(64, 32, 32, 2, 16, 32), (64, 32, 32, 3, 64, 96), (64, 32, 32, 40, 256, 256), (64, 32, 32, 1, 8, 64),
                                               (0, 64, 64, 2, 16, 32), (0, 64, 64, 3, 40, 96), (0, 64, 64, 40, 128, 128)])
def test_halo_pingpong_equals_4wave_kernel_bitwise(device, cup, c, cout, N, H, W, tune):
    """The 8-wave ping-pong form of conv8_1 / conv7_2 (two 4-wave groups on one resident weight copy, default) against the 4-wave kernel
    (V2X_HALO_PP=0): same K order and fragment mapping -> bit-identical, for one tile pair, ragged persistent walks (tile
    pairs not a multiple of the grid), image borders, and the bench's 256x256 maps (40 maps = 10 240 tiles, 20 per workgroup);
    repeated launches are bit-identical too (no race between the groups' phases)."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(N * 1000 + H + W + cup)
    x = bf16r(torch.randn(N, c, H, W, generator=g))
    x_up = bf16r(torch.randn(N, cup, H // 2, W // 2, generator=g)) if cup else None
    w = torch.randn(cout, cup + c, 3, 3, generator=g) * (2.0 / ((cup + c) * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    pc = packing.pack_conv_halo("t", w, scale, shift, C0=cup if cup else c, C1=c if cup else 0, relu=True, device=device)
    xs = to_nhwc_bf16(x, device)
    run = (lambda xu=to_nhwc_bf16(x_up, device): ops.conv2d(pc, xu, xs)) if cup else (lambda: ops.conv2d(pc, xs))
    tune("HALO_PP", 0)
    old = run().clone()
    tune("HALO_PP", 1)
    new = [run().clone() for _ in range(3)]
    tune.reset("HALO_PP")
    assert ops.conv_kernel_name(pc, H, W) == "conv3x3_halo_pp_kernel<%d, %d, %d, 0>" % (cup, c, cout)
    assert all(torch.equal(old.view(torch.int16), y.view(torch.int16)) for y in new)
    if N <= 3:
        ref = _halo_ref(x, w, scale, shift, True, x_up)
        assert torch.allclose(from_nhwc(new[0]), ref, atol=2e-3, rtol=2 ** -7)


@pytest.mark.parametrize("N,H,W", [(2, 16, 32), (3, 48, 96), (40, 128, 128)])
def test_halo_pingpong_chained_1x1(device, N, H, W):
    """conv1_2 -> conv3d_1 on the ping-pong kernel: 3x3 64 -> 64 (+BN+ReLU, rounded to bf16) chained with the 1x1 64 -> 64
    (+BN+ReLU) in the store phase.  vs torch fp32 on the same bf16 operands and vs the wide streamed kernel's chained epilogue,
    at the tolerance of tests/test_gpu_stream.py::test_stream_chain_conv1x1 (the hidden map is one bf16 rounding apart between
    implementations with different K walks, and 64 such values feed every output); bit-stable over launches."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(N + H + W)
    x = bf16r(torch.randn(N, 64, H, W, generator=g))
    w1 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / (64 * 9)) ** 0.5
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    w2 = torch.randn(64, 64, 1, 1, generator=g) * 0.2
    s2, t2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    pp = packing.pack_conv_halo("c", w1, s1, t1, relu=True, chain=(w2, s2, t2, True), device=device)
    st = packing.pack_conv_stream("c", w1, s1, t1, relu=True, chain=(w2, s2, t2, True), device=device)
    xs = to_nhwc_bf16(x, device)
    assert ops.conv_kernel_name(pp, H, W) == "conv3x3_halo_pp_kernel<0, 64, 64, 64>"
    a = ops.conv2d(pp, xs)
    assert all(torch.equal(a, ops.conv2d(pp, xs)) for _ in range(3))
    b = ops.conv2d(st, xs)
    d = (a.float() - b.float()).abs()
    assert torch.allclose(a.float(), b.float(), atol=3e-2, rtol=2 ** -6) and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))
    if N <= 3:
        hid = bf16r(_halo_ref(x, w1, s1, t1, True))
        ref = F.relu(F.conv2d(hid, bf16r(w2)) * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1))
        e = (from_nhwc(a) - ref).abs()
        assert torch.allclose(from_nhwc(a), ref, atol=3e-2, rtol=2 ** -6) and float(e.mean()) < 2e-3, (float(e.max()), float(e.mean()))


@pytest.mark.parametrize("H,W", [(24, 32), (40, 32), (8, 96)])
def test_halo_chain_odd_tile_count(device, H, W):
    """conv1_2 -> conv3d_1 is packed only as the chained halo layout, whose ping-pong kernel pairs tiles.  A map with an ODD
    number of 8x32 tiles and an odd batch gives an odd tile count: that launch takes the 4-wave form of the same layer, with the same bits
    (rows of an odd batch == the same rows inside an even batch, which the ping-pong kernel serves) -- the choice may depend on the
    batch without breaking the R-rank == 1-rank equality -- and the result is the torch reference's."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(H + W)
    x = bf16r(torch.randn(4, 64, H, W, generator=g))
    w1 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / (64 * 9)) ** 0.5
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    w2 = torch.randn(64, 64, 1, 1, generator=g) * 0.2
    s2, t2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    pp = packing.pack_conv_halo("c", w1, s1, t1, relu=True, chain=(w2, s2, t2, True), device=device)
    xs = to_nhwc_bf16(x, device)
    assert ((H // 8) * (W // 32)) % 2 == 1
    even = ops.conv2d(pp, xs)                          # 4 maps: even tile count -> ping-pong kernel
    for n in (1, 3):
        odd = ops.conv2d(pp, xs[:n].contiguous())      # odd tile count -> 4-wave kernel
        assert torch.equal(odd, even[:n]), n
    hid = bf16r(_halo_ref(x, w1, s1, t1, True))
    ref = F.relu(F.conv2d(hid, bf16r(w2)) * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1))
    e = (from_nhwc(even) - ref).abs()
    assert torch.allclose(from_nhwc(even), ref, atol=3e-2, rtol=2 ** -6) and float(e.mean()) < 2e-3, (float(e.max()), float(e.mean()))


def test_halo_equals_gather_kernel_bitwise(device):
    """Same operands, same fp32 accumulation order per output?  Not guaranteed (different K walk), so the
    two kernels are compared at 1 bf16 ulp; both against the same torch reference elsewhere."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(77)
    x = bf16r(torch.randn(2, 32, 32, 64, generator=g))
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    a = ops.conv2d(packing.pack_conv_halo("h", w, scale, shift, device=device), to_nhwc_bf16(x, device)).float()
    b = ops.conv2d(packing.pack_conv("g", w, scale, shift, device=device), to_nhwc_bf16(x, device)).float()
    assert torch.allclose(a, b, atol=1e-3, rtol=2 ** -7)
    assert float((a != b).float().mean()) < 0.01  # only accumulation-order flips of the bf16 rounding


def test_halo_chain_heads_split_f32(device):
    """det heads: 3x3 32->64 (+BN+ReLU) chained with 1x1 64->48, fp32, split into cls(12) | loc(36)."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(8)
    N, H, W = 2, 16, 64
    x = bf16r(torch.randn(N, 32, H, W, generator=g))
    w1 = torch.randn(64, 32, 3, 3, generator=g) * (2.0 / (32 * 9)) ** 0.5
    s1, t1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    w2 = torch.randn(48, 64, 1, 1, generator=g) * 0.2
    b2 = torch.randn(48, generator=g)
    hid = bf16r(_halo_ref(x, w1, s1, t1, True))
    ref = F.conv2d(hid, bf16r(w2)) + b2.view(1, -1, 1, 1)
    pc = packing.pack_conv_halo("h", w1, s1, t1, relu=True, chain=(w2, torch.ones(48), b2, False),
                                epilogue=ops.V2X_EPI_F32, device=device)
    cls, loc = ops.conv2d(pc, to_nhwc_bf16(x, device), split=12)
    assert cls.shape == (N, H, W, 12) and loc.shape == (N, H, W, 36) and cls.dtype == torch.float32
    got = torch.cat([from_nhwc(cls), from_nhwc(loc)], 1)
    assert torch.allclose(got, ref, atol=2e-2, rtol=1e-2), float((got - ref).abs().max())
    assert float((got - ref).abs().mean()) < 1e-3


# ------------------------------------------------------------------------------------- a1, early fusion (upperbound)
@pytest.mark.parametrize("lds", [0, 2])
def test_voxelize_early_fusion_bit_exact(device, tune, lds):
    """BASELINE.json config 1: every ego grid = union of all agents' sweeps moved into the ego frame.  Bit-exact vs the
    oracle's fp32 transform (separately rounded ops) + voxelize_occupy; also checks jobs sharing a target grid.  Both forms: the
    global-atomic scatter (VOXELIZE_LDS = 0) and the LDS-binned one workgroup per target grid (= 2; the default takes it from 48 grids on)."""
    from v2x_sim_amd import ops
    tune("VOXELIZE_LDS", lds)
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    A, n = 5, 30000
    clouds = [VR.synthetic_points(n, seed=70 + a, n_edge=32) for a in range(A)]
    T = synthetic_poses(1, A, seed=5)[0]                      # T[i, j]: agent j's frame -> agent i's frame
    pts = torch.from_numpy(np.stack(clouds)).to(device)
    cnt = torch.full((A,), n, dtype=torch.int32, device=device)
    xf, src, dst = [], [], []
    for i in range(A):
        for j in range(A):
            xf.append(T[i, j][:3])
            src.append(j)
            dst.append(i)
    grid = ops.VoxelGrid()
    bits = ops.voxelize_fused_bits(pts, cnt, torch.tensor(np.stack(xf), dtype=torch.float32, device=device),
                                   torch.tensor(src, dtype=torch.int32, device=device),
                                   torch.tensor(dst, dtype=torch.int32, device=device), A, grid)
    dense = ops.bits_to_dense(bits, 13).cpu().numpy()
    for i in range(A):
        ref = VR.voxelize_early_fusion(clouds, [T[i, j] for j in range(A)])
        assert np.array_equal(dense[i], ref), "ego %d" % i
    # identity transform on a single job == plain voxelizer
    eye = torch.eye(4)[:3].unsqueeze(0).contiguous().to(device)
    one = ops.voxelize_fused_bits(pts, cnt, eye, torch.tensor([2], dtype=torch.int32, device=device),
                                  torch.tensor([0], dtype=torch.int32, device=device), 1, grid)
    assert torch.equal(one[0], ops.voxelize_bits(pts, cnt, grid)[2])
    # jobs in ANY order, a grid nobody writes (all zeros), a job naming a grid that does not exist (skipped), a pt_stride of 3
    order = torch.randperm(A * A, generator=torch.Generator().manual_seed(1))
    xf_t = torch.tensor(np.stack(xf), dtype=torch.float32)[order].contiguous().to(device)
    src_t = torch.tensor(src, dtype=torch.int32)[order].contiguous().to(device)
    dst_t = torch.tensor(dst, dtype=torch.int32)[order].contiguous()
    dst_t[dst_t == 3] = 77                                   # ego 3's jobs point nowhere
    shuffled = ops.voxelize_fused_bits(pts, cnt, xf_t, src_t, dst_t.to(device), A, grid)
    for i in range(A):
        assert torch.equal(shuffled[i], bits[i]) if i != 3 else int(shuffled[i].abs().sum()) == 0, i
    pts3 = pts[..., :3].contiguous()
    assert torch.equal(ops.voxelize_fused_bits(pts3, cnt, torch.tensor(np.stack(xf), dtype=torch.float32, device=device), torch.tensor(src, dtype=torch.int32, device=device),
                                               torch.tensor(dst, dtype=torch.int32, device=device), A, grid), bits)


def test_pixel_weighted_fuse_vs_torch(device):
    """f-4 / DiscoNet tail: per-pixel exp / normalise over the valid sources + weighted sum, vs torch fp32 (one bf16 ulp)."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(77)
    n, A, H, W, C = 3, 5, 8, 16, 64
    maps = bf16r(torch.randn(n, A, H, W, C, generator=g))
    scores = torch.rand(n, A, H, W, 4, generator=g) * 3.0
    valid = torch.ones(n, A)
    valid[1, 4] = 0
    valid[2, 2:] = 0
    e = torch.exp(scores[..., 0]) * valid.view(n, A, 1, 1)
    w = e / e.sum(1, keepdim=True)
    ref = (w.unsqueeze(-1) * maps).sum(1)
    out = ops.pixel_weighted_fuse(scores.to(device), valid.to(device), maps.to(torch.bfloat16).to(device))
    assert torch.allclose(out.float().cpu(), bf16r(ref), atol=2e-3, rtol=2 ** -7)


def test_voxelize_counts_beyond_capacity_and_bad_jobs(device, tune):
    """n_pts[i] > max_pts must be clamped in EVERY form (the scatter kernels used to run into the next cloud),
    and an early-fusion job naming a cloud / grid that does not exist is skipped, not executed out of bounds."""
    from v2x_sim_amd import ops
    grid = ops.VoxelGrid()
    clouds = [VR.synthetic_points(4096, seed=300 + i, n_edge=16) for i in range(3)]
    pts = torch.from_numpy(np.stack(clouds)).to(device)
    over = torch.tensor([4096 + 5000, 4096, 1 << 30], dtype=torch.int32, device=device)   # counts beyond the capacity
    exact = torch.full((3,), 4096, dtype=torch.int32, device=device)
    ref = ops.voxelize_bits(pts, exact, grid).clone()
    for form in ("2", "0"):
        tune("VOXELIZE_LDS", form)
        assert torch.equal(ops.voxelize_bits(pts, over, grid), ref), "form %s read past the cloud" % form
    tune.reset("VOXELIZE_LDS")
    eye = torch.eye(4)[:3].unsqueeze(0).repeat(4, 1, 1).contiguous().to(device)
    src = torch.tensor([0, 7, 1, -1], dtype=torch.int32, device=device)       # jobs 1 and 3 name clouds that do not exist
    dst = torch.tensor([0, 0, 9, 1], dtype=torch.int32, device=device)        # job 2 names a grid that does not exist
    got = ops.voxelize_fused_bits(pts, over, eye, src, dst, 2, grid)
    assert torch.equal(got[0], ref[0]) and int(got[1].abs().sum()) == 0
    with pytest.raises(ValueError, match="one count per cloud"):
        ops.voxelize_bits(pts, exact[:2], grid)


def test_halo_xcd_walk_is_a_pure_reordering(device, tune):
    """HALO_XCD permutes which workgroup of the halo kernels' persistent grids computes which tile (an XCD owns a contiguous eighth of the tiles):
    identical bits for the single-buffer, ping-pong, chained and bit-input forms, with tile counts that are not multiples of 8 or of the grid."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(5)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    cases = [(0, 32, 32, ncu // 8 * 3 + 5, 24, 64), (64, 32, 32, 7, 32, 32), (0, 64, 64, ncu // 4 + 3, 16, 32)]
    for cup, c, cout, N, H, W in cases:
        w = torch.randn(cout, cup + c, 3, 3, generator=g) * 0.05
        pc = packing.pack_conv_halo("h", w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=cup if cup else c,
                                    C1=c if cup else 0, relu=True, device=device)
        x = torch.randn(N, H, W, c, generator=g).to(torch.bfloat16).to(device)
        xu = torch.randn(N, H // 2, W // 2, cup, generator=g).to(torch.bfloat16).to(device) if cup else None
        run = (lambda: ops.conv2d(pc, xu, x)) if cup else (lambda: ops.conv2d(pc, x))
        tune("HALO_XCD", 0)
        ref = run().clone()
        tune("HALO_XCD", 1)
        assert torch.equal(run(), ref), (cup, c, cout, N)
