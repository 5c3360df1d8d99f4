"""CPU tests of the ORACLE (not gpu-marked): restatements agree with each other and with the
committed golden vectors.  PARITY UNPINNED -- /root/reference has no tests/fixtures to pin
the oracle itself against; these tests pin it against drift."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import coperception_ref as R
from oracle import voxelize_ref as VR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def c_voxelize(lib, pts, voxel=VR.VOXEL_SIZE, extents=VR.AREA_EXTENTS):
    dims = VR.grid_dims(voxel, extents)
    occ = np.zeros(tuple(dims), dtype=np.uint8)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    ext = np.ascontiguousarray(extents, dtype=np.float64).reshape(-1)
    vs = np.asarray(voxel, dtype=np.float64)
    d = np.asarray(dims, dtype=np.int32)
    lib.oracle_voxelize_occupy(pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(pts.shape[0]),
                               ctypes.c_int(pts.shape[1]), ext.ctypes.data_as(ctypes.c_void_p),
                               vs.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
                               occ.ctypes.data_as(ctypes.c_void_p))
    idx = np.zeros((int(occ.sum()), 3), dtype=np.int32)
    m = lib.oracle_occupancy_indices(occ.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
                                     idx.ctypes.data_as(ctypes.c_void_p))
    assert m == idx.shape[0]
    return occ, idx


def test_grid_dims():
    assert tuple(VR.grid_dims()) == (256, 256, 13)
    cross = np.array([[-32.0, 32.0], [-32.0, 32.0], [-8.0, -3.0]])
    assert tuple(VR.grid_dims(VR.VOXEL_SIZE, cross)) == (256, 256, 13)


@pytest.mark.parametrize("n,seed", [(65536, 0), (4096, 1), (300, 2)])
def test_voxel_restatements_agree(oracle_c_lib, n, seed):
    pts = VR.synthetic_points(n, seed, n_edge=min(64, n // 4))
    grid, idx = VR.voxelize_occupy(pts, return_indices=True)
    assert np.array_equal(grid, VR.voxelize_direct(pts))
    occ, cidx = c_voxelize(oracle_c_lib, pts)
    assert np.array_equal(occ.astype(np.float32), grid)
    assert np.array_equal(cidx, idx.astype(np.int32))  # lexicographic order identical
    assert np.array_equal(VR.densify(idx, grid.shape), grid.astype(bool))


def test_voxel_golden(oracle_c_lib):
    g = np.load(os.path.join(GOLD, "voxel_2048.npz"))
    _, idx = VR.voxelize_occupy(g["points"], return_indices=True)
    assert np.array_equal(idx.astype(np.int32), g["indices"])
    _, cidx = c_voxelize(oracle_c_lib, g["points"])
    assert np.array_equal(cidx, g["indices"])


def test_voxel_edge_cases(oracle_c_lib):
    # empty cloud, everything out of range
    for pts in (np.zeros((0, 4), np.float32), np.full((10, 4), 100.0, np.float32)):
        grid = VR.voxelize_occupy(pts)
        assert grid.sum() == 0
        occ, idx = c_voxelize(oracle_c_lib, pts)
        assert occ.sum() == 0 and idx.shape[0] == 0
    # strict inequalities: points exactly on the extents are dropped
    on = np.array([[-32, 0, 0, 0], [32, 0, 0, 0], [0, -32, 0, 0], [0, 32, 0, 0], [0, 0, -3, 0], [0, 0, 2, 0]], np.float32)
    assert VR.voxelize_occupy(on).sum() == 0
    # a point on a voxel boundary belongs to the upper voxel (floor); fp64 division for z
    p = np.array([[0.25, -0.25, np.float32(0.4), 0]], np.float32)
    _, idx = VR.voxelize_occupy(p, return_indices=True)
    # float32(0.4) = 0.4000000059 > 0.4 -> floor(z/0.4) = 1 -> 1 - (-8) = 9
    assert idx.tolist() == [[129, 127, 9]]
    p = np.array([[0, 0, np.float32(1.2), 0]], np.float32)  # float32(1.2)/0.4 = 3.0000001 -> 3
    _, idx = VR.voxelize_occupy(p, return_indices=True)
    assert idx.tolist() == [[128, 128, 11]]
    # duplicates collapse
    d = np.repeat(np.array([[1.1, 2.2, 0.3, 0]], np.float32), 50, 0)
    assert VR.voxelize_occupy(d).sum() == 1


def test_golden_conv():
    g = np.load(os.path.join(GOLD, "conv_small.npz"))
    x, w = torch.from_numpy(g["x"]), torch.from_numpy(g["w"])
    sc, sf = torch.from_numpy(g["scale"]).view(1, -1, 1, 1), torch.from_numpy(g["shift"]).view(1, -1, 1, 1)
    for s in (1, 2):
        y = F.relu(F.conv2d(x, w, None, s, 1) * sc + sf)
        assert torch.allclose(y, torch.from_numpy(g["y_s%d" % s]), atol=1e-5)


def test_golden_warp_and_identity():
    g = np.load(os.path.join(GOLD, "warp_2agent.npz"))
    feat, T = torch.from_numpy(g["feat"]), torch.from_numpy(g["T"])
    w = R.feature_transformation(feat[1], T, (1, 8, 32, 32))
    assert torch.allclose(w, torch.from_numpy(g["warped"]), atol=1e-6)
    ident = R.feature_transformation(feat[0], torch.eye(4), (1, 8, 32, 32))
    assert torch.allclose(ident, feat[0], atol=1e-5)


def test_golden_gru_and_h0_identity():
    g = np.load(os.path.join(GOLD, "gru_32.npz"))
    cell = R.Conv2dGRUCell(64, 32, 3)
    with torch.no_grad():
        cell.weight_ih_l0.copy_(torch.from_numpy(g["w_ih"]))
        cell.weight_hh_l0.copy_(torch.from_numpy(g["w_hh"]))
        cell.bias_ih_l0.copy_(torch.from_numpy(g["b_ih"]))
        cell.bias_hh_l0.copy_(torch.from_numpy(g["b_hh"]))
        x = torch.from_numpy(g["x"])
        assert torch.allclose(cell(x, None, emulate=True), torch.from_numpy(g["h"]), atol=1e-6)
        h = cell(x, None)
        assert torch.allclose(h, torch.from_numpy(g["h_fp32"]), atol=1e-6)
        # h0 = 0  =>  h = (1 - z) * n with gh = b_hh: W_hh never matters
        gi = F.conv2d(x, cell.weight_ih_l0, cell.bias_ih_l0, 1, 1)
        i_r, i_z, i_n = gi.chunk(3, 1)
        b_r, b_z, b_n = cell.bias_hh_l0.view(1, -1, 1, 1).chunk(3, 1)
        r, z = torch.sigmoid(i_r + b_r), torch.sigmoid(i_z + b_z)
        n = torch.tanh(i_n + r * b_n)
        assert torch.allclose(h, (1 - z) * n, atol=1e-6)


def test_golden_attention():
    g = np.load(os.path.join(GOLD, "attn_5x5.npz"))
    A, B = 5, 2
    attn = R.MIMOGeneralDotProductAttention(32, 1024)
    with torch.no_grad():
        attn.linear.weight.copy_(torch.from_numpy(g["w"]))
        attn.linear.bias.copy_(torch.from_numpy(g["b"]))
        keys, querys = torch.from_numpy(g["keys"]), torch.from_numpy(g["querys"])
        key_mat = torch.stack([keys[B * i: B * (i + 1)] for i in range(A)], 1)
        query_mat = torch.stack([querys[B * i: B * (i + 1)] for i in range(A)], 1)
        prob = attn.scores(query_mat, key_mat)
    assert torch.allclose(prob, torch.from_numpy(g["prob"]), atol=1e-6)
    assert torch.allclose(prob.sum(1), torch.ones(B, A), atol=1e-5)  # softmax over keys
    arg = torch.from_numpy(g["coef_argmax"])
    assert torch.equal(arg.sum(1), torch.ones(B, A))


def _small_v2vnet(A=3):
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    pm = init_synthetic_weights(V2VNet(Config("train"), num_agent=A), seed=3)
    om = R.V2VNet(num_agent=A).eval()
    om.load_state_dict(pm.state_dict())
    return pm, om


def test_golden_v2vnet_small():
    g = np.load(os.path.join(GOLD, "v2vnet_small.npz"))
    A = 3
    pm, om = _small_v2vnet(A)
    wsum = float(sum(p.double().abs().sum() for p in pm.state_dict().values()))
    if not math.isclose(wsum, float(g["weight_abs_sum"]), rel_tol=1e-9):
        pytest.skip("torch RNG stream differs from the one the fixture was generated with")
    shape = tuple(int(v) for v in g["bev_shape"])
    bev = np.unpackbits(g["bev"])[: int(np.prod(shape))].reshape(shape).astype(np.float32)
    nat = torch.full((1, A), A)
    with torch.no_grad():
        for tag, emu in (("fp32", False), ("emu", True)):
            om.emulate_bf16 = emu
            r = om(torch.from_numpy(bev), torch.from_numpy(g["T"]), nat, batch_size=1)
            cls = r["cls"].view(A, 64, 64, 12)[:, ::4, ::4]
            loc = r["loc"].reshape(A, 64, 64, 36)[:, ::4, ::4]
            assert torch.allclose(cls, torch.from_numpy(g["cls_" + tag]), atol=2e-4), tag
            assert torch.allclose(loc, torch.from_numpy(g["loc_" + tag]), atol=2e-4), tag


def test_v2vnet_iterations_and_sources():
    """gnn_iter_times=2: 'initial' and 'updated' neighbour sources differ; padding agents untouched."""
    A = 3
    pm, om = _small_v2vnet(A)
    rng = np.random.default_rng(5)
    bev = torch.from_numpy((rng.uniform(size=(A, 1, 64, 64, 13)) < 0.05).astype(np.float32))
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    T = synthetic_poses(1, A, seed=1)
    T[..., :2, 3] *= 0.25
    T = torch.from_numpy(T)
    with torch.no_grad():
        enc = om.u_encoder(bev.permute(0, 1, 4, 2, 3))
        lcm = om.local_com_mat(enc[3], 1)
        om.gnn_iter_num = 2
        u_init = om.fuse(lcm, T, torch.full((1, A), A), 1)
        om.neighbor_source = "updated"
        u_upd = om.fuse(lcm, T, torch.full((1, A), A), 1)
        assert not torch.allclose(u_init, u_upd)
        # third reading (ASSUMPTIONS.md row 25): the ego map also from the encoder maps -> every round recomputes round 1
        om.neighbor_source = "frozen"
        u_frz = om.fuse(lcm, T, torch.full((1, A), A), 1)
        om.gnn_iter_num, om.neighbor_source = 1, "initial"
        u_one = om.fuse(lcm, T, torch.full((1, A), A), 1)
        assert torch.equal(u_frz, u_one) and not torch.allclose(u_frz, u_init)
        from v2x_sim_amd.configs import Config
        assert type(pm)(Config("train"), gnn_iter_times=3, num_agent=A, neighbor_source="frozen").gnn_rounds() == 1
        # 2 real agents + 1 padding agent: the padding agent keeps its encoder features
        om.gnn_iter_num, om.neighbor_source = 1, "initial"
        u2 = om.fuse(lcm, T, torch.tensor([[2, 2, 2]]), 1)
        assert torch.equal(u2[0, 2], lcm[0, 2]) and not torch.equal(u2[0, 0], lcm[0, 0])


def test_confusion_matrix():
    pred = torch.tensor([0, 1, 1, 7, 3])
    label = torch.tensor([0, 1, 2, 7, 3])
    cm = R.confusion_matrix(pred, label)
    assert cm.sum() == 5 and cm[2, 1] == 1 and cm[7, 7] == 1


def test_early_fusion_transform_spec():
    """fp32 transform with separately rounded ops: identity is exact, a generic pose agrees with the
    fp64 product to fp32 round-off, and the merged grid is the union of the individually voxelized moved clouds."""
    pts = VR.synthetic_points(5000, seed=11)
    eye = np.eye(4, dtype=np.float32)
    assert np.array_equal(VR.transform_points_f32(pts, eye)[:, :3], pts[:, :3])
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    T = synthetic_poses(1, 3, seed=2)[0]
    moved = VR.transform_points_f32(pts, T[0, 1])
    ref = pts[:, :3].astype(np.float64) @ T[0, 1][:3, :3].astype(np.float64).T + T[0, 1][:3, 3].astype(np.float64)
    assert np.abs(moved[:, :3] - ref).max() < 2e-5
    assert np.array_equal(moved[:, 3], pts[:, 3])
    clouds = [VR.synthetic_points(3000, seed=20 + a) for a in range(3)]
    merged = VR.voxelize_early_fusion(clouds, [T[0, j] for j in range(3)])
    union = np.zeros_like(merged)
    for j in range(3):
        union = np.maximum(union, VR.voxelize_occupy(VR.transform_points_f32(clouds[j], T[0, j])))
    assert np.array_equal(merged, union)


def test_warp_pose_convention_of_synthetic_scenes():
    """The synthetic scenes' `trans` (utils/synthetic_scene.make_scene) is in the axis convention the upstream warp formula
    presumes: a feature blob at a world point in agent j's map lands on that world point's cell in agent i's map."""
    import math
    from v2x_sim_amd.utils import synthetic_scene as S
    sc = S.make_scene(agents=4, n_cars=6, seed=5)
    rng = np.random.default_rng(1)
    hits = plain = n = 0
    for i in range(4):
        for j in range(4):
            if i == j:
                continue
            G = sc["trans_geo"][i, j].astype(np.float64)          # p_i = G p_j
            for _ in range(5):
                pj = np.array([*rng.uniform(-20, 20, 2), 0.0, 1.0])
                pi = G @ pj
                ci = (int(math.floor(pi[0] / 2)) + 16, int(math.floor(pi[1] / 2)) + 16)   # layer-3 cell (2 m), dim0 = x
                cj = (int(math.floor(pj[0] / 2)) + 16, int(math.floor(pj[1] / 2)) + 16)
                if not all(2 <= c < 30 for c in ci + cj):
                    continue
                n += 1
                F = torch.zeros(1, 32, 32)
                F[0, cj[0], cj[1]] = 1.0
                for T, which in ((sc["trans"][i, j], "conv"), (sc["trans_geo"][i, j], "plain")):
                    out = R.feature_transformation(F, torch.from_numpy(T), (1, 1, 32, 32))[0]
                    pk = np.unravel_index(int(out.argmax()), out.shape)
                    ok = float(out.max()) > 0 and max(abs(pk[0] - ci[0]), abs(pk[1] - ci[1])) <= 1
                    if which == "conv":
                        hits += ok
                    else:
                        plain += ok
    assert n >= 20 and hits == n, (n, hits)
    assert plain < n // 2          # the geometric matrix itself is NOT what the formula expects


@pytest.mark.parametrize("with_state", [False, True])
def test_convgru_cell_is_torch_grucell_per_pixel(with_state):
    """Pins oracle/ASSUMPTIONS.md row 22 (gate order r, z, n; h' = n + z (h - n); b_hn inside the reset product) to torch's own code:
    `convolutional_rnn.Conv2dGRUCell` is torch's GRUCell with the two linear maps replaced by convolutions, so with a 1x1 kernel every
    pixel of the oracle's cell must BE torch.nn.GRUCell on that pixel's channel vector with the same parameters -- and with a 3x3 kernel
    whose off-centre taps are zero as well."""
    torch.manual_seed(11)
    cin, hid = 24, 16
    ref = torch.nn.GRUCell(cin, hid)
    x = torch.randn(3, cin, 5, 7)
    h0 = torch.randn(3, hid, 5, 7) if with_state else None
    px = x.permute(0, 2, 3, 1).reshape(-1, cin)
    ph = h0.permute(0, 2, 3, 1).reshape(-1, hid) if with_state else None
    want = ref(px, ph).reshape(3, 5, 7, hid).permute(0, 3, 1, 2)
    for k in (1, 3):
        cell = R.Conv2dGRUCell(cin, hid, k)
        with torch.no_grad():
            for name in ("weight_ih", "weight_hh"):
                w = torch.zeros_like(getattr(cell, name + "_l0"))
                w[:, :, k // 2, k // 2] = getattr(ref, name)
                getattr(cell, name + "_l0").copy_(w)
            cell.bias_ih_l0.copy_(ref.bias_ih)
            cell.bias_hh_l0.copy_(ref.bias_hh)
            got = cell(x, h0)
        assert torch.allclose(got, want, atol=1e-6, rtol=1e-6), float((got - want).abs().max())
