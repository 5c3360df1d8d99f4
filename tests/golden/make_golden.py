"""Generates the committed golden vectors under tests/golden/ from the build-owned CPU oracle.

PARITY UNPINNED: the reference tree (/root/reference) holds no code, tests or fixtures, so
these vectors pin the ORACLE (and through it the HIP kernels), not the reference.  Run from
the repo root:   python tests/golden/make_golden.py
Every fixture stores its inputs and the oracle outputs; sizes are kept to a few hundred KiB.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))

from oracle import coperception_ref as R  # noqa: E402
from oracle import voxelize_ref as VR  # noqa: E402


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def g_voxel():
    pts = VR.synthetic_points(2048, seed=7)
    grid, idx = VR.voxelize_occupy(pts, return_indices=True)
    assert np.array_equal(grid, VR.voxelize_direct(pts))
    np.savez_compressed(os.path.join(HERE, "voxel_2048.npz"), points=pts, indices=idx.astype(np.int32),
                        dims=np.asarray(grid.shape, dtype=np.int32))


def g_conv():
    g = torch.Generator().manual_seed(11)
    x = bf16(torch.randn(2, 16, 12, 20, generator=g))
    w = bf16(torch.randn(24, 16, 3, 3, generator=g) * 0.1)
    scale = torch.rand(24, generator=g) + 0.5
    shift = torch.randn(24, generator=g) * 0.1
    out = {}
    for stride in (1, 2):
        y = F.conv2d(x, w, None, stride, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        out["y_s%d" % stride] = F.relu(y).numpy()
    # two-source: nearest-upsampled x_up (2,8,6,10) concatenated in front of x
    x_up = bf16(torch.randn(2, 8, 6, 10, generator=g))
    w2 = bf16(torch.randn(24, 24, 3, 3, generator=g) * 0.1)
    cat = torch.cat((F.interpolate(x_up, scale_factor=(2, 2)), x), dim=1)
    out["y_upcat"] = F.relu(F.conv2d(cat, w2, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).numpy()
    np.savez_compressed(os.path.join(HERE, "conv_small.npz"), x=x.numpy(), w=w.numpy(), scale=scale.numpy(),
                        shift=shift.numpy(), x_up=x_up.numpy(), w2=w2.numpy(), **out)


def g_warp():
    g = torch.Generator().manual_seed(13)
    feat = bf16(torch.randn(2, 8, 32, 32, generator=g))
    import math
    yaw = 0.7
    T = torch.eye(4)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
    T[0, 3], T[1, 3] = 5.3, -3.1
    warped = R.feature_transformation(feat[1], T, (1, 8, 32, 32))
    np.savez_compressed(os.path.join(HERE, "warp_2agent.npz"), feat=feat.numpy(), T=T.numpy(), warped=warped.numpy())


def g_gru():
    torch.manual_seed(17)
    cell = R.Conv2dGRUCell(64, 32, 3)
    g = torch.Generator().manual_seed(17)
    x = bf16(torch.randn(1, 64, 8, 8, generator=g))
    with torch.no_grad():
        for p in cell.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        h = cell(x, None, emulate=True)
        h_fp32 = cell(x, None, emulate=False)
    np.savez_compressed(os.path.join(HERE, "gru_32.npz"), x=x.numpy(), w_ih=cell.weight_ih_l0.detach().numpy(),
                        w_hh=cell.weight_hh_l0.detach().numpy(), b_ih=cell.bias_ih_l0.detach().numpy(),
                        b_hh=cell.bias_hh_l0.detach().numpy(), h=h.numpy(), h_fp32=h_fp32.numpy())


def g_attn():
    g = torch.Generator().manual_seed(19)
    A, B = 5, 2
    keys = torch.randn(A * B, 1024, generator=g) * 0.3
    querys = torch.randn(A * B, 32, generator=g)
    attn = R.MIMOGeneralDotProductAttention(32, 1024)
    with torch.no_grad():
        attn.linear.weight.copy_(torch.randn(1024, 32, generator=g) * 0.05)
        attn.linear.bias.copy_(torch.randn(1024, generator=g) * 0.05)
        key_mat = torch.stack([keys[B * i: B * (i + 1)] for i in range(A)], 1)
        query_mat = torch.stack([querys[B * i: B * (i + 1)] for i in range(A)], 1)
        prob = attn.scores(query_mat, key_mat)
    m = R.When2com.__new__(R.When2com)
    m.agent_num = A
    act, _ = R.When2com.coefficients(m, prob, "activated", False)
    arg, _ = R.When2com.coefficients(m, prob, "argmax_test", False)
    np.savez_compressed(os.path.join(HERE, "attn_5x5.npz"), keys=keys.numpy(), querys=querys.numpy(),
                        w=attn.linear.weight.detach().numpy(), b=attn.linear.bias.detach().numpy(),
                        prob=prob.numpy(), coef_activated=act.numpy(), coef_argmax=arg.numpy())


def g_v2vnet_small():
    """End-to-end V2VNet on a 64x64x13 grid, 3 agents, 1 frame (weights from the seeded synthetic init)."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_poses
    A = 3
    pm = init_synthetic_weights(V2VNet(Config("train"), num_agent=A), seed=3)
    om = R.V2VNet(num_agent=A).eval()
    om.load_state_dict(pm.state_dict())
    rng = np.random.default_rng(23)
    bev = (rng.uniform(size=(A, 1, 64, 64, 13)) < 0.05).astype(np.float32)
    T = synthetic_poses(1, A, seed=29)
    T[..., :2, 3] *= 0.25  # keep the overlap on the small grid
    nat = torch.full((1, A), A)
    wsum = float(sum(p.double().abs().sum() for p in pm.state_dict().values()))
    out = {}
    with torch.no_grad():
        for tag, emu in (("fp32", False), ("emu", True)):
            om.emulate_bf16 = emu
            r = om(torch.from_numpy(bev), torch.from_numpy(T), nat, batch_size=1)
            out["cls_" + tag] = r["cls"].view(A, 64, 64, 12)[:, ::4, ::4].contiguous().numpy()
            out["loc_" + tag] = r["loc"].reshape(A, 64, 64, 36)[:, ::4, ::4].contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "v2vnet_small.npz"), bev=np.packbits(bev.astype(bool)), bev_shape=bev.shape,
                        T=T, weight_abs_sum=wsum, **out)


def g_postprocess():
    """Detection post-processing (build-owned spec): logits of a 32x32x6 anchor map with a few confident
    blobs -> boxes / scores / kept anchor indices of the ORACLE (oracle/postprocess_ref.py, scalar float64 -- not the
    product's utils/postprocess.py, which is checked against this fixture like the HIP kernel is)."""
    from oracle import postprocess_ref as PR
    rng = np.random.default_rng(31)
    X = Y = 32
    A = 6
    anc = np.zeros((X, Y, A, 6), np.float32)
    anc[..., 0] = (-4.0 + (np.arange(X) + 0.5) * 0.25)[:, None, None]
    anc[..., 1] = (-4.0 + (np.arange(Y) + 0.5) * 0.25)[None, :, None]
    for a, (w, h, yaw) in enumerate(R.ANCHOR_SIZE):
        anc[:, :, a, 2], anc[:, :, a, 3] = w * 0.25, h * 0.25
        anc[:, :, a, 4], anc[:, :, a, 5] = np.sin(yaw), np.cos(yaw)
    cls = np.zeros((X, Y, A, 2), np.float32)
    cls[..., 0] = 2.0 + rng.normal(0, 0.3, (X, Y, A))
    cls[..., 1] = -2.0 + rng.normal(0, 0.3, (X, Y, A))
    loc = rng.normal(0, 0.05, (X, Y, A, 1, 6)).astype(np.float32)
    loc[..., 5] += 1.0
    for _ in range(12):
        x, y, a = rng.integers(3, X - 3), rng.integers(3, Y - 3), rng.integers(0, A)
        for dx in range(-1, 2):
            for dy in range(-1, 2):
                s = 4.0 - 1.2 * (abs(dx) + abs(dy)) + rng.normal(0, 0.05)
                cls[x + dx, y + dy, a] = (-s, s)
    cls = cls.astype(np.float32).reshape(-1, 2)
    det = PR.detect(cls, loc.reshape(-1, 6), anc.reshape(-1, 6), 0.7, 0.01)
    np.savez_compressed(os.path.join(HERE, "postprocess_small.npz"), cls=cls, loc=loc, anchors=anc,
                        boxes=np.asarray([d["box"] for d in det], np.float64), scores=np.asarray([d["score"] for d in det], np.float64),
                        corners=np.asarray([d["corners"] for d in det], np.float64), index=np.asarray([d["index"] for d in det], np.int32))


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for fn in (g_voxel, g_conv, g_warp, g_gru, g_attn, g_v2vnet_small, g_postprocess):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
