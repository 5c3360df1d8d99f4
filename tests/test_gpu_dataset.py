"""GPU: the on-device densify (v2x_indices_to_bits) equals the voxelizer's own bits, and a frame read from the
parsed-dataset layout gives bit-identical logits through the dense (upstream-style) and the sparse->GPU paths."""
import os

import numpy as np
import pytest
import torch

from oracle import voxelize_ref as VR

pytestmark = pytest.mark.gpu


def test_indices_to_bits_roundtrip(device):
    from v2x_sim_amd import ops
    grid = ops.VoxelGrid()
    clouds = [VR.synthetic_points(n, seed=50 + i, n_edge=16) for i, n in enumerate((30000, 500, 64))]
    mp = max(c.shape[0] for c in clouds)
    buf = np.zeros((3, mp, 4), np.float32)
    for i, c in enumerate(clouds):
        buf[i, :c.shape[0]] = c
    cnt = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int32, device=device)
    bits = ops.voxelize_bits(torch.from_numpy(buf).to(device), cnt, grid)
    idx, counts = ops.bits_to_indices(bits, 13, 32768)
    again = ops.indices_to_bits(idx, counts, grid)
    assert torch.equal(bits, again)                                       # bit-exact round trip
    # out-of-range and duplicate indices: dropped / idempotent
    bad = torch.tensor([[[0, 0, 0], [0, 0, 0], [256, 0, 0], [-1, 5, 5], [3, 4, 13], [3, 4, 12]]], dtype=torch.int32, device=device)
    b = ops.indices_to_bits(bad, torch.tensor([6], dtype=torch.int32, device=device), grid)
    assert int(b[0, 0, 0]) == 1 and int(b[0, 3, 4]) == (1 << 12) and int((b != 0).sum()) == 2


def test_dataset_dense_and_sparse_paths_agree(device, tmp_path):
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.datasets import V2XSimDet, collate_dense, collate_to_device, write_sample
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
    A, frames = 3, 1
    pts = synthetic_points(A, 20000, seed=9)
    T = synthetic_poses(frames, A, seed=10)
    for a in range(A):
        _, idx = VR.voxelize_occupy(pts[a], return_indices=True)
        write_sample(str(tmp_path), "test", a, 0, 0, idx, T[0, a], A)
    roots = [os.path.join(str(tmp_path), "test", "agent%d" % a) for a in range(A)]
    cfg = Config("test")
    pm = init_synthetic_weights(V2VNet(cfg, num_agent=A), seed=1).to(device)
    dense = V2XSimDet(dataset_roots=roots, config=cfg, split="test")
    sparse = V2XSimDet(dataset_roots=roots, config=cfg, split="test", densify="none")
    bevs, trans, nat = collate_dense([dense[0]])
    x0, trans_d, nat2 = collate_to_device([sparse[0]], ops.VoxelGrid(), device)
    with torch.no_grad():
        a = pm(bevs.to(device), trans.to(device), nat, batch_size=1)
        b = pm.forward_nhwc(x0, trans_d, nat2, batch_size=1)
    assert torch.equal(a["cls"], b["cls"]) and torch.equal(a["loc"], b["loc"])
