"""CPU tests of the parsed-dataset reader/writer: README.md:66-79 layout, densify,
agent-major collation."""
import os

import numpy as np
import pytest
import torch

from oracle import voxelize_ref as VR
from v2x_sim_amd.configs import Config
from v2x_sim_amd.datasets import V2XSimDet, collate_dense, write_sample
from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses


def make_tree(tmp, A=3, frames=2, n_pts=3000):
    pts = synthetic_points(A * frames, n_pts, seed=3)
    T = synthetic_poses(frames, A, seed=4)
    grids = {}
    for f in range(frames):
        for a in range(A):
            grid, idx = VR.voxelize_occupy(pts[a * frames + f], return_indices=True)
            grids[(a, f)] = grid
            # agent0 = RSU directory exists too (README.md:70); scene 7, frames 0..
            write_sample(str(tmp), "test", a, 7, f, idx[::-1], T[f, a], A)   # unsorted on purpose
    roots = [os.path.join(str(tmp), "test", "agent%d" % a) for a in range(A)]
    return roots, grids, T


def test_layout_and_roundtrip(tmp_path):
    roots, grids, T = make_tree(tmp_path)
    assert os.path.isfile(os.path.join(str(tmp_path), "test", "agent0", "7_1", "0.npy"))   # README.md:66-79 layout
    ds = V2XSimDet(dataset_roots=roots, config=Config("test"), split="test", val=True)
    assert len(ds) == 2 and ds.seq_names == ["7_0", "7_1"]
    s = ds[1]
    assert len(s) == 3 and len(s[0]) == 13 and s[0][12].shape == (0, 5)
    for a in range(3):
        bev = s[a][0]
        assert bev.shape == (1, 256, 256, 13) and bev.dtype == np.float32
        assert np.array_equal(bev[0], grids[(a, 1)])                      # densify == voxelize_occupy's grid
        assert s[a][9] == a and s[a][10] == 3 and np.allclose(s[a][11], T[1, a])
    bevs, trans, nat = collate_dense([ds[0], ds[1]])
    assert bevs.shape == (6, 1, 256, 256, 13) and trans.shape == (2, 3, 3, 4, 4) and nat.tolist() == [[3] * 3] * 2
    assert torch.equal(bevs[2 * 1 + 1, 0], torch.from_numpy(grids[(1, 1)]))   # row = agent*B + frame
    sp = V2XSimDet(dataset_roots=roots, config=Config("test"), split="test", densify="none")[0]
    idx = sp[0][0]
    assert idx.dtype == np.int32 and np.array_equal(VR.densify(idx, (256, 256, 13)), grids[(0, 0)].astype(bool))
    assert np.array_equal(idx, idx[np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))])  # stored sorted


def test_training_target_fields_of_the_tuple(tmp_path):
    """targets=True fills positions 2-5 of upstream's per-agent tuple (label_one_hot, reg_target, reg_loss_mask, anchors_map), dense, and
    they are the same targets the GPU loop scatters from the sparse form (train/loop.py::dataset_batch_on_device)."""
    from v2x_sim_amd.utils import postprocess, synthetic_scene
    cfg = Config("train")
    pts = synthetic_points(1, 2000, seed=5)
    _, idx = VR.voxelize_occupy(pts[0], return_indices=True)
    boxes = np.array([[4.0, -6.0, 2.0, 4.5, 0.3], [-12.0, 9.0, 1.9, 4.2, -1.2]], np.float32)
    write_sample(str(tmp_path), "train", 0, 1, 0, idx, np.eye(4, dtype=np.float32)[None], 1, gt_boxes=boxes)
    ds = V2XSimDet(dataset_roots=[os.path.join(str(tmp_path), "train", "agent0")], config=cfg, split="train", targets=True)
    t = ds[0][0]
    assert len(t) == 13
    label, reg, mask, anchors = t[2], t[3], t[4], t[5]
    X, Y, A = anchors.shape[:3]
    assert label.shape == (X, Y, A, 2) and reg.shape == (X, Y, A, 1, 6) and mask.shape == (X, Y, A, 1) and mask.dtype == bool
    assert np.array_equal(anchors, postprocess.build_anchor_map(cfg))
    pos, code = synthetic_scene.anchor_targets_sparse(boxes, anchors)
    assert pos.shape[0] > 0 and int(mask.sum()) == pos.shape[0] == int(label[..., 1].sum())
    assert np.array_equal(np.argwhere(mask[..., 0]), pos) and np.allclose(reg[mask[..., 0]][:, 0], code)
    assert np.all(label.sum(-1) == 1.0) and not reg[~mask[..., 0]].any()
    # without the flag the fields stay None (the GPU loops build the targets on the device)
    assert V2XSimDet(dataset_roots=[os.path.join(str(tmp_path), "train", "agent0")], config=cfg, split="train")[0][0][2] is None


def test_upstream_style_sample_is_tolerated(tmp_path):
    """oracle/ASSUMPTIONS.md row 7 (UNVERIFIED key names): a 0.npy written with the keys the author recalls of upstream's create_data_det.py --
    `num_agent` / `target_agent` spellings, `trans_matrices_no_cross_road`, the sparse training targets (`allocation_mask`, `label_sparse`,
    `reg_target_sparse`, `reg_loss_mask`, `gt_max_iou`) and extra keys -- loads instead of being refused; the dense targets are upstream's own
    assignment (not the build's rule), and a dict without the sweep or the poses is refused with the missing key named."""
    from v2x_sim_amd.datasets.V2XSimDet import normalize_sample, upstream_dense_targets
    cfg = Config("train")
    X, Y, Z = cfg.map_dims
    A = len(cfg.anchor_size)
    pts = synthetic_points(1, 2000, seed=6)
    grid, idx = VR.voxelize_occupy(pts[0], return_indices=True)
    alloc = np.zeros((X, Y, A), bool)
    alloc[10, 20, 1] = alloc[10, 21, 1] = alloc[200, 7, 4] = True
    rmask = np.zeros((X, Y, A, 1), bool)
    rmask[10, 20, 1, 0] = rmask[200, 7, 4, 0] = True            # one allocated anchor is background: in the label, not in the regression mask
    sparse = np.arange(3 * 6, dtype=np.float32).reshape(3, 1, 6) / 10.0
    sample = {"voxel_indices_0": idx.astype(np.int32), "trans_matrices": np.eye(4, dtype=np.float32)[None] * 2.0,
              "trans_matrices_no_cross_road": np.eye(4, dtype=np.float32)[None], "num_agent": 1, "target_agent": 0,
              "allocation_mask": alloc, "label_sparse": np.array([1, 0, 1]), "reg_target_sparse": sparse, "reg_loss_mask": rmask,
              "gt_max_iou": np.array([0.7, 0.1, 0.9], np.float32), "current_sample_token": "abc", "3d_dimension": np.zeros(3)}
    d = os.path.join(str(tmp_path), "train", "agent0", "3_0")
    os.makedirs(d)
    np.save(os.path.join(d, "0.npy"), sample, allow_pickle=True)
    t = V2XSimDet(dataset_roots=[os.path.join(str(tmp_path), "train", "agent0")], config=cfg, split="train", targets=True)[0][0]
    assert np.array_equal(t[0][0], grid) and t[9] == 0 and t[10] == 1
    assert np.allclose(t[11], np.eye(4)[None])                  # not a cross-road config: the *_no_cross_road poses
    label, reg, mask = t[2], t[3], t[4]
    assert label.shape == (X, Y, A, 2) and reg.shape == (X, Y, A, 1, 6) and mask.shape == (X, Y, A, 1)
    assert label[10, 20, 1].tolist() == [0.0, 1.0] and label[10, 21, 1].tolist() == [1.0, 0.0] and label[0, 0, 0].tolist() == [1.0, 0.0]
    assert np.allclose(reg[10, 20, 1, 0], sparse[0, 0]) and np.allclose(reg[200, 7, 4, 0], sparse[2, 0]) and not reg[10, 21, 1].any()
    assert int(label[..., 1].sum()) == 2 and np.array_equal(t[7], sample["gt_max_iou"])
    cross = normalize_sample(sample, is_cross_road=True)
    assert np.allclose(cross["trans_matrices"], 2.0 * np.eye(4)[None])
    assert upstream_dense_targets({"allocation_mask": alloc}) is None
    with pytest.raises(KeyError, match="voxel_indices_0"):
        normalize_sample({"trans_matrices": 0, "num_sensor": 1, "target_agent_id": 0})
    with pytest.raises(KeyError, match="trans_matrices"):
        normalize_sample({"voxel_indices_0": idx, "num_sensor": 1, "target_agent_id": 0})


def test_errors(tmp_path):
    with pytest.raises(ValueError):
        V2XSimDet(dataset_roots=None, config=Config("test"), split="test")
    with pytest.raises(FileNotFoundError):
        V2XSimDet(dataset_roots=[str(tmp_path / "nope")], config=Config("test"), split="test")
    os.makedirs(tmp_path / "empty")
    with pytest.raises(RuntimeError):
        V2XSimDet(dataset_roots=[str(tmp_path / "empty")], config=Config("test"), split="test")


def test_prefetcher_host_logic():
    """DevicePrefetcher (row f-2) host side: structure mapping, argument checks; the copies themselves need the GPU
    (tests/test_gpu_prefetch.py)."""
    import pytest
    import torch
    from v2x_sim_amd.datasets import DevicePrefetcher
    from v2x_sim_amd.datasets.prefetch import _map
    seen = []
    out = _map({"a": [1, (2, 3)], "b": "x"}, lambda v: seen.append(v) or (v, type(v).__name__))
    assert out == {"a": [(1, "int"), ((2, "int"), (3, "int"))], "b": ("x", "str")} and seen == [1, 2, 3, "x"]
    with pytest.raises(RuntimeError):
        DevicePrefetcher(iter([]), "cpu")
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            DevicePrefetcher(iter([]), "cuda:0", depth=0)


def test_seg_dataset_layout_and_roundtrip(tmp_path):
    """V2X-Sim-seg (README.md:66-79 lists it next to V2X-Sim-det): same tree, 0.npy additionally holds the class map."""
    from v2x_sim_amd.datasets import V2XSimSeg, write_seg_sample
    A, frames = 3, 2
    pts = synthetic_points(A * frames, 2000, seed=5)
    T = synthetic_poses(frames, A, seed=6)
    rng = np.random.default_rng(0)
    segs, grids = {}, {}
    for f in range(frames):
        for a in range(A):
            grid, idx = VR.voxelize_occupy(pts[a * frames + f], return_indices=True)
            seg = rng.integers(0, 8, (256, 256))
            segs[(a, f)], grids[(a, f)] = seg, grid
            write_seg_sample(str(tmp_path), "train", a, 11, f, idx, T[f, a], A, seg)
    roots = [os.path.join(str(tmp_path), "train", "agent%d" % a) for a in range(A)]
    assert os.path.isfile(os.path.join(roots[0], "11_1", "0.npy"))
    ds = V2XSimSeg(dataset_roots=roots, config=Config("train"), split="train")
    assert len(ds) == 2 and ds.seq_names == ["11_0", "11_1"]
    s = ds[1]
    for a in range(A):
        pvp, seg, name, tid, ns, trans = s[a]
        assert pvp.shape == (1, 256, 256, 13) and np.array_equal(pvp[0], grids[(a, 1)])
        assert seg.dtype == np.uint8 and np.array_equal(seg, segs[(a, 1)])
        assert name.endswith("11_1") and tid == a and ns == A and np.allclose(trans, T[1, a])
    sparse = V2XSimSeg(dataset_roots=roots, config=Config("train"), split="train", densify="none")[0][2][0]
    assert sparse.shape[1] == 3 and sparse.dtype == np.int32
    # a det sample is not a seg sample; an out-of-range or mis-shaped class map is refused at write time
    write_sample(str(tmp_path), "val", 0, 1, 0, np.zeros((1, 3), np.int32), T[0, 0], 1)
    with pytest.raises(KeyError, match="bev_seg"):
        V2XSimSeg(dataset_roots=[os.path.join(str(tmp_path), "val", "agent0")], config=Config("train"), split="val")[0]
    with pytest.raises(ValueError):
        write_seg_sample(str(tmp_path), "val", 0, 1, 1, np.zeros((1, 3), np.int32), T[0, 0], 1, np.full((256, 256), 300))
