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


@pytest.mark.parametrize("com", ["v2v", "lowerbound", "upperbound", "who2com"])
def test_test_codet_driver_runs_on_a_parsed_tree(device, tmp_path, com, capsys):
    """tools/det/test_codet.py (upstream flag surface) end to end on a synthetic parsed dataset + a saved checkpoint."""
    import importlib.util
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.datasets import write_sample
    from v2x_sim_amd.models.det import FaFNet, V2VNet, When2com
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
    A, frames = 3, 2
    pts = synthetic_points(A * frames, 15000, seed=31)
    T = synthetic_poses(frames, A, seed=32)
    rng = np.random.default_rng(1)
    for f in range(frames):
        for a in range(A):
            if com == "upperbound":
                # upstream's upperbound set stores the MERGED sweep of every agent (early fusion at data-creation time):
                # here the HIP early-fusion voxeliser writes it (3 jobs into one grid), read back as the sparse index list
                from v2x_sim_amd import ops
                grid = ops.VoxelGrid()
                cl = torch.from_numpy(np.stack([pts[j * frames + f] for j in range(A)])).to(device)
                bits = ops.voxelize_fused_bits(cl, torch.full((A,), cl.shape[1], dtype=torch.int32, device=device),
                                               torch.tensor(np.stack([T[f, a, j][:3] for j in range(A)]), dtype=torch.float32, device=device),
                                               torch.arange(A, dtype=torch.int32, device=device),
                                               torch.zeros(A, dtype=torch.int32, device=device), 1, grid)
                ix, ct = ops.bits_to_indices(bits, grid.dims[2], 65536)
                idx = ix[0, :int(ct[0])].cpu().numpy()
                assert idx.shape[0] > VR.voxelize_occupy(pts[a * frames + f]).sum()
            else:
                _, idx = VR.voxelize_occupy(pts[a * frames + f], return_indices=True)
            gt = np.concatenate([rng.uniform(-25, 25, (6, 2)), np.tile([2.0, 4.0], (6, 1)), rng.uniform(-1, 1, (6, 1))], 1)
            write_sample(str(tmp_path), "test", a, 3, f, idx, T[f, a], A, gt_boxes=gt)
    cls = {"v2v": V2VNet, "lowerbound": FaFNet, "upperbound": FaFNet, "who2com": When2com}[com]
    kw = {"kd_flag": 0} if com in ("lowerbound", "upperbound") else {}
    ckpt = os.path.join(str(tmp_path), "ckpt.pth")
    torch.save({"epoch": 1, "model_state_dict": init_synthetic_weights(cls(Config("test"), num_agent=A, **kw), seed=4).state_dict()}, ckpt)
    spec = importlib.util.spec_from_file_location("test_codet", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "det", "test_codet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(["--data", os.path.join(str(tmp_path), "test"), "--com", com, "--resume", ckpt, "--num_agent", str(A),
                    "--batch", "2", "--score_thr", "0.55"])
    out = capsys.readouterr().out
    assert "average local mAP@0.5" in out and out.count("agent") >= A
    assert 0.0 <= res[0.5] <= 1.0 and 0.0 <= res[0.7] <= 1.0


def test_train_then_test_drivers_end_to_end(device, tmp_path, capsys):
    """The drop-in pair of BASELINE.json's north_star: tools/det/train_codet.py trains (synthetic scenes, PyTorch-ROCm autograd
    over the engine's parameters) and saves upstream's checkpoint format; a parsed dataset in the README.md:66-79 layout is
    written from fresh scenes; tools/det/test_codet.py --resume evaluates it on the HIP inference path (device NMS, rotated
    IoU AP).  A short training already detects the synthetic cars: mAP@0.5 well above chance."""
    import importlib.util
    from v2x_sim_amd import ops
    from v2x_sim_amd.datasets import write_sample
    from v2x_sim_amd.utils import synthetic_scene
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "det")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    logdir = os.path.join(str(tmp_path), "log")
    load("train_codet").main(["--data", "synthetic", "--com", "v2v", "--steps", "250", "--batch", "2", "--logpath", logdir])
    ckpt = os.path.join(logdir, "epoch_1.pth")
    assert os.path.exists(ckpt) and "model_state_dict" in torch.load(ckpt, map_location="cpu")
    # parsed dataset: 4 frames x 5 agents, sparse voxel indices from the GPU voxeliser, poses in the dataset convention
    A, frames = 5, 4
    grid = ops.VoxelGrid()
    root = os.path.join(str(tmp_path), "V2X-Sim-det")
    for f in range(frames):
        sc = synthetic_scene.make_scene(A, seed=9000 + f)
        bits = ops.voxelize_bits(torch.from_numpy(sc["points"]).to(device), torch.from_numpy(sc["n_pts"]).to(device), grid)
        idx, cnt = ops.bits_to_indices(bits, grid.dims[2], 32768)
        idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
        for a in range(A):
            write_sample(root, "test", a, 7, f, idx[a, :cnt[a]], sc["trans"][a], A, gt_boxes=sc["gt_boxes"][a])
    res = load("test_codet").main(["--data", os.path.join(root, "test"), "--com", "v2v", "--resume", ckpt, "--batch", "2"])
    out = capsys.readouterr().out
    print(out[-400:])
    assert "average local mAP@0.5" in out
    assert res[0.5] > 0.3, res


def test_train_on_a_parsed_dataset_then_test(device, tmp_path, capsys):
    """f-3 on the dataset path: tools/det/train_codet.py --data <parsed train split> (sparse sweeps densified and anchor
    targets built from the stored boxes on the GPU) -> checkpoint -> tools/det/test_codet.py on a parsed test split."""
    import importlib.util
    from v2x_sim_amd import ops
    from v2x_sim_amd.datasets import write_sample
    from v2x_sim_amd.utils import synthetic_scene
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "det")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    A = 5
    grid = ops.VoxelGrid()
    root = os.path.join(str(tmp_path), "V2X-Sim-det")
    for split, frames, seed0 in (("train", 48, 20000), ("test", 4, 30000)):
        for f in range(frames):
            sc = synthetic_scene.make_scene(A, seed=seed0 + f)
            bits = ops.voxelize_bits(torch.from_numpy(sc["points"]).to(device), torch.from_numpy(sc["n_pts"]).to(device), grid)
            idx, cnt = ops.bits_to_indices(bits, grid.dims[2], 32768)
            idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
            for a in range(A):
                write_sample(root, split, a, 3, f, idx[a, :cnt[a]], sc["trans"][a], A, gt_boxes=sc["gt_boxes"][a])
    logdir = os.path.join(str(tmp_path), "log")
    load("train_codet").main(["--data", os.path.join(root, "train"), "--com", "v2v", "--nepoch", "10", "--batch", "2",
                              "--logpath", logdir, "--log"])
    out = capsys.readouterr().out
    losses = [float(ln.split("steps")[1]) for ln in out.splitlines() if ln.startswith("epoch") and "mean loss" in ln]
    print("epoch losses", losses)
    assert len(losses) == 10 and losses[-1] < 0.5 * losses[0]
    ckpt = os.path.join(logdir, "epoch_10.pth")
    res = load("test_codet").main(["--data", os.path.join(root, "test"), "--com", "v2v", "--resume", ckpt, "--batch", "2"])
    out = capsys.readouterr().out
    print(out[-300:])
    assert res[0.5] > 0.3, res


def test_train_driver_keeps_optimizer_and_schedule_across_epochs_and_resume(device, tmp_path, capsys):
    """one Adam + one MultiStepLR for the whole run (not per epoch), both in the checkpoint, --resume restores
    them and continues the epoch numbering.  3 epochs x 10 steps: milestones at steps 18 and 25 -> lr 1e-3, 3e-4, 9e-5 at
    the ends of epochs 1, 2, 3; a run resumed from epoch_1.pth ends with the same schedule state and never rewrites epoch_1."""
    import importlib.util
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "det")
    spec = importlib.util.spec_from_file_location("train_codet", os.path.join(tools, "train_codet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    full, part = os.path.join(str(tmp_path), "full"), os.path.join(str(tmp_path), "part")
    common = ["--data", "synthetic", "--com", "lowerbound", "--steps", "10", "--batch", "1", "--num_agent", "2"]
    mod.main(common + ["--nepoch", "3", "--logpath", full])
    lrs = []
    for ep in (1, 2, 3):
        ck = torch.load(os.path.join(full, "epoch_%d.pth" % ep), map_location="cpu")
        assert ck["epoch"] == ep and {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict"} <= set(ck)
        assert ck["scheduler_state_dict"]["last_epoch"] == 10 * ep                      # the schedule never restarts
        assert float(ck["optimizer_state_dict"]["state"][0]["step"]) == 10 * ep               # neither do Adam's moments
        lrs.append(ck["optimizer_state_dict"]["param_groups"][0]["lr"])
    assert lrs[0] == pytest.approx(1e-3) and lrs[1] == pytest.approx(3e-4) and lrs[2] == pytest.approx(9e-5)
    capsys.readouterr()
    mod.main(common + ["--nepoch", "3", "--logpath", part, "--resume", os.path.join(full, "epoch_1.pth")])
    out = capsys.readouterr().out
    assert "epoch 2 " in out and "epoch 3 " in out and "epoch 1 " not in out
    assert not os.path.exists(os.path.join(part, "epoch_1.pth"))
    ck = torch.load(os.path.join(part, "epoch_3.pth"), map_location="cpu")
    assert ck["scheduler_state_dict"]["last_epoch"] == 30 and ck["optimizer_state_dict"]["param_groups"][0]["lr"] == pytest.approx(9e-5)
    with pytest.raises(RuntimeError, match="Missing key|Unexpected key|size mismatch"):   # strict loading: a foreign checkpoint is refused
        mod.main(["--data", "synthetic", "--com", "v2v", "--steps", "1", "--nepoch", "1", "--resume", os.path.join(full, "epoch_1.pth")])


def test_seg_drivers_on_a_parsed_dataset(device, tmp_path, capsys):
    """Row f-2 for segmentation: a V2X-Sim-seg tree (README.md:66-79 layout, 0.npy with 'bev_seg') written from synthetic scenes ->
    tools/seg/train_seg.py --data <dir> (GPU densify shared with the det reader) -> tools/seg/test_seg.py --data <dir> --resume.
    12 frames x 8 epochs learn the vehicle class well above chance; the confusion matrix counts every labelled cell once."""
    import importlib.util
    import sys
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.datasets import write_seg_sample
    from v2x_sim_amd.utils import synthetic_scene
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "seg")
    sys.path.insert(0, tools)

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    A = 5
    cfg = Config("train")
    grid = ops.VoxelGrid()
    root = os.path.join(str(tmp_path), "V2X-Sim-seg")
    for split, frames, seed0 in (("train", 12, 41000), ("test", 3, 42000)):
        for f in range(frames):
            sc = synthetic_scene.make_scene(A, seed=seed0 + f)
            bits = ops.voxelize_bits(torch.from_numpy(sc["points"]).to(device), torch.from_numpy(sc["n_pts"]).to(device), grid)
            idx, cnt = ops.bits_to_indices(bits, grid.dims[2], 32768)
            idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
            for a in range(A):
                write_seg_sample(root, split, a, 5, f, idx[a, :cnt[a]], sc["trans"][a], A, synthetic_scene.seg_labels(sc["gt_boxes"][a], cfg))
    logdir = os.path.join(str(tmp_path), "log")
    load("train_seg").main(["--data", os.path.join(root, "train"), "--com", "v2v", "--nepoch", "8", "--batch", "2", "--logpath", logdir])
    ck = torch.load(os.path.join(logdir, "epoch_8.pth"), map_location="cpu")
    assert ck["epoch"] == 8 and "optimizer_state_dict" in ck
    res = load("test_seg").main(["--data", os.path.join(root, "test"), "--com", "v2v", "--resume", os.path.join(logdir, "epoch_8.pth")])
    print(capsys.readouterr().out[-400:])
    assert int(res["confusion"].sum()) == 3 * A * 256 * 256
    assert float(res["iou"][0]) > 0.95 and float(res["iou"][1]) > 0.25, res["iou"]


@pytest.mark.parametrize("engine,com", [("hip", "v2v"), ("hip-graph", "lowerbound")])
def test_train_driver_engine_flag(device, tmp_path, capsys, engine, com, tune):
    """tools/det/train_codet.py --engine hip | hip-graph: the driver trains on the hand-written kernels (bf16 NHWC graph; hip-graph = every
    step one hipGraph replay), the loss falls, and the checkpoint loads into the inference path."""
    import importlib.util
    tune("TRAIN_HIP", 0)      # main() sets these; the fixture restores them afterwards
    tune("TRAIN_GRAPH", 0)
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "det")
    spec = importlib.util.spec_from_file_location("train_codet", os.path.join(tools, "train_codet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    logdir = os.path.join(str(tmp_path), "log")
    mod.main(["--data", "synthetic", "--com", com, "--steps", "60", "--batch", "1", "--num_agent", "2", "--logpath", logdir, "--engine", engine, "--log"])
    from v2x_sim_amd import tuning
    assert tuning.get("TRAIN_HIP") == 1 and tuning.get("TRAIN_GRAPH") == (1 if engine == "hip-graph" else 0)
    out = capsys.readouterr().out
    print(out[-300:])
    ckpt = torch.load(os.path.join(logdir, "epoch_1.pth"), map_location="cpu")
    assert "model_state_dict" in ckpt and all(torch.isfinite(v).all() for v in ckpt["model_state_dict"].values() if v.is_floating_point())
    m = [float(x) for x in __import__("re").findall(r"mean loss of the last \d+ steps ([0-9.]+)", out)]
    assert m and m[-1] < 2.5, out[-300:]          # starts near 3.4 (focal loss at the prior): 60 steps bring it well below
