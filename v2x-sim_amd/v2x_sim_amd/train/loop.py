"""Training loops shared by tools/det/train_codet.py and the trained-detector parity test.

Input side stays on the HIP path: sweeps are voxelised on the GPU (ops.voxelize_bits -> ops.bits_to_dense) into the
dense (A*B, 1, X, Y, Z) occupancy the reference Dataset yields; targets arrive sparse and are scattered on the device.
"""
import torch

from .. import ops, packing, tuning
from ..utils import postprocess, synthetic_scene
from ..utils.CoDetModule import FaFModule


def init_for_training(model, fg_prior=0.01, seed=0):
    """Kaiming-normal conv weights, identity BN, and a classification bias that starts every anchor at
    p(foreground) = fg_prior (the usual focal-loss initialisation)."""
    import math
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                m.bias.zero_()
            elif isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.reset_parameters()
                m.reset_running_stats()
        b = model.classification.conv2.bias.view(-1, model.category_num)
        half = 0.5 * math.log((1.0 - fg_prior) / fg_prior)
        b[:, 0], b[:, 1] = half, -half
        model.classification.conv2.weight.mul_(0.1)
        model.regression.box_prediction[3].weight.mul_(0.1)
    return model


def synthetic_batch_on_device(config, frames, agents, seed, device, grid=None, anchors=None, with_targets=True, **scene_kw):
    """-> data dict in FaFModule.step / predict_all's format (+ 'gt_boxes'[agent][frame] host arrays)."""
    anchors = postprocess.build_anchor_map(config) if anchors is None else anchors
    grid = grid or ops.VoxelGrid(config.voxel_size, config.area_extents)
    b = synthetic_scene.make_batch(frames, agents, seed=seed, anchors=anchors if with_targets else None, targets="sparse",
                                   **scene_kw)
    pts = torch.from_numpy(b["points"]).to(device)
    bits = ops.voxelize_bits(pts, torch.from_numpy(b["n_pts"]).to(device), grid)
    data = {"bev_seq": ops.bits_to_dense(bits, grid.dims[2])[:, None],
            "trans_matrices": torch.from_numpy(b["trans"]).to(device), "num_agent": torch.from_numpy(b["num_agent"]),
            "gt_boxes": b["gt_boxes"]}
    if with_targets:
        data["labels"], data["reg_targets"], data["reg_loss_mask"] = synthetic_scene.dense_targets_on_device(
            b["pos"], b["pos_reg"], frames * agents, anchors.shape, device)
    return data


def make_optimizer(model, lr, total_steps):
    """Adam + the step-wise MultiStepLR (x0.3 at 60 % and 85 % of `total_steps`) every loop here uses.  Drivers that
    train for several epochs create the pair ONCE over the whole run and hand it to the loops, so that Adam's moments and
    the decay schedule survive epoch boundaries (and checkpoints)."""
    if tuning.get("TRAIN_GRAPH") == 1 and next(model.parameters()).is_cuda:
        # hipGraph-captured steps (train/graph_step.py): step counter and learning rate live on the device; the scheduler updates the
        # lr tensor in place, so the captured optimizer step sees every decay
        dev = next(model.parameters()).device
        opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(float(lr), device=dev), capturable=True, fused=True)
    else:
        # fused: ONE multi-tensor kernel per step on the device (the default per-parameter form issues two tiny elementwise launches per
        # parameter: 184 of them = 0.47 ms of a 10-map FaFNet step)
        opt = torch.optim.Adam(model.parameters(), lr=lr, fused=next(model.parameters()).is_cuda)
    packing.watch_optimizer(opt)     # a fused step does not bump the parameters' version counters: the packed-weight caches need the hook
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(total_steps * 0.6), int(total_steps * 0.85)], gamma=0.3)
    return opt, sched


def train_synthetic(model, config, steps, frames_per_step=2, lr=1e-3, seed=0, device="cuda:0", log=None, agents=None,
                    opt=None, sched=None, **scene_kw):
    """Adam on freshly generated synthetic scenes (never the same scene twice).  -> list of (loss, cls, loc).
    opt / sched: the run's optimizer and scheduler (make_optimizer); created here for a single self-contained run."""
    agents = agents or model.agent_num
    device = torch.device(device)
    model.to(device)
    if opt is None:
        opt, sched = make_optimizer(model, lr, steps)
    module = FaFModule(model, None, config, opt, 0)
    grid = ops.VoxelGrid(config.voxel_size, config.area_extents)
    hist = []
    for it in range(steps):
        data = synthetic_batch_on_device(config, frames_per_step, agents, seed * 1000003 + it, device, grid, module.anchors,
                                         **scene_kw)
        hist.append(module.step(data, frames_per_step, agents))
        if sched is not None:
            sched.step()
        if log and (it % log == 0 or it == steps - 1):
            print("step %4d  loss %.4f  cls %.4f  loc %.4f" % ((it,) + hist[-1]), flush=True)
    model.eval()
    return hist


def dataset_batch_on_device(samples, grid, anchors, device):
    """Parsed-dataset samples (datasets.V2XSimDet with densify='none': sparse voxel indices + poses + gt boxes) -> the
    data dict of FaFModule.step.  The sweep is densified on the GPU (v2x_indices_to_bits -> v2x_bits_to_dense_f32); the
    anchor targets upstream's Dataset would return (label_one_hot, reg_target, reg_loss_mask) are built from the stored
    ground-truth boxes (synthetic_scene.anchor_targets_sparse, the assignment rule of DESIGN.md section 3.7) and scattered
    on the device."""
    import numpy as np
    B, A = len(samples), len(samples[0])
    items = [(a, b) for a in range(A) for b in range(B)]                       # agent-major, as the models batch
    cap = max(1, max(samples[b][a][0].shape[0] for a, b in items))
    idx = np.zeros((len(items), cap, 3), np.int32)
    cnt = np.zeros((len(items),), np.int32)
    pos, reg = [], []
    for m, (a, b) in enumerate(items):
        it = samples[b][a][0]
        idx[m, :it.shape[0]] = it
        cnt[m] = it.shape[0]
        p, r = synthetic_scene.anchor_targets_sparse(samples[b][a][12], anchors)
        pos.append(np.concatenate([np.full((p.shape[0], 1), m, np.int64), p], 1))
        reg.append(r)
    bits = ops.indices_to_bits(torch.from_numpy(idx).to(device), torch.from_numpy(cnt).to(device), grid)
    data = {"bev_seq": ops.bits_to_dense(bits, grid.dims[2])[:, None],
            "trans_matrices": torch.from_numpy(np.stack([np.stack([samples[b][a][11] for a in range(A)])
                                                         for b in range(B)])).to(device),
            "num_agent": torch.tensor([[samples[b][a][10] for a in range(A)] for b in range(B)]),
            "gt_boxes": [[samples[b][a][12] for b in range(B)] for a in range(A)]}
    data["labels"], data["reg_targets"], data["reg_loss_mask"] = synthetic_scene.dense_targets_on_device(
        np.concatenate(pos), np.concatenate(reg), len(items), anchors.shape, device)
    return data


def train_dataset(model, config, dataset, epochs=1, batch=2, lr=1e-3, seed=0, device="cuda:0", log=None, num_workers=0,
                  opt=None, sched=None):
    """Adam over a parsed dataset (datasets.V2XSimDet, densify='none'), shuffled per epoch; upstream's
    train_codet.py loop with the densify and the target scatter moved to the GPU.  -> list of (loss, cls, loc) per step.
    opt / sched as in train_synthetic."""
    from torch.utils.data import DataLoader
    device = torch.device(device)
    model.to(device)
    own = opt is None
    if own:
        opt = torch.optim.Adam(model.parameters(), lr=lr)
    module = FaFModule(model, None, config, opt, 0)
    grid = ops.VoxelGrid(config.voxel_size, config.area_extents)
    loader = DataLoader(dataset, batch_size=batch, shuffle=True, drop_last=False, num_workers=num_workers,
                        collate_fn=lambda x: x, generator=torch.Generator().manual_seed(seed))
    steps = epochs * len(loader)
    if own:
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(steps * 0.6), int(steps * 0.85)], gamma=0.3)
    hist = []
    for ep in range(epochs):
        for samples in loader:
            data = dataset_batch_on_device(samples, grid, module.anchors, device)
            hist.append(module.step(data, len(samples), len(samples[0])))
            if sched is not None:
                sched.step()
            if log and (len(hist) % log == 1 or len(hist) == steps):
                print("epoch %d step %4d  loss %.4f  cls %.4f  loc %.4f" % ((ep + 1, len(hist)) + hist[-1]), flush=True)
    model.eval()
    return hist
