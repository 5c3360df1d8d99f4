"""Training loops shared by tools/det/train_codet.py and the trained-detector parity test.

Input side stays on the HIP path: sweeps are voxelised on the GPU (ops.voxelize_bits -> ops.bits_to_dense) into the
dense (A*B, 1, X, Y, Z) occupancy the reference Dataset yields; targets arrive sparse and are scattered on the device.
"""
import torch

from .. import ops
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


def train_synthetic(model, config, steps, frames_per_step=2, lr=1e-3, seed=0, device="cuda:0", log=None, agents=None,
                    **scene_kw):
    """Adam on freshly generated synthetic scenes (never the same scene twice).  -> list of (loss, cls, loc)."""
    agents = agents or model.agent_num
    device = torch.device(device)
    model.to(device)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[int(steps * 0.6), int(steps * 0.85)], gamma=0.3)
    module = FaFModule(model, None, config, opt, 0)
    grid = ops.VoxelGrid(config.voxel_size, config.area_extents)
    hist = []
    for it in range(steps):
        data = synthetic_batch_on_device(config, frames_per_step, agents, seed * 1000003 + it, device, grid, module.anchors,
                                         **scene_kw)
        hist.append(module.step(data, frames_per_step, agents))
        sched.step()
        if log and (it % log == 0 or it == steps - 1):
            print("step %4d  loss %.4f  cls %.4f  loc %.4f" % ((it,) + hist[-1]), flush=True)
    model.eval()
    return hist
