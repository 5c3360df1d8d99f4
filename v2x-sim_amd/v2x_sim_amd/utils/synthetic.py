"""Seeded synthetic weights / inputs: there is no network for the
V2X-Sim dataset or the released checkpoints (README.md:45-46), so benches and tests use
random-init weights of the real architecture and synthetic sweeps of the real shape."""
import math

import numpy as np
import torch
import torch.nn as nn


def init_synthetic_weights(model, seed=0):
    """He-normal conv/linear weights (activations stay O(1) through ~25 layers) and a
    non-trivial eval-mode BatchNorm (random gamma/beta/mean/var) so that BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
        for name, p in model.named_parameters():
            if name.endswith("weight_ih_l0") or name.endswith("weight_hh_l0"):
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(1.0 / fan_in))
            elif name.endswith("bias_ih_l0") or name.endswith("bias_hh_l0"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    model.eval()
    return model


def synthetic_points(n_clouds, n_pts, seed=0):
    """x,y ~ U(-40,40), z ~ U(-5,4), intensity ~ U(0,1); ~36 % fall outside the BEV extents."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n_clouds, n_pts, 4), dtype=np.float32)
    pts[..., 0] = rng.uniform(-40, 40, (n_clouds, n_pts))
    pts[..., 1] = rng.uniform(-40, 40, (n_clouds, n_pts))
    pts[..., 2] = rng.uniform(-5, 4, (n_clouds, n_pts))
    pts[..., 3] = rng.uniform(0, 1, (n_clouds, n_pts))
    return pts


def synthetic_poses(batch, agents, seed=0):
    """Random SE(2) poses; trans[b, i, j] = inv(P_i) @ P_j maps agent j's frame into agent i's
    (the matrix upstream stores as trans_matrices[b, i, j] and feeds to feature_transformation)."""
    rng = np.random.default_rng(seed)
    T = np.zeros((batch, agents, agents, 4, 4), dtype=np.float32)
    for b in range(batch):
        P = []
        for a in range(agents):
            yaw = rng.uniform(-math.pi, math.pi)
            x, y = rng.uniform(-20, 20, 2)
            M = np.eye(4)
            M[0, 0], M[0, 1], M[1, 0], M[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
            M[0, 3], M[1, 3] = x, y
            P.append(M)
        for i in range(agents):
            for j in range(agents):
                T[b, i, j] = (np.linalg.inv(P[i]) @ P[j]).astype(np.float32)
    return T
