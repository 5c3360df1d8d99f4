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


def synthetic_ring_sweep(n_clouds, n_pts=65536, seed=0, beams=32, sensor_height=1.8):
    """A 32-beam spinning-LiDAR sweep as V2X-Sim's sensors produce them (nuScenes-style LIDAR_TOP, /root/reference/README.md:50-64), in the sensor
    frame: `beams` elevation angles from -30.7 to +10.7 degrees, n_pts / beams azimuth steps per ring.  Downward beams hit the ground plane
    (z = -sensor_height) at r = h / tan(-elevation) unless one of ~40 random boxes (vehicles, walls) stands in the way; upward beams hit a box /
    facade or return nothing (such points are parked far outside the BEV extents, as a driver's max-range returns are).  2 cm range noise.  What
    the uniform generator lacks and this has: the steep inner rings put their 2 048 returns on circles of a few metres radius -- tens of points
    per 0.25 m voxel (duplicates that collide in one LDS word), empty space between the rings, and a dense / sparse mix across the map."""
    rng = np.random.default_rng(seed)
    per = n_pts // beams
    elev = np.deg2rad(np.linspace(-30.67, 10.67, beams))
    out = np.empty((n_clouds, per * beams, 4), dtype=np.float32)
    for c in range(n_clouds):
        az = (np.arange(per) + rng.uniform(0, 1)) * (2 * math.pi / per)
        azg, elg = np.meshgrid(az, elev)                         # (beams, per)
        dx, dy, dz = np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(dz < -1e-3, -sensor_height / dz, np.inf)
        # boxes: axis-aligned in the sensor frame for simplicity (centre, half extents, top); a ray stops at the nearest box face it enters
        nb = 40
        bc = rng.uniform(-45, 45, (nb, 2))
        bh = np.stack([rng.uniform(0.8, 6.0, nb), rng.uniform(0.8, 3.0, nb)], 1)
        top = rng.uniform(-0.3, 4.0, nb)
        for k in range(nb):
            if (np.abs(bc[k]) - bh[k]).max() < 2.5:
                continue                                         # nothing within 2.5 m of the sensor (the vehicle itself is there)
            with np.errstate(divide="ignore", invalid="ignore"):
                tx0, tx1 = (bc[k, 0] - bh[k, 0]) / dx, (bc[k, 0] + bh[k, 0]) / dx
                ty0, ty1 = (bc[k, 1] - bh[k, 1]) / dy, (bc[k, 1] + bh[k, 1]) / dy
            tn = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1))
            tf = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
            hit = (tn < tf) & (tn > 0) & (-sensor_height + 0 * tn <= top[k]) & (dz * tn <= top[k]) & (tn < r)
            r = np.where(hit, tn, r)
        r = r + rng.normal(0, 0.02, r.shape)
        lost = ~np.isfinite(r) | (r > 70.0)
        r = np.where(lost, 200.0, r)                             # no return: far outside the extents
        pts = np.stack([r * dx, r * dy, r * dz, rng.uniform(0, 1, r.shape)], -1).reshape(-1, 4)
        out[c] = pts[rng.permutation(pts.shape[0])].astype(np.float32)      # packets do not arrive ring by ring
    return out


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
