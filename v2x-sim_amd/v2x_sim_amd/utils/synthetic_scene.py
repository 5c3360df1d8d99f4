"""Seeded synthetic collaborative scenes WITH ground truth.

There is no network for the V2X-Sim dataset (README.md:42-48 are external downloads), so the training path and the
trained-detector mAP parity test need a learnable stand-in with the real shapes: a world of car-sized boxes seen by
A agents at random SE(2) poses.  Per agent the sweep holds (i) points on the roof and the sides of every car within
sensor range, (ii) a ground plane, (iii) sparse clutter -- all expressed in that agent's frame -- and the ground truth
is the list of cars inside its BEV extents, [x, y, w, h, yaw] in the convention of utils/postprocess.py
(w along the box's local x, h along its local y).

`anchor_targets` is the label half of upstream's Dataset tuple (label_one_hot, reg_target, reg_loss_mask; upstream
coperception/datasets/V2XSimDet.py builds them from gt_max_iou -- not in /root/reference).  Build-owned assignment
(frozen here): an anchor is positive iff its cell centre lies within POS_RADIUS of a car centre and it is the car-size
anchor whose heading is closest to the car's (mod pi); the regression code is the inverse of postprocess.decode_boxes.
"""
import math

import numpy as np

POS_RADIUS = 0.75      # metres
SENSOR_RANGE = 30.0    # metres: an agent only receives returns from cars closer than this
CAR_W, CAR_H = 2.0, 4.4


def _pose(yaw, x, y):
    M = np.eye(4)
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
    M[0, 3], M[1, 3] = x, y
    return M


def make_scene(agents=5, n_cars=24, seed=0, ground_pts=6000, clutter_pts=1200, pts_per_car=400, max_pts=16384,
               sensor_range=SENSOR_RANGE, gt="own"):
    """-> dict(points (A, max_pts, 4) f32 zero-padded, n_pts (A,) i32, trans (A, A, 4, 4) f32 = the pose of agent j w.r.t.
    agent i in the warp's axis convention (see the end of this function), trans_geo = inv(P_i) @ P_j, gt_boxes list of
    (G_a, 5) f32 per agent).
    gt='own': an agent's ground truth = the cars inside its BEV extents AND its own sensor range;
    gt='any': ... inside its BEV extents and in range of ANY agent -- cars the ego cannot see itself but a neighbour can
    (with a short sensor_range this is the setting in which collaboration must help, tests/test_gpu_train.py)."""
    rng = np.random.default_rng(seed)
    P = [_pose(rng.uniform(-math.pi, math.pi), *rng.uniform(-12, 12, 2)) for _ in range(agents)]
    cars = []
    tries = 0
    while len(cars) < n_cars and tries < 2000:
        tries += 1
        c = rng.uniform(-40, 40, 2)
        if all(np.hypot(*(c - o[:2])) > 6.0 for o in cars):
            cars.append(np.array([c[0], c[1], CAR_W + rng.uniform(-0.15, 0.15), CAR_H + rng.uniform(-0.4, 0.4),
                                  rng.uniform(-math.pi, math.pi)]))
    cars = np.asarray(cars)
    points = np.zeros((agents, max_pts, 4), np.float32)
    n_pts = np.zeros((agents,), np.int32)
    gts = []
    for a in range(agents):
        inv = np.linalg.inv(P[a])
        ax, ay = P[a][0, 3], P[a][1, 3]
        chunks = []
        for cx, cy, w, h, yaw in cars:
            if math.hypot(cx - ax, cy - ay) > sensor_range:
                continue
            n = pts_per_car
            lx = rng.uniform(-w / 2, w / 2, n)
            ly = rng.uniform(-h / 2, h / 2, n)
            lz = np.full(n, -0.25) + rng.normal(0, 0.03, n)       # roof
            side = rng.random(n) < 0.5                             # half of the returns come from the sides
            edge = rng.integers(0, 4, n)
            lx = np.where(side & (edge == 0), w / 2, np.where(side & (edge == 1), -w / 2, lx))
            ly = np.where(side & (edge == 2), h / 2, np.where(side & (edge == 3), -h / 2, ly))
            lz = np.where(side, rng.uniform(-1.6, -0.3, n), lz)
            c, s = math.cos(yaw), math.sin(yaw)
            chunks.append(np.stack([cx + lx * c - ly * s, cy + lx * s + ly * c, lz], 1))
        r = sensor_range * np.sqrt(rng.random(ground_pts))
        t = rng.uniform(0, 2 * math.pi, ground_pts)
        chunks.append(np.stack([ax + r * np.cos(t), ay + r * np.sin(t), -1.8 + rng.normal(0, 0.03, ground_pts)], 1))
        chunks.append(np.stack([ax + rng.uniform(-32, 32, clutter_pts), ay + rng.uniform(-32, 32, clutter_pts),
                                rng.uniform(-1.7, 1.5, clutter_pts)], 1))
        w_pts = np.concatenate(chunks)
        local = w_pts @ inv[:3, :3].T + inv[:3, 3]
        local = local[rng.permutation(local.shape[0])[:max_pts]]
        points[a, :local.shape[0], :3] = local.astype(np.float32)
        points[a, :local.shape[0], 3] = rng.random(local.shape[0]).astype(np.float32)
        n_pts[a] = local.shape[0]
        yaw_a = math.atan2(P[a][1, 0], P[a][0, 0])
        ctr = cars[:, :2] @ inv[:2, :2].T + inv[:2, 3]
        if gt == "any":
            seen = np.zeros(cars.shape[0], bool)
            for b in range(agents):
                seen |= np.hypot(cars[:, 0] - P[b][0, 3], cars[:, 1] - P[b][1, 3]) <= sensor_range
        else:
            seen = np.hypot(cars[:, 0] - ax, cars[:, 1] - ay) <= sensor_range
        inside = (np.abs(ctr[:, 0]) < 29.0) & (np.abs(ctr[:, 1]) < 29.0) & seen
        g = np.concatenate([ctr, cars[:, 2:4], (cars[:, 4:5] - yaw_a)], 1)[inside]
        gts.append(g.astype(np.float32))
    # trans_geo[i, j] = inv(P_i) @ P_j: the geometric transform, agent j's BEV frame -> agent i's (p_i = R p_j + t).
    # trans[i, j] = the same pose in the axis convention upstream's warp PRESUMES of the dataset's trans_matrices: the
    # formula shifts the WIDTH axis (BEV y) by 4*T03/128 and the HEIGHT axis (BEV x) by -4*T13/128, which aligns features
    # iff T03 = -t_y and T13 = t_x (sensor axes rotated 90 degrees against the BEV array axes; the rotation block commutes
    # and is unchanged).  Verified numerically against the oracle's feature_transformation (tests/test_oracle_cpu.py):
    # with the plain (t_x, t_y) a warped blob never lands on its cell, with (-t_y, t_x) it always does.
    T_geo = np.zeros((agents, agents, 4, 4), np.float32)
    T = np.zeros((agents, agents, 4, 4), np.float32)
    for i in range(agents):
        for j in range(agents):
            T_geo[i, j] = (np.linalg.inv(P[i]) @ P[j]).astype(np.float32)
            T[i, j] = T_geo[i, j]
            T[i, j, 0, 3], T[i, j, 1, 3] = -T_geo[i, j, 1, 3], T_geo[i, j, 0, 3]
    return {"points": points, "n_pts": n_pts, "trans": T, "trans_geo": T_geo, "gt_boxes": gts}


def anchor_targets_sparse(gt_boxes, anchors):
    """-> (pos (P, 3) int64 [ix, iy, anchor], reg (P, 6) f32): the positive anchors and their regression codes."""
    xs, ys = anchors[:, 0, 0, 0], anchors[0, :, 0, 1]
    a_yaw = np.arctan2(anchors[0, 0, :, 4], anchors[0, 0, :, 5])
    a_w, a_h = anchors[0, 0, :, 2], anchors[0, 0, :, 3]
    pos, reg = {}, {}
    for gx, gy, gw, gh, gyaw in np.asarray(gt_boxes, np.float64).reshape(-1, 5):
        # car-size anchors only (closest in area), heading difference folded to [-pi/2, pi/2)
        size_cost = np.abs(np.log(a_w * a_h / (gw * gh)))
        d = (gyaw - a_yaw + math.pi / 2) % math.pi - math.pi / 2
        k = int(np.argmin(size_cost * 10.0 + np.abs(d)))
        ix = np.nonzero(np.abs(xs - gx) <= POS_RADIUS)[0]
        iy = np.nonzero(np.abs(ys - gy) <= POS_RADIUS)[0]
        for i in ix:
            for j in iy:
                if math.hypot(xs[i] - gx, ys[j] - gy) > POS_RADIUS:
                    continue
                pos[(int(i), int(j), k)] = (gx - xs[i], gy - ys[j], math.log(gw / a_w[k]), math.log(gh / a_h[k]),
                                            math.sin(d[k]), math.cos(d[k]))   # later cars overwrite (cars never overlap)
    keys = sorted(pos)
    return (np.asarray(keys, np.int64).reshape(-1, 3), np.asarray([pos[k] for k in keys], np.float32).reshape(-1, 6))


def anchor_targets(gt_boxes, anchors):
    """gt_boxes (G, 5), anchors (X, Y, A, 6) from postprocess.build_anchor_map ->
    label_one_hot (X, Y, A, 2) f32, reg_target (X, Y, A, 1, 6) f32, reg_loss_mask (X, Y, A, 1) bool."""
    X, Y, A, _ = anchors.shape
    label = np.zeros((X, Y, A, 2), np.float32)
    label[..., 0] = 1.0
    reg = np.zeros((X, Y, A, 1, 6), np.float32)
    mask = np.zeros((X, Y, A, 1), bool)
    pos, code = anchor_targets_sparse(gt_boxes, anchors)
    if pos.shape[0]:
        i, j, k = pos[:, 0], pos[:, 1], pos[:, 2]
        label[i, j, k] = (0.0, 1.0)
        reg[i, j, k, 0] = code
        mask[i, j, k, 0] = True
    return label, reg, mask


def make_batch(frames, agents=5, seed=0, anchors=None, targets="dense", **kw):
    """`frames` scenes in the agent-major layout the models consume: points (A*B, max_pts, 4), n_pts (A*B,),
    trans (B, A, A, 4, 4), num_agent (B, A), gt_boxes[a][b].  With `anchors`, the training targets:
    targets='dense'  -> labels (A*B, X, Y, A', 2), reg_targets (A*B, X, Y, A', 1, 6), reg_loss_mask (A*B, X, Y, A', 1)
                        (upstream's Dataset fields);
    targets='sparse' -> pos (P, 4) int64 [item, ix, iy, anchor], pos_reg (P, 6): the same information without the
                        94 MB of zeros (dense_targets_on_device rebuilds the dense tensors on the GPU)."""
    scenes = [make_scene(agents, seed=seed * 100003 + b, **kw) for b in range(frames)]
    out = {"points": np.stack([scenes[b]["points"][a] for a in range(agents) for b in range(frames)]),
           "n_pts": np.asarray([scenes[b]["n_pts"][a] for a in range(agents) for b in range(frames)], np.int32),
           "trans": np.stack([s["trans"] for s in scenes]),
           "trans_geo": np.stack([s["trans_geo"] for s in scenes]),
           "num_agent": np.full((frames, agents), agents, np.int64),
           "gt_boxes": [[scenes[b]["gt_boxes"][a] for b in range(frames)] for a in range(agents)]}
    if anchors is not None and targets == "dense":
        tg = [anchor_targets(scenes[b]["gt_boxes"][a], anchors) for a in range(agents) for b in range(frames)]
        out["labels"] = np.stack([t[0] for t in tg])
        out["reg_targets"] = np.stack([t[1] for t in tg])
        out["reg_loss_mask"] = np.stack([t[2] for t in tg])
    elif anchors is not None:
        pos, reg = [], []
        for m, (a, b) in enumerate((a, b) for a in range(agents) for b in range(frames)):
            p, r = anchor_targets_sparse(scenes[b]["gt_boxes"][a], anchors)
            pos.append(np.concatenate([np.full((p.shape[0], 1), m, np.int64), p], 1))
            reg.append(r)
        out["pos"], out["pos_reg"] = np.concatenate(pos), np.concatenate(reg)
    return out


def dense_targets_on_device(pos, pos_reg, n_items, anchors_shape, device):
    """sparse targets -> the dense (labels, reg_targets, reg_loss_mask) tensors of upstream's Dataset, built on `device`."""
    import torch
    X, Y, A = anchors_shape[:3]
    labels = torch.zeros((n_items, X, Y, A, 2), dtype=torch.float32, device=device)
    labels[..., 0] = 1.0
    reg = torch.zeros((n_items, X, Y, A, 1, 6), dtype=torch.float32, device=device)
    mask = torch.zeros((n_items, X, Y, A, 1), dtype=torch.bool, device=device)
    p = torch.as_tensor(pos, device=device)
    if p.shape[0]:
        m, i, j, k = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
        labels[m, i, j, k] = torch.tensor([0.0, 1.0], device=device)
        reg[m, i, j, k, 0] = torch.as_tensor(pos_reg, device=device)
        mask[m, i, j, k, 0] = True
    return labels, reg, mask


def seg_labels(gt_boxes, config, n_classes=8):
    """(G, 5) boxes -> (X, Y) uint8 BEV label map: class 1 = vehicle footprint (cell centre inside the rotated box),
    class 0 = everything else.  (V2X-Sim's seg set has 8 classes, README.md:36; the synthetic scenes only contain cars.)"""
    X, Y = config.map_dims[0], config.map_dims[1]
    vx, vy = config.voxel_size[0], config.voxel_size[1]
    x0, y0 = config.area_extents[0][0], config.area_extents[1][0]
    lab = np.zeros((X, Y), np.uint8)
    for gx, gy, gw, gh, gyaw in np.asarray(gt_boxes, np.float64).reshape(-1, 5):
        r = 0.5 * math.hypot(gw, gh)
        i0, i1 = max(0, int((gx - r - x0) / vx)), min(X, int((gx + r - x0) / vx) + 2)
        j0, j1 = max(0, int((gy - r - y0) / vy)), min(Y, int((gy + r - y0) / vy) + 2)
        if i0 >= i1 or j0 >= j1:
            continue
        cx = x0 + (np.arange(i0, i1) + 0.5) * vx - gx
        cy = y0 + (np.arange(j0, j1) + 0.5) * vy - gy
        c, s_ = math.cos(gyaw), math.sin(gyaw)
        lx = cx[:, None] * c + cy[None, :] * s_          # cell centre in the box frame
        ly = -cx[:, None] * s_ + cy[None, :] * c
        lab[i0:i1, j0:j1][(np.abs(lx) <= gw / 2) & (np.abs(ly) <= gh / 2)] = 1
    return lab
