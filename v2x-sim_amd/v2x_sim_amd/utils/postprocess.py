"""Detection post-processing and mAP, host-side numpy like the
reference's own Python post-processing.

Mirrors (from recollection -- the files are not in /root/reference; README.md:36 names the
detection task, README.md:101 the benchmark scripts that call them):
  * coperception/utils/postprocess.py::apply_nms_det  -- softmax foreground score, score
    threshold, 'faf' anchor decode, NMS on the axis-aligned "stand-up" boxes of the rotated boxes;
  * coperception/utils/mean_ap.py::eval_map           -- mmdet-style AP (area under the
    monotone precision envelope) with ROTATED-box IoU.  Upstream uses shapely polygons; shapely is
    not available here, so the convex-polygon intersection is a Sutherland-Hodgman clip.

Build-owned spec, frozen here because upstream cannot be consulted:
  box code (x, y, w, h, sin, cos):  x = xa + dx,  y = ya + dy,  w = wa*exp(dw),  h = ha*exp(dh),
  heading = atan2(sa, ca) + atan2(ds, dc);   score = softmax(cls)[..., 1] per anchor;
  keep score >= 0.7 (upstream's test-time threshold), NMS IoU threshold 0.01 on stand-up boxes.
This module never touches the GPU kernels; it consumes the fp32 logits they (or the oracle) emit,
so "mAP parity" = same function applied to both sets of logits.
"""
import math

import numpy as np


def build_anchor_map(config):
    """(X, Y, A, 6) anchors [x, y, w, h, sin, cos] at the BEV cell centres, metres."""
    X, Y = config.map_dims[0], config.map_dims[1]
    vx, vy = config.voxel_size[0], config.voxel_size[1]
    x0, y0 = config.area_extents[0][0], config.area_extents[1][0]
    xs = x0 + (np.arange(X) + 0.5) * vx
    ys = y0 + (np.arange(Y) + 0.5) * vy
    anc = np.zeros((X, Y, len(config.anchor_size), 6), dtype=np.float32)
    anc[..., 0] = xs[:, None, None]
    anc[..., 1] = ys[None, :, None]
    for a, (w, h, yaw) in enumerate(config.anchor_size):
        anc[:, :, a, 2], anc[:, :, a, 3] = w, h
        anc[:, :, a, 4], anc[:, :, a, 5] = math.sin(yaw), math.cos(yaw)
    return anc


def softmax_fg(cls_logits):
    """cls (..., 2) logits -> foreground probability."""
    z = cls_logits - cls_logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e[..., 1] / e.sum(axis=-1)


def decode_boxes(loc, anchors):
    """loc, anchors (..., 6) -> (..., 5) [x, y, w, h, yaw]."""
    x = anchors[..., 0] + loc[..., 0]
    y = anchors[..., 1] + loc[..., 1]
    w = anchors[..., 2] * np.exp(np.clip(loc[..., 2], -4.0, 4.0))
    h = anchors[..., 3] * np.exp(np.clip(loc[..., 3], -4.0, 4.0))
    yaw = np.arctan2(anchors[..., 4], anchors[..., 5]) + np.arctan2(loc[..., 4], loc[..., 5])
    return np.stack([x, y, w, h, yaw], axis=-1)


WH_AXES = ("w_along_heading", "h_along_heading")


def box_corners(boxes, wh_axis="w_along_heading"):
    """(N, 5) [x, y, w, h, yaw] -> (N, 4, 2) corners, counter-clockwise.  wh_axis (oracle/ASSUMPTIONS.md row 48, both readings): which extent
    runs along the box's heading -- "w_along_heading" (default) or "h_along_heading" (the rectangle with the extents exchanged);
    Config.box_wh_axis selects it for FaFModule."""
    x, y, w, h, yaw = (boxes[:, i] for i in range(5))
    if wh_axis == "h_along_heading":
        w, h = h, w
    elif wh_axis != "w_along_heading":
        raise ValueError("wh_axis must be one of %s" % (WH_AXES,))
    c, s = np.cos(yaw), np.sin(yaw)
    dx = np.stack([w / 2, -w / 2, -w / 2, w / 2], 1)
    dy = np.stack([h / 2, h / 2, -h / 2, -h / 2], 1)
    cx = x[:, None] + dx * c[:, None] - dy * s[:, None]
    cy = y[:, None] + dx * s[:, None] + dy * c[:, None]
    return np.stack([cx, cy], -1)


def standup(corners):
    """(N, 4, 2) -> (N, 4) axis-aligned [x1, y1, x2, y2]."""
    return np.concatenate([corners.min(1), corners.max(1)], 1)


def nms_standup(boxes_xyxy, scores, iou_thr=0.01, max_out=None):
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes_xyxy[:, i] for i in range(4))
    area = (x2 - x1) * (y2 - y1)
    keep = []
    while order.size:
        i = order[0]
        keep.append(i)
        if max_out and len(keep) >= max_out:
            break
        rest = order[1:]
        iw = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        ih = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = iw * ih
        iou = inter / (area[i] + area[rest] - inter + 1e-12)
        order = rest[iou <= iou_thr]
    return np.asarray(keep, dtype=np.int64)


def apply_nms_det(loc, cls, anchors, score_thr=0.7, nms_thr=0.01, max_out=None, wh_axis="w_along_heading"):
    """One agent.  loc (X, Y, A, 1, 6) or (X, Y, A, 6); cls (X*Y*A, 2); anchors (X, Y, A, 6).
    -> dict(boxes (M, 5), corners (M, 4, 2), scores (M,))."""
    loc = np.asarray(loc, dtype=np.float32).reshape(-1, 6)
    anchors = np.asarray(anchors, dtype=np.float32).reshape(-1, 6)
    score = softmax_fg(np.asarray(cls, dtype=np.float32).reshape(-1, 2))
    sel = np.nonzero(score >= score_thr)[0]
    if sel.size == 0:
        return {"boxes": np.zeros((0, 5), np.float32), "corners": np.zeros((0, 4, 2), np.float32),
                "scores": np.zeros((0,), np.float32)}
    if sel.size > 20000:
        # upstream's greedy NMS is O(kept x candidates): a detector that fires on a large part of the map (an untrained or diverged model)
        # keeps this host path busy for minutes per map -- say so instead of looking hung
        import warnings
        warnings.warn("apply_nms_det: %d of %d anchors pass the score threshold %.2f; the host NMS will take a long time "
                      "(is the model trained?)" % (sel.size, score.size, score_thr), RuntimeWarning, stacklevel=2)
    boxes = decode_boxes(loc[sel], anchors[sel])
    corners = box_corners(boxes, wh_axis)
    keep = nms_standup(standup(corners), score[sel], nms_thr, max_out)
    return {"boxes": boxes[keep], "corners": corners[keep], "scores": score[sel][keep]}


# ------------------------------------------------------------------ rotated IoU + AP
def _poly_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _clip(subject, a, b):
    """Sutherland-Hodgman: keep the part of `subject` left of the directed edge a->b."""
    out = []
    n = len(subject)
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_iou(c1, c2):
    """IoU of two convex quadrilaterals given as (4, 2) counter-clockwise corners."""
    poly = [tuple(p) for p in c1]
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, tuple(c2[i]), tuple(c2[(i + 1) % 4]))
    if len(poly) < 3:
        return 0.0
    inter = _poly_area(np.asarray(poly))
    union = _poly_area(np.asarray(c1)) + _poly_area(np.asarray(c2)) - inter
    return inter / union if union > 0 else 0.0


def eval_map(det_results, annotations, iou_thr=0.5):
    """Single-class mmdet-style AP.

    det_results: list (one per sample) of dict(corners (M,4,2), scores (M,));
    annotations: list of (G, 4, 2) ground-truth corner arrays.
    Returns (ap, dict(recall, precision, num_gt, num_det))."""
    scores, tps = [], []
    num_gt = 0
    for det, gt in zip(det_results, annotations):
        gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4, 2)
        num_gt += gt.shape[0]
        order = np.argsort(-det["scores"], kind="stable")
        taken = np.zeros(gt.shape[0], dtype=bool)
        gt_box = standup(gt) if gt.shape[0] else None
        for j in order:
            c = np.asarray(det["corners"][j], dtype=np.float64)
            best, best_g = 0.0, -1
            if gt.shape[0]:
                bb = np.concatenate([c.min(0), c.max(0)])
                cand = np.nonzero((gt_box[:, 0] <= bb[2]) & (gt_box[:, 2] >= bb[0]) &
                                  (gt_box[:, 1] <= bb[3]) & (gt_box[:, 3] >= bb[1]))[0]
                for g in cand:
                    v = rotated_iou(c, gt[g])
                    if v > best:
                        best, best_g = v, g
            scores.append(det["scores"][j])
            if best >= iou_thr and not taken[best_g]:
                taken[best_g] = True
                tps.append(1)
            else:
                tps.append(0)
    return average_precision(scores, tps, num_gt)


def average_precision(scores, tps, num_gt):
    """mmdet 'area' AP from per-detection (score, true-positive flag) pairs of ALL images (image-major, per image in descending
    score order) and the number of ground-truth boxes.  -> (ap, info) like eval_map."""
    scores, tps = np.asarray(scores, dtype=np.float64).reshape(-1), np.asarray(tps).reshape(-1).astype(np.int64)
    if num_gt == 0 or scores.size == 0:
        return 0.0, {"recall": np.zeros(0), "precision": np.zeros(0), "num_gt": num_gt, "num_det": int(scores.size)}
    order = np.argsort(-scores, kind="stable")
    tp = np.cumsum(tps[order])
    fp = np.cumsum(1 - tps[order])
    recall = tp / num_gt
    precision = tp / np.maximum(tp + fp, 1e-12)
    mrec = np.concatenate([[0.0], recall, [1.0]])
    mpre = np.concatenate([[0.0], precision, [0.0]])
    for i in range(mpre.size - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
    ap = float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))
    return ap, {"recall": recall, "precision": precision, "num_gt": num_gt, "num_det": int(scores.size)}


def eval_map_device(det_boxes, det_scores, det_count, gt_boxes_list, iou_thr=0.5):
    """eval_map with the IoU + matching step on the MI355X (v2x_match_detections): det_boxes (n, cap, 5), det_scores (n, cap),
    det_count (n,) as v2x_det_postprocess leaves them ON THE DEVICE (descending score per map), gt_boxes_list = per image a
    (G, 5) host array.  Only (score, tp) pairs come back to the host for the final sort + cumulative sums."""
    import torch
    from .. import ops
    n = det_boxes.shape[0]
    gt_cap = max(1, max((len(g) for g in gt_boxes_list), default=1))
    gt = np.zeros((n, gt_cap, 5), np.float32)
    gcount = np.zeros((n,), np.int32)
    for i, g in enumerate(gt_boxes_list):
        g = np.asarray(g, np.float32).reshape(-1, 5)
        gt[i, :g.shape[0]] = g
        gcount[i] = g.shape[0]
    dev = det_boxes.device
    if bool((det_count < 0).any()):
        # v2x_det_postprocess's overflow signal (more than `cap` candidates): clamping would score that map with ZERO detections while its
        # ground truth still counts.  The caller re-runs such maps through the host path (FaFModule.postprocess does) before scoring.
        bad = torch.nonzero(det_count < 0).flatten().tolist()
        raise ValueError("eval_map_device: maps %s overflowed the device candidate capacity (count < 0); run them through "
                         "apply_nms_det and pass their detections in, or raise the capacity" % bad[:8])
    cnt = det_count.to(torch.int32).contiguous()
    tp = ops.match_detections(det_boxes.contiguous(), cnt, torch.from_numpy(gt).to(dev), torch.from_numpy(gcount).to(dev), iou_thr)
    cnt_h, tp_h, sc_h = cnt.cpu().numpy(), tp.cpu().numpy(), det_scores.cpu().numpy()
    scores = np.concatenate([sc_h[i, :cnt_h[i]] for i in range(n)]) if n else np.zeros(0)
    tps = np.concatenate([tp_h[i, :cnt_h[i]] for i in range(n)]) if n else np.zeros(0, np.int64)
    return average_precision(scores, tps, int(gcount.sum()))
