"""TEST INFRASTRUCTURE ONLY -- independent CPU restatement of the detection post-processing and the
mAP metric.  PARITY UNPINNED: the upstream files it restates,
  coperception/utils/postprocess.py::apply_nms_det   (score, threshold, 'faf' decode, stand-up NMS)
  coperception/utils/mean_ap.py::eval_map            (mmdet-style single-class AP, rotated IoU via shapely)
are not in /root/reference (README.md:36 names the detection task, README.md:101 the scripts that call
them); every recollected detail is listed in oracle/ASSUMPTIONS.md.

Why a second implementation: round 1 checked the HIP kernels against `v2x_sim_amd/utils/postprocess.py`,
which is PRODUCT code (it is also the > cap fallback of FaFModule.predict_all) and scored both sides of the
mAP tests, so a bug in it cancelled.  This file shares no code and no algorithm with it:
  * scalar float64 python loops instead of vectorised fp32 numpy;
  * rotated IoU by VERTEX COLLECTION (corners of one box inside the other + edge/edge crossings, ordered by
    angle around their centroid, shoelace area) instead of Sutherland-Hodgman clipping, plus a third, dumb
    estimator (`raster_iou`: point-in-box counting on a grid) that the tests use to referee both;
  * AP by explicit per-image greedy matching and a trapezoid-free "area under the monotone envelope" sum
    written from the definition.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math

# anchor table of upstream Config (recollected): (w, h, yaw) in metres / radians
ANCHOR_SIZE = ((2.0, 4.0, 0.0), (2.0, 4.0, math.pi / 2.0), (2.0, 4.0, -math.pi / 4.0),
               (3.0, 12.0, 0.0), (3.0, 12.0, math.pi / 2.0), (3.0, 12.0, -math.pi / 4.0))
SCORE_THR = 0.7     # test-time foreground threshold
NMS_THR = 0.01      # IoU threshold on the axis-aligned stand-up boxes
DECODE_CLIP = 4.0   # BUILD-OWNED (not recalled upstream): dw, dh clipped to +-4 before exp so that garbage logits cannot overflow


# ------------------------------------------------------------------ score / decode / NMS
def fg_score(c0, c1):
    """softmax((c0, c1))[1] for one anchor."""
    m = max(c0, c1)
    e0, e1 = math.exp(c0 - m), math.exp(c1 - m)
    return e1 / (e0 + e1)


def decode_faf(code, anchor):
    """'faf' box code (dx, dy, dw, dh, dsin, dcos) on anchor (xa, ya, wa, ha, sina, cosa) -> (x, y, w, h, yaw)."""
    dx, dy, dw, dh, ds, dc = (float(v) for v in code)
    xa, ya, wa, ha, sa, ca = (float(v) for v in anchor)
    dw = min(max(dw, -DECODE_CLIP), DECODE_CLIP)
    dh = min(max(dh, -DECODE_CLIP), DECODE_CLIP)
    return (xa + dx, ya + dy, wa * math.exp(dw), ha * math.exp(dh), math.atan2(sa, ca) + math.atan2(ds, dc))


WH_AXES = ("w_along_heading", "h_along_heading")


def corners_of(box, wh_axis="w_along_heading"):
    """(x, y, w, h, yaw) -> 4 corners, counter-clockwise, first = (+w/2, +h/2) rotated.  wh_axis: which extent runs along the box's local x
    axis (its heading) before rotation -- ASSUMPTIONS.md row 48, both readings: "w_along_heading" (default) or "h_along_heading" (the
    rectangle with the two extents exchanged)."""
    x, y, w, h, yaw = box
    if wh_axis == "h_along_heading":
        w, h = h, w
    elif wh_axis != "w_along_heading":
        raise ValueError("wh_axis must be one of %s" % (WH_AXES,))
    c, s = math.cos(yaw), math.sin(yaw)
    out = []
    for sx, sy in ((0.5, 0.5), (-0.5, 0.5), (-0.5, -0.5), (0.5, -0.5)):
        lx, ly = sx * w, sy * h
        out.append((x + lx * c - ly * s, y + lx * s + ly * c))
    return out


def standup_of(corners):
    xs, ys = [p[0] for p in corners], [p[1] for p in corners]
    return (min(xs), min(ys), max(xs), max(ys))


def aabb_iou(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0])
    ih = min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0.0 or ih <= 0.0:
        return 0.0
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def detect(cls, loc, anchors, score_thr=SCORE_THR, nms_thr=NMS_THR, rotated=False, wh_axis="w_along_heading"):
    """One map.  cls [M][2] logits, loc [M][6] codes, anchors [M][6] (any nested sequence / array).
    -> list of dict(index, score, box (x, y, w, h, yaw), corners) in the order the detections are kept:
    score descending, ties by anchor index ascending; a candidate is dropped iff its stand-up box overlaps an
    already KEPT one with IoU > nms_thr.  rotated=True (not upstream's default): the IoU of the rotated boxes themselves."""
    cand = []
    idx = range(len(cls))
    if hasattr(cls, "shape") and len(cls) > 4096:
        # full-size maps (393 216 anchors): visit only the anchors whose logit margin c1 - c0 is within 1e-3 of the threshold's or above
        # (score >= thr  <=>  c1 - c0 >= log(thr / (1 - thr))); the exact scalar test below still decides every one of them
        import numpy as np
        c = np.asarray(cls, dtype=np.float64)
        idx = np.nonzero(c[:, 1] - c[:, 0] >= math.log(score_thr / (1.0 - score_thr)) - 1e-3)[0].tolist()
    for i in idx:
        s = fg_score(float(cls[i][0]), float(cls[i][1]))
        if s >= score_thr:
            cand.append((-s, i))
    cand.sort()
    kept = []
    for neg_s, i in cand:
        box = decode_faf(loc[i], anchors[i])
        cor = corners_of(box, wh_axis)
        sb = standup_of(cor)
        if rotated:
            if any(aabb_iou(sb, k["standup"]) > 0.0 and rotated_iou(cor, k["corners"]) > nms_thr for k in kept):
                continue
        elif any(aabb_iou(sb, k["standup"]) > nms_thr for k in kept):
            continue
        kept.append({"index": i, "score": -neg_s, "box": box, "corners": cor, "standup": sb})
    return kept


# ------------------------------------------------------------------ rotated IoU, two independent ways
def _inside_convex(p, poly, eps=1e-12):
    """p inside or on the counter-clockwise convex polygon."""
    n = len(poly)
    for i in range(n):
        ax, ay = poly[i]
        bx, by = poly[(i + 1) % n]
        if (bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax) < -eps:
            return False
    return True


def _segment_crossing(p1, p2, q1, q2):
    """Proper or touching crossing point of segments p1p2 and q1q2, None when parallel or disjoint."""
    rx, ry = p2[0] - p1[0], p2[1] - p1[1]
    sx, sy = q2[0] - q1[0], q2[1] - q1[1]
    den = rx * sy - ry * sx
    if abs(den) < 1e-14:
        return None
    t = ((q1[0] - p1[0]) * sy - (q1[1] - p1[1]) * sx) / den
    u = ((q1[0] - p1[0]) * ry - (q1[1] - p1[1]) * rx) / den
    if -1e-12 <= t <= 1.0 + 1e-12 and -1e-12 <= u <= 1.0 + 1e-12:
        return (p1[0] + t * rx, p1[1] + t * ry)
    return None


def _shoelace(poly):
    a = 0.0
    for i in range(len(poly)):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % len(poly)]
        a += x1 * y2 - x2 * y1
    return 0.5 * abs(a)


def intersection_area(c1, c2):
    """Area of the intersection of two convex quadrilaterals by vertex collection (no clipping)."""
    c1 = [(float(p[0]), float(p[1])) for p in c1]
    c2 = [(float(p[0]), float(p[1])) for p in c2]
    pts = [p for p in c1 if _inside_convex(p, c2)] + [p for p in c2 if _inside_convex(p, c1)]
    for i in range(4):
        for j in range(4):
            x = _segment_crossing(c1[i], c1[(i + 1) % 4], c2[j], c2[(j + 1) % 4])
            if x is not None:
                pts.append(x)
    if len(pts) < 3:
        return 0.0
    cx = sum(p[0] for p in pts) / len(pts)
    cy = sum(p[1] for p in pts) / len(pts)
    pts.sort(key=lambda p: math.atan2(p[1] - cy, p[0] - cx))
    uniq = [pts[0]]
    for p in pts[1:]:
        if abs(p[0] - uniq[-1][0]) > 1e-10 or abs(p[1] - uniq[-1][1]) > 1e-10:
            uniq.append(p)
    return _shoelace(uniq) if len(uniq) >= 3 else 0.0


def rotated_iou(c1, c2):
    inter = intersection_area(c1, c2)
    union = _shoelace([tuple(p) for p in c1]) + _shoelace([tuple(p) for p in c2]) - inter
    return inter / union if union > 0.0 else 0.0


def raster_iou(b1, b2, n=400):
    """Referee: IoU of two (x, y, w, h, yaw) boxes by counting cell centres of an n x n grid over their joint bounding
    square (numpy; error ~ perimeter / n, a few 1e-3 at n = 400)."""
    import numpy as np
    cs = [corners_of(b1), corners_of(b2)]
    xs = [p[0] for c in cs for p in c]
    ys = [p[1] for c in cs for p in c]
    gx = np.linspace(min(xs), max(xs), n, endpoint=False) + (max(xs) - min(xs)) / (2 * n)
    gy = np.linspace(min(ys), max(ys), n, endpoint=False) + (max(ys) - min(ys)) / (2 * n)
    X, Y = np.meshgrid(gx, gy, indexing="ij")
    masks = []
    for (x, y, w, h, yaw) in (b1, b2):
        c, s = math.cos(yaw), math.sin(yaw)
        lx = (X - x) * c + (Y - y) * s
        ly = -(X - x) * s + (Y - y) * c
        masks.append((np.abs(lx) <= w / 2) & (np.abs(ly) <= h / 2))
    inter = float((masks[0] & masks[1]).sum())
    union = float((masks[0] | masks[1]).sum())
    return inter / union if union else 0.0


# ------------------------------------------------------------------ AP
def eval_map(det_results, annotations, iou_thr=0.5):
    """Single-class AP, mmdet 'area' mode as upstream's eval_map uses it.
    det_results: per image a list of (score, corners); annotations: per image a list of ground-truth corners.
    Per image the detections are visited by descending score; each takes the ground truth of HIGHEST IoU and is a true
    positive iff that IoU >= iou_thr and that ground truth is still free (a better-overlapping but taken box makes it a
    false positive -- there is no second choice).  AP = sum over recall steps of (recall increment) x (best precision at
    this or any higher recall).  -> (ap, num_gt, num_det)"""
    flagged = []     # (score, image, order, is_tp)
    num_gt = 0
    for img, (dets, gts) in enumerate(zip(det_results, annotations)):
        num_gt += len(gts)
        free = [True] * len(gts)
        order = sorted(range(len(dets)), key=lambda j: (-dets[j][0], j))
        for j in order:
            best, arg = 0.0, -1
            for g, gc in enumerate(gts):
                v = rotated_iou(dets[j][1], gc)
                if v > best:
                    best, arg = v, g
            tp = arg >= 0 and best >= iou_thr and free[arg]
            if tp:
                free[arg] = False
            flagged.append((dets[j][0], img, len(flagged), tp))
    if num_gt == 0 or not flagged:
        return 0.0, num_gt, len(flagged)
    flagged.sort(key=lambda r: (-r[0], r[2]))
    precision, recall = [], []
    tp = fp = 0
    for _, _, _, is_tp in flagged:
        tp += 1 if is_tp else 0
        fp += 0 if is_tp else 1
        precision.append(tp / (tp + fp))
        recall.append(tp / num_gt)
    ap, prev_r = 0.0, 0.0
    for k in range(len(recall)):
        if recall[k] > prev_r:
            ap += (recall[k] - prev_r) * max(precision[k:])
            prev_r = recall[k]
    return ap, num_gt, len(flagged)
