"""Row f-1 checker hygiene (VERDICT r1 'what's weak' #5): the product's host post-processing
(v2x_sim_amd/utils/postprocess.py -- also FaFModule's overflow fallback and the scorer of every mAP test) is held to
the INDEPENDENT restatement oracle/postprocess_ref.py (scalar float64, vertex-collection polygon intersection,
from-the-definition AP) and to the fixture generated from it; rotated IoU is refereed by a third, rasterising estimator."""
import math
import os

import numpy as np

from oracle import postprocess_ref as PR
from v2x_sim_amd.utils import postprocess as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rand_boxes(rng, n, spread=3.0):
    return np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(0.5, 5.0, n),
                     rng.uniform(0.5, 5.0, n), rng.uniform(-math.pi, math.pi, n)], 1)


def test_rotated_iou_three_algorithms_agree():
    rng = np.random.default_rng(0)
    a, b = _rand_boxes(rng, 1000), _rand_boxes(rng, 1000)
    ca, cb = P.box_corners(a), P.box_corners(b)
    n_overlap = 0
    for i in range(1000):
        clip = P.rotated_iou(ca[i], cb[i])                       # product: Sutherland-Hodgman
        coll = PR.rotated_iou(PR.corners_of(a[i]), PR.corners_of(b[i]))   # oracle: vertex collection, own corner function
        assert abs(clip - coll) < 1e-9, (i, clip, coll)
        n_overlap += clip > 0
        if i < 120:                                              # referee: point counting on a 1000 x 1000 grid
            assert abs(PR.raster_iou(tuple(a[i]), tuple(b[i]), n=1000) - coll) < 2e-3, i
    assert n_overlap > 400                                       # the sample is not dominated by disjoint pairs
    # degenerate / special poses: identical, contained, touching at an edge, shared corner, 90-degree multiples
    base = (0.0, 0.0, 2.0, 4.0, 0.3)
    for other, want in (((0.0, 0.0, 2.0, 4.0, 0.3), 1.0), ((0.0, 0.0, 1.0, 2.0, 0.3), 0.25),
                        ((0.0, 0.0, 2.0, 4.0, 0.3 + math.pi), 1.0), ((0.0, 0.0, 4.0, 2.0, 0.3 + math.pi / 2), 1.0)):
        assert abs(PR.rotated_iou(PR.corners_of(base), PR.corners_of(other)) - want) < 1e-9
        assert abs(P.rotated_iou(P.box_corners(np.array([base]))[0], P.box_corners(np.array([other]))[0]) - want) < 1e-9
    touch = (2.0 * math.cos(0.3), 2.0 * math.sin(0.3), 2.0, 4.0, 0.3)          # shares one edge: zero area
    assert PR.rotated_iou(PR.corners_of(base), PR.corners_of(touch)) < 1e-9
    assert P.rotated_iou(P.box_corners(np.array([base]))[0], P.box_corners(np.array([touch]))[0]) < 1e-9


def test_rotated_iou_closed_form_known_answers():
    """The referee itself anchored to arithmetic (tests/iou_kats.py): identical boxes, concentric squares at 30 / 45 / 60 degrees (octagon
    formula), half-side shifts, a cross, containment, edge / corner contact, disjoint -- oracle, product and raster estimator."""
    from iou_kats import IOU_KATS
    for a, b, want, what in IOU_KATS:
        a, b = tuple(float(v) for v in a), tuple(float(v) for v in b)
        assert abs(PR.rotated_iou(PR.corners_of(a), PR.corners_of(b)) - want) < 1e-9, what
        assert abs(P.rotated_iou(P.box_corners(np.array([a]))[0], P.box_corners(np.array([b]))[0]) - want) < 1e-9, what
        assert abs(PR.raster_iou(a, b, n=1200) - want) < 3e-3, what


def _blob_logits(rng, X, Y, A, n_obj):
    cls = np.zeros((X, Y, A, 2), np.float32)
    cls[..., 0] = 2.0 + rng.normal(0, 0.3, (X, Y, A))
    cls[..., 1] = -2.0 + rng.normal(0, 0.3, (X, Y, A))
    loc = rng.normal(0, 0.08, (X, Y, A, 1, 6)).astype(np.float32)
    loc[..., 5] += 1.0
    for _ in range(n_obj):
        x, y, a = rng.integers(3, X - 3), rng.integers(3, Y - 3), rng.integers(0, A)
        for dx in range(-2, 3):
            for dy in range(-2, 3):
                s = 4.0 - 0.9 * (abs(dx) + abs(dy)) + rng.normal(0, 0.05)
                cls[x + dx, y + dy, a] = (-s, s)
    return cls.reshape(-1, 2), loc


def _anchors(X, Y, cell=0.25):
    anc = np.zeros((X, Y, 6, 6), np.float32)
    anc[..., 0] = (-X * cell / 2 + (np.arange(X) + 0.5) * cell)[:, None, None]
    anc[..., 1] = (-Y * cell / 2 + (np.arange(Y) + 0.5) * cell)[None, :, None]
    for a, (w, h, yaw) in enumerate(PR.ANCHOR_SIZE):
        anc[:, :, a, 2], anc[:, :, a, 3] = w * cell, h * cell
        anc[:, :, a, 4], anc[:, :, a, 5] = math.sin(yaw), math.cos(yaw)
    return anc


def test_product_decode_and_nms_equal_the_oracle():
    for seed, nms_thr in ((1, 0.01), (2, 0.01), (3, 0.3), (4, 0.5)):
        rng = np.random.default_rng(seed)
        X = Y = 40
        cls, loc = _blob_logits(rng, X, Y, 6, 25)
        loc.reshape(-1, 6)[::97, 2] = 9.0                         # exercises the dw clip (build-owned, ASSUMPTIONS.md)
        anc = _anchors(X, Y)
        got = P.apply_nms_det(loc, cls, anc, 0.7, nms_thr)
        ref = PR.detect(cls, loc.reshape(-1, 6), anc.reshape(-1, 6), 0.7, nms_thr)
        assert got["scores"].shape[0] == len(ref) > 10
        assert np.allclose(got["scores"], [d["score"] for d in ref], atol=1e-6)
        rb = np.asarray([d["box"] for d in ref])
        assert np.allclose(got["boxes"][:, :4], rb[:, :4], atol=1e-4, rtol=1e-5)
        assert np.abs(np.angle(np.exp(1j * (got["boxes"][:, 4] - rb[:, 4])))).max() < 1e-5
        assert np.allclose(got["corners"], np.asarray([d["corners"] for d in ref]), atol=2e-4)


def test_product_matches_the_oracle_generated_fixture():
    g = np.load(os.path.join(GOLD, "postprocess_small.npz"))
    got = P.apply_nms_det(g["loc"], g["cls"], g["anchors"], 0.7, 0.01)
    assert got["scores"].shape[0] == g["scores"].shape[0] == g["index"].shape[0]
    assert np.allclose(got["scores"], g["scores"], atol=1e-6) and np.allclose(got["boxes"], g["boxes"], atol=1e-4)
    fg = P.softmax_fg(g["cls"])
    assert np.allclose(fg[g["index"]], g["scores"], atol=1e-6)


def test_product_eval_map_equals_the_oracle():
    """Random detection sets around jittered ground truth incl. duplicates on one GT, empty images, images without GT."""
    rng = np.random.default_rng(5)
    for trial in range(6):
        dets_p, dets_o, gts_p, gts_o = [], [], [], []
        for img in range(12):
            ng = int(rng.integers(0, 7))
            gt = _rand_boxes(rng, ng, spread=20.0)
            picks = [gt[i] for i in range(ng) for _ in range(int(rng.integers(0, 3)))] + list(_rand_boxes(rng, int(rng.integers(0, 4)), 20.0))
            d = np.asarray(picks).reshape(-1, 5).copy()
            d[:, :2] += rng.normal(0, 0.4, (d.shape[0], 2))
            d[:, 2:4] *= rng.uniform(0.8, 1.25, (d.shape[0], 2))
            d[:, 4] += rng.normal(0, 0.15, d.shape[0])
            sc = np.round(rng.uniform(0.7, 1.0, d.shape[0]), 2)    # rounded: tied scores occur
            dets_p.append({"corners": P.box_corners(d) if d.shape[0] else np.zeros((0, 4, 2)), "scores": sc})
            dets_o.append([(float(sc[j]), PR.corners_of(tuple(d[j]))) for j in range(d.shape[0])])
            gts_p.append(P.box_corners(gt) if ng else np.zeros((0, 4, 2)))
            gts_o.append([PR.corners_of(tuple(b)) for b in gt])
        for thr in (0.5, 0.7):
            ap_p, info = P.eval_map(dets_p, gts_p, thr)
            ap_o, ngt, ndet = PR.eval_map(dets_o, gts_o, thr)
            assert abs(ap_p - ap_o) < 1e-12 and info["num_gt"] == ngt and info["num_det"] == ndet, (trial, thr, ap_p, ap_o)
    assert PR.eval_map([[]], [[PR.corners_of((0, 0, 1, 1, 0))]], 0.5)[0] == 0.0
    perfect = [[(0.9, PR.corners_of((0, 0, 2, 4, 0.1)))]]
    assert abs(PR.eval_map(perfect, [[PR.corners_of((0, 0, 2, 4, 0.1))]], 0.7)[0] - 1.0) < 1e-12


def test_detect_prefilter_equals_full_scan():
    """Full-size maps go through a vectorised pre-selection of the anchors near or above the threshold (the exact scalar test still
    decides each): identical detections to the plain scan, including anchors sitting exactly on the threshold."""
    import math
    rng = np.random.default_rng(5)
    M = 6000
    cls = rng.normal(0, 1, (M, 2))
    cls[:, 0] += 2.0                                   # most anchors are background
    thr = 0.7
    cls[17] = (0.0, math.log(thr / (1 - thr)))         # on the threshold
    cls[18] = (0.0, math.log(thr / (1 - thr)) - 1e-9)  # a hair below
    loc = rng.normal(0, 0.2, (M, 6))
    anchors = np.concatenate([rng.uniform(-30, 30, (M, 2)), np.tile([[2.0, 4.0, 0.0, 1.0]], (M, 1))], 1)
    fast = PR.detect(cls, loc, anchors, thr, 0.01)
    slow = PR.detect(cls.tolist(), loc.tolist(), anchors.tolist(), thr, 0.01)
    assert [d["index"] for d in fast] == [d["index"] for d in slow] and len(fast) > 20
    assert all(a["score"] == b["score"] and a["box"] == b["box"] for a, b in zip(fast, slow))
