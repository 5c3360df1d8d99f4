"""CPU tests of the host-side post-processing + mAP: rotated IoU against
closed-form cases, NMS, AP of hand-built detection sets, anchor map geometry."""
import math

import numpy as np

from v2x_sim_amd.configs import Config
from v2x_sim_amd.utils import postprocess as P


def box(x, y, w, h, yaw):
    return np.array([[x, y, w, h, yaw]], dtype=np.float64)


def test_anchor_map():
    anc = P.build_anchor_map(Config("test"))
    assert anc.shape == (256, 256, 6, 6)
    assert np.allclose(anc[0, 0, 0, :2], [-31.875, -31.875]) and np.allclose(anc[255, 255, 0, :2], [31.875, 31.875])
    assert np.allclose(anc[3, 4, 1, 2:], [2.0, 4.0, 1.0, 0.0], atol=1e-6)      # yaw = pi/2
    assert np.allclose(anc[3, 4, 5, 2:4], [3.0, 12.0])


def test_rotated_iou_closed_form():
    a = P.box_corners(box(0, 0, 2, 2, 0))[0]
    assert abs(P.rotated_iou(a, a) - 1.0) < 1e-12
    b = P.box_corners(box(1, 0, 2, 2, 0))[0]                 # half overlap: inter 2, union 6
    assert abs(P.rotated_iou(a, b) - 2.0 / 6.0) < 1e-12
    c = P.box_corners(box(0, 0, 2, 2, math.pi / 4))[0]       # octagon: inter = 8(sqrt2-1), union = 8 - inter
    inter = 8 * (math.sqrt(2) - 1)
    assert abs(P.rotated_iou(a, c) - inter / (8 - inter)) < 1e-9
    d = P.box_corners(box(5, 5, 2, 2, 0.3))[0]
    assert P.rotated_iou(a, d) == 0.0
    e = P.box_corners(box(0, 0, 1, 1, 1.0))[0]               # contained: 1/4
    assert abs(P.rotated_iou(a, e) - 0.25) < 1e-9
    # symmetric, rotation invariant
    f, g = P.box_corners(box(0.3, -0.2, 4, 2, 0.5))[0], P.box_corners(box(0.8, 0.1, 3, 2.5, -0.4))[0]
    assert abs(P.rotated_iou(f, g) - P.rotated_iou(g, f)) < 1e-12


def test_decode_and_nms():
    anchors = np.array([[1.0, 2.0, 2.0, 4.0, 0.0, 1.0]], dtype=np.float32)
    loc = np.array([[0.5, -0.5, math.log(2.0), 0.0, 1.0, 0.0]], dtype=np.float32)   # +90 deg
    b = P.decode_boxes(loc, anchors)[0]
    assert np.allclose(b, [1.5, 1.5, 4.0, 4.0, math.pi / 2], atol=1e-6)
    boxes = np.array([[0, 0, 2, 2], [0.1, 0.1, 2.1, 2.1], [5, 5, 6, 6]], dtype=np.float64)
    keep = P.nms_standup(boxes, np.array([0.9, 0.8, 0.7]), 0.01)
    assert keep.tolist() == [0, 2]
    assert P.nms_standup(boxes, np.array([0.8, 0.9, 0.7]), 0.01).tolist() == [1, 2]


def test_softmax_and_apply_nms_det():
    cfg = Config("test")
    anc = P.build_anchor_map(cfg)[:4, :4]
    cls = np.zeros((4 * 4 * 6, 2), np.float32)
    cls[:, 0] = 2.0
    cls[10, 1], cls[50, 1] = 6.0, 5.0                       # two confident anchors
    loc = np.zeros((4, 4, 6, 1, 6), np.float32)
    loc[..., 5] = 1.0                                       # zero heading residual
    out = P.apply_nms_det(loc, cls, anc, score_thr=0.7, nms_thr=0.01)
    assert out["scores"].shape[0] in (1, 2) and out["scores"][0] > 0.98
    empty = P.apply_nms_det(loc, np.zeros_like(cls), anc)
    assert empty["boxes"].shape == (0, 5)


def _det(boxes, scores):
    b = np.asarray(boxes, dtype=np.float64)
    return {"corners": P.box_corners(b), "scores": np.asarray(scores, dtype=np.float64)}


def test_eval_map_hand_cases():
    gt = P.box_corners(np.array([[0, 0, 2, 4, 0.0], [10, 0, 2, 4, 0.5], [0, 10, 2, 4, 1.0]]))
    perfect = _det([[0, 0, 2, 4, 0.0], [10, 0, 2, 4, 0.5], [0, 10, 2, 4, 1.0]], [0.9, 0.8, 0.7])
    ap, info = P.eval_map([perfect], [gt], 0.5)
    assert abs(ap - 1.0) < 1e-12 and info["num_gt"] == 3
    # one false positive ranked first, then the three hits: precision envelope 3/4 over recall 0..1
    fp_first = _det([[50, 50, 2, 4, 0], [0, 0, 2, 4, 0.0], [10, 0, 2, 4, 0.5], [0, 10, 2, 4, 1.0]], [0.95, 0.9, 0.8, 0.7])
    ap, _ = P.eval_map([fp_first], [gt], 0.5)
    assert abs(ap - 0.75) < 1e-12
    # a duplicate detection of the same GT counts as a false positive; one GT is missed
    dup = _det([[0, 0, 2, 4, 0.0], [0.05, 0, 2, 4, 0.0], [10, 0, 2, 4, 0.5]], [0.9, 0.8, 0.7])
    ap, _ = P.eval_map([dup], [gt], 0.5)
    assert abs(ap - (1 / 3 * 1.0 + 1 / 3 * (2 / 3))) < 1e-12
    # IoU threshold matters: a shifted box passes 0.5 but not 0.7
    shifted = _det([[0.3, 0, 2, 4, 0.0]], [0.9])
    assert P.eval_map([shifted], [gt[:1]], 0.5)[0] == 1.0 and P.eval_map([shifted], [gt[:1]], 0.8)[0] == 0.0
    assert P.eval_map([_det(np.zeros((0, 5)), [])], [gt], 0.5)[0] == 0.0


def test_postprocess_golden_known_answer():
    """tests/golden/postprocess_small.npz pins the host spec (softmax score, 0.7 threshold, 'faf' decode, stand-up NMS)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_small.npz"))
    det = P.apply_nms_det(g["loc"], g["cls"], g["anchors"], 0.7, 0.01)
    assert det["boxes"].shape == g["boxes"].shape == (7, 5)
    assert np.allclose(det["boxes"], g["boxes"], atol=1e-6) and np.allclose(det["scores"], g["scores"], atol=1e-7)
    assert np.allclose(det["corners"], g["corners"], atol=1e-6)
