"""Row f-1 (post-processing and metric): python wrappers over v2x_det_postprocess[_rotated], the fused candidate heads + v2x_det_nms_candidates,
v2x_rotated_iou and v2x_match_detections.  Re-exported by ops.py (`ops.det_postprocess` ...)."""
import ctypes as C

import torch

from . import _launch, _lib
from ._launch import _Prof, _dev, _stream  # noqa: F401
from ._lib import ConvDesc, V2X_EPI_DET


def det_postprocess(cls, loc, anchors, score_thr=0.7, nms_thr=0.01, cap=4096, rotated=False):
    """cls (n, M, 2) fp32, loc (n, ..., 6) fp32 with M anchors per map, anchors (M, 6) fp32 on the device ->
    (boxes (n, cap, 5), scores (n, cap), index (n, cap) int32, count (n,) int32); count < 0: more than `cap` candidates.
    rotated=True: suppression on the rotated boxes' polygon IoU instead of upstream's stand-up boxes."""
    lib = _lib.load()
    n, M = cls.shape[0], cls.shape[1]
    loc = loc.reshape(n, M, 6)
    anchors = anchors.reshape(M, 6)
    dev = cls.device
    boxes = torch.empty((n, cap, 5), dtype=torch.float32, device=dev)
    scores = torch.empty((n, cap), dtype=torch.float32, device=dev)
    index = torch.empty((n, cap), dtype=torch.int32, device=dev)
    count = torch.empty((n,), dtype=torch.int32, device=dev)
    keys = torch.empty((n, cap), dtype=torch.int64, device=dev)
    cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    fn = lib.v2x_det_postprocess_rotated if rotated else lib.v2x_det_postprocess
    _lib.check(fn(_dev(cls, torch.float32, "cls"), _dev(loc, torch.float32, "loc"),
                                       _dev(anchors, torch.float32, "anchors"), n, M, C.c_float(score_thr),
                                       C.c_float(nms_thr), cap, _dev(boxes, torch.float32, "boxes"),
                                       _dev(scores, torch.float32, "scores"), _dev(index, torch.int32, "index"),
                                       _dev(count, torch.int32, "count"), _dev(keys, torch.int64, "keys"),
                                       _dev(cnt, torch.int32, "cnt"), _stream()), "v2x_det_postprocess")
    return boxes, scores, index, count


def conv2d_det(pc, x, score_thr, cap=4096):
    """The fused detection heads (packing.pack_heads_det; conv_halo.hip V2X_EPI_DET) on the decoder's output x (N, H, W, 32) bf16:
    softmax(cls)[1] >= score_thr is evaluated in the epilogue, only the candidates leave the kernel.
    -> keys (N, cap) int64, codes (N, cap, 6) fp32, counts (N,) int32 (the true count even when > cap) for det_nms_candidates."""
    lib = _lib.load()
    N, H, W, Cx = x.shape
    if Cx != pc.C0 or pc.epilogue != _lib.V2X_EPI_DET:
        raise ValueError("conv2d_det needs the det-heads packing and a %d-channel input" % pc.C0)
    keys = torch.empty((N, cap), dtype=torch.int64, device=x.device)
    codes = torch.empty((N, cap, 6), dtype=torch.float32, device=x.device)
    counts = torch.zeros((N,), dtype=torch.int32, device=x.device)
    d = ConvDesc()
    d.in0, d.in1 = _dev(x, torch.bfloat16, "x").value, None
    d.C0, d.C1, d.up0 = pc.C0, 0, 0
    d.N, d.H, d.W = N, H, W
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.Cout, d.w_rows, d.w_kpad = pc.Cout, pc.w_rows, pc.w_kpad
    d.weight, d.scale, d.shift = pc.weight.data_ptr(), pc.scale.data_ptr(), pc.shift.data_ptr()
    d.epilogue, d.relu = _lib.V2X_EPI_DET, int(bool(pc.relu))
    d.out, d.out_cstride, d.out_coff = keys.data_ptr(), cap, 0
    d.out2, d.split, d.out2_cstride = codes.data_ptr(), 0, 6
    d.w_layout = 1
    d.Cout2, d.relu2 = pc.Cout2, 0
    d.weight2, d.scale2, d.shift2 = pc.weight2.data_ptr(), pc.scale2.data_ptr(), pc.shift2.data_ptr()
    d.det_counts, d.det_thr, d.det_cap = counts.data_ptr(), float(score_thr), cap
    prof = None
    if _launch.PROFILE is not None:
        M = N * H * W
        prof = _Prof("conv3x3_halo_kernel<0, 32, 64, 64, 3>", 2.0 * M * (64 * 9 * 32 + 48 * 64), x.numel() * 2 + pc.weight.numel() * 2, pc.name)
    rc = lib.v2x_conv2d(C.byref(d), _stream())
    if prof is not None:
        prof.done()
    _lib.check(rc, "v2x_conv2d(%s, det)" % pc.name)
    return keys, codes, counts


def det_nms_candidates(keys, codes, counts, anchors, nms_thr=0.01, rotated=False):
    """Second half of det_postprocess for the candidates conv2d_det selected: sort, 'faf' decode, greedy NMS.
    -> (boxes (n, cap, 5), scores (n, cap), index (n, cap) int32, count (n,) int32), identical to det_postprocess on the logits."""
    lib = _lib.load()
    n, cap = keys.shape
    anchors = anchors.reshape(-1, 6)
    M = anchors.shape[0]
    dev = keys.device
    boxes = torch.empty((n, cap, 5), dtype=torch.float32, device=dev)
    scores = torch.empty((n, cap), dtype=torch.float32, device=dev)
    index = torch.empty((n, cap), dtype=torch.int32, device=dev)
    count = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.check(lib.v2x_det_nms_candidates(_dev(keys, torch.int64, "keys"), _dev(codes, torch.float32, "codes"), _dev(counts, torch.int32, "counts"),
                                          _dev(anchors, torch.float32, "anchors"), n, M, cap, C.c_float(nms_thr), int(bool(rotated)),
                                          _dev(boxes, torch.float32, "boxes"), _dev(scores, torch.float32, "scores"),
                                          _dev(index, torch.int32, "index"), _dev(count, torch.int32, "count"), _stream()),
               "v2x_det_nms_candidates")
    return boxes, scores, index, count


def rotated_iou(boxes_a, boxes_b):
    """boxes (na, 5), (nb, 5) fp32 (x, y, w, h, yaw) on the device -> (na, nb) fp32 IoU of the rotated rectangles."""
    lib = _lib.load()
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    out = torch.zeros((na, nb), dtype=torch.float32, device=boxes_a.device)
    if na and nb:
        _lib.check(lib.v2x_rotated_iou(_dev(boxes_a, torch.float32, "boxes_a"), na, _dev(boxes_b, torch.float32, "boxes_b"), nb,
                                       _dev(out, torch.float32, "iou"), _stream()), "v2x_rotated_iou")
    return out


def match_detections(det_boxes, det_count, gt_boxes, gt_count, iou_thr, want_iou=False):
    """eval_map's matching on the device.  det_boxes (n, det_cap, 5) fp32 in descending-score order, det_count (n,) int32,
    gt_boxes (n, gt_cap, 5) fp32, gt_count (n,) int32 -> tp (n, det_cap) int32 [, best_iou (n, det_cap) fp32]."""
    lib = _lib.load()
    n, det_cap, _ = det_boxes.shape
    gt_cap = gt_boxes.shape[1]
    tp = torch.zeros((n, det_cap), dtype=torch.int32, device=det_boxes.device)
    best = torch.zeros((n, det_cap), dtype=torch.float32, device=det_boxes.device) if want_iou else None
    _lib.check(lib.v2x_match_detections(_dev(det_boxes, torch.float32, "det_boxes"), _dev(det_count, torch.int32, "det_count"), det_cap,
                                        _dev(gt_boxes, torch.float32, "gt_boxes"), _dev(gt_count, torch.int32, "gt_count"), gt_cap, n,
                                        C.c_float(iou_thr), _dev(tp, torch.int32, "tp"),
                                        _dev(best, torch.float32, "best_iou") if best is not None else None, _stream()),
               "v2x_match_detections")
    return (tp, best) if want_iou else tp
