"""Row f-1 on the device: v2x_det_postprocess (score, threshold, 'faf' decode, stand-up NMS) against the ORACLE
(oracle/postprocess_ref.py: scalar float64 restatement of upstream's apply_nms_det, independent of the product's
utils/postprocess.py -- which is itself held to the same oracle in tests/test_postprocess_ref_cpu.py).

The kernel computes in fp32, the oracle in float64: same number of detections, the SAME anchors kept in the same order,
boxes within 1e-4 m / 1e-5 rad, scores within 1e-6.  Candidate overflow (> cap) is reported as a negative count."""
import numpy as np
import pytest
import torch

from oracle import postprocess_ref as PR

pytestmark = pytest.mark.gpu


def _oracle_dets(cls, loc, anchors, score_thr, nms_thr):
    det = PR.detect(cls, np.asarray(loc).reshape(-1, 6), np.asarray(anchors).reshape(-1, 6), score_thr, nms_thr)
    return {"boxes": np.asarray([d["box"] for d in det], np.float64).reshape(-1, 5),
            "scores": np.asarray([d["score"] for d in det], np.float64),
            "corners": np.asarray([d["corners"] for d in det], np.float64).reshape(-1, 4, 2),
            "index": np.asarray([d["index"] for d in det], np.int64)}


def _synthetic_logits(n, X, Y, A, n_obj, seed):
    """Logits with n_obj confident blobs per map (several neighbouring anchors fire per object -> NMS has work to do)
    over a low-score background."""
    rng = np.random.default_rng(seed)
    cls = np.zeros((n, X, Y, A, 2), np.float32)
    cls[..., 0] = 2.0 + rng.normal(0, 0.3, (n, X, Y, A))
    cls[..., 1] = -2.0 + rng.normal(0, 0.3, (n, X, Y, A))
    loc = rng.normal(0, 0.05, (n, X, Y, A, 1, 6)).astype(np.float32)
    loc[..., 5] += 1.0
    for i in range(n):
        for _ in range(n_obj):
            x, y, a = rng.integers(4, X - 4), rng.integers(4, Y - 4), rng.integers(0, A)
            for dx in range(-2, 3):
                for dy in range(-2, 3):
                    s = 4.0 - 0.8 * (abs(dx) + abs(dy)) + rng.normal(0, 0.05)
                    cls[i, x + dx, y + dy, a] = (-s, s)
    return cls.reshape(n, -1, 2), loc


def _compare(dev_dets, host_dets):
    for d, h in zip(dev_dets, host_dets):
        assert d["scores"].shape == h["scores"].shape, (d["scores"].shape, h["scores"].shape)
        if h["scores"].shape[0] == 0:
            continue
        assert np.allclose(d["scores"], h["scores"], atol=1e-6)
        assert np.allclose(d["boxes"][:, :4], h["boxes"][:, :4], atol=1e-4)
        dyaw = np.angle(np.exp(1j * (d["boxes"][:, 4] - h["boxes"][:, 4])))
        assert np.abs(dyaw).max() < 1e-5
        assert np.allclose(d["corners"], h["corners"], atol=2e-4)


def test_device_postprocess_equals_oracle(device):
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils import postprocess as P
    cfg = Config("test")
    anchors = P.build_anchor_map(cfg)
    X, Y, A = anchors.shape[:3]
    n = 6
    cls, loc = _synthetic_logits(n, X, Y, A, n_obj=40, seed=3)
    cls[5] = 0.0                                       # a map with every score at 0.5 < thr: no detections at all
    boxes, scores, index, count = ops.det_postprocess(torch.from_numpy(cls).to(device), torch.from_numpy(loc).to(device),
                                                      torch.from_numpy(anchors.reshape(-1, 6)).to(device), 0.7, 0.01, 4096)
    count = count.cpu().numpy()
    assert count[5] == 0 and (count[:5] > 10).all()
    dev = []
    for i in range(n):
        b = boxes[i, :count[i]].cpu().numpy()
        dev.append({"boxes": b, "scores": scores[i, :count[i]].cpu().numpy(),
                    "corners": P.box_corners(b) if count[i] else np.zeros((0, 4, 2), np.float32)})
    ref = [_oracle_dets(cls[i], loc[i], anchors, 0.7, 0.01) for i in range(n)]
    for i in range(n):
        # the kept ANCHORS are the comparison key: the kernel ranks by its fp32 score, the oracle by a float64 one, so two
        # candidates whose scores agree to the last fp32 bit may swap places -- allowed only there
        di, ri = index[i, :count[i]].cpu().numpy().astype(np.int64), ref[i]["index"]
        assert np.array_equal(np.sort(di), np.sort(ri)), "map %d: kept anchor sets differ" % i
        pos = {int(a): k for k, a in enumerate(ri)}
        perm = np.array([pos[int(a)] for a in di], dtype=np.int64)
        moved = np.nonzero(perm != np.arange(perm.size))[0]
        for k in moved:
            assert abs(ref[i]["scores"][perm[k]] - ref[i]["scores"][k]) < 2e-7, (i, k, perm[k])
        dev[i] = {key: (v[np.argsort(perm)] if v.shape[0] == perm.size else v) for key, v in dev[i].items()}
    _compare(dev, ref)


def test_candidate_overflow_is_reported(device):
    from v2x_sim_amd import ops
    n, M = 2, 4096
    cls = torch.zeros((n, M, 2))
    cls[0, :, 1] = 5.0                                  # every anchor of map 0 passes: 4096 > cap = 256
    cls[1, :100, 1] = 5.0
    loc = torch.zeros((n, M, 6))
    loc[..., 5] = 1.0
    anchors = torch.zeros((M, 6))
    anchors[:, 0] = torch.arange(M) * 10.0              # far apart: nothing suppressed
    anchors[:, 2:4] = 1.0
    anchors[:, 5] = 1.0
    _, _, _, count = ops.det_postprocess(cls.to(device), loc.to(device), anchors.to(device), 0.7, 0.01, 256)
    assert count.cpu().tolist() == [-4096, 100]


def test_predict_all_device_equals_host_path(device):
    """FaFModule.predict_all with the device post-processing (default) vs upstream's host-side numpy path, on V2VNet output
    whose head biases are shifted so that a realistic number of anchors (hundreds per map) passes the 0.7 threshold."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
    from v2x_sim_amd import ops
    A, B = 5, 1
    cfg = Config("test")
    pm = init_synthetic_weights(V2VNet(cfg), seed=0)
    pm = pm.to(device)
    grid = ops.VoxelGrid()
    pts = torch.from_numpy(synthetic_points(A * B, 20000, seed=41)).to(device)
    bits = ops.voxelize_bits(pts, torch.full((A * B,), 20000, dtype=torch.int32, device=device), grid)
    data = {"bev_seq": ops.bits_to_dense(bits, 13)[:, None], "trans_matrices": torch.from_numpy(synthetic_poses(B, A, seed=42)).to(device),
            "num_agent": torch.full((B, A), A)}
    module = FaFModule(pm, None, cfg, None, 0)
    with torch.no_grad():
        res = pm(data["bev_seq"], data["trans_matrices"], data["num_agent"], batch_size=B)
    from v2x_sim_amd.utils import postprocess as P
    fg = P.softmax_fg(res["cls"].float().cpu().numpy().reshape(-1, 2))
    module.score_thr = float(np.quantile(fg, 0.998))    # ~800 candidates per map
    module.nms_thr = 0.3
    _, _, _, seq_dev = module.predict_all(data, B, validation=False, num_agent=A)
    module.device_postprocess = False
    _, _, _, seq_host = module.predict_all(data, B, validation=False, num_agent=A)
    _compare([seq_dev[k][0] for k in range(A)], [seq_host[k][0] for k in range(A)])
    assert sum(s[0]["scores"].shape[0] for s in seq_host) > 30


def test_device_postprocess_golden(device):
    """Known-answer test: tests/golden/postprocess_small.npz (generated from oracle/postprocess_ref.py by make_golden.py)."""
    import os
    from v2x_sim_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_small.npz"))
    cls = torch.from_numpy(g["cls"])[None].to(device)
    loc = torch.from_numpy(g["loc"])[None].to(device)
    boxes, scores, index, count = ops.det_postprocess(cls, loc, torch.from_numpy(g["anchors"].reshape(-1, 6)).to(device),
                                                      0.7, 0.01, 256)
    n = int(count[0])
    assert n == g["boxes"].shape[0]
    assert np.allclose(boxes[0, :n].cpu().numpy(), g["boxes"], atol=1e-4)
    assert np.allclose(scores[0, :n].cpu().numpy(), g["scores"], atol=1e-6)
    assert np.array_equal(index[0, :n].cpu().numpy(), g["index"])


def _rand_boxes(rng, n, spread=3.0):
    import math
    return np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(0.5, 5.0, n),
                     rng.uniform(0.5, 5.0, n), rng.uniform(-math.pi, math.pi, n)], 1).astype(np.float32)


def test_rotated_iou_kernel_closed_form_known_answers(device):
    """v2x_rotated_iou against the hand-computable answers of tests/iou_kats.py (fp32 boxes in, fp64 clip inside: 1e-6)."""
    from iou_kats import IOU_KATS
    from v2x_sim_amd import ops
    a = torch.tensor([k[0] for k in IOU_KATS], dtype=torch.float32, device=device)
    b = torch.tensor([k[1] for k in IOU_KATS], dtype=torch.float32, device=device)
    got = ops.rotated_iou(a, b).cpu().numpy()
    for i, (_, _, want, what) in enumerate(IOU_KATS):
        assert abs(got[i, i] - want) < 2e-6, (what, got[i, i], want)


def test_rotated_iou_kernel_vs_oracle(device):
    """v2x_rotated_iou (Sutherland-Hodgman in fp64 on the device) against the oracle's vertex-collection IoU on 60 x 50 random
    box pairs + the degenerate poses (identical, contained, edge contact, 90-degree turns)."""
    from v2x_sim_amd import ops
    rng = np.random.default_rng(0)
    a, b = _rand_boxes(rng, 60), _rand_boxes(rng, 50)
    a[:4] = [[0, 0, 2, 4, 0.3]] * 4
    b[:4] = [[0, 0, 2, 4, 0.3], [0, 0, 1, 2, 0.3], [0, 0, 4, 2, 0.3 + np.pi / 2], [2 * np.cos(0.3), 2 * np.sin(0.3), 2, 4, 0.3]]
    got = ops.rotated_iou(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)).cpu().numpy()
    want = np.array([[PR.rotated_iou(PR.corners_of(tuple(float(v) for v in x)), PR.corners_of(tuple(float(v) for v in y))) for y in b] for x in a])
    assert np.abs(got - want).max() < 1e-6, float(np.abs(got - want).max())
    assert abs(got[0, 0] - 1.0) < 1e-6 and abs(got[1, 1] - 0.25) < 1e-6 and abs(got[2, 2] - 1.0) < 1e-6 and got[3, 3] < 1e-6
    assert (got > 0).mean() > 0.3
    assert ops.rotated_iou(torch.zeros((0, 5), device=device), torch.from_numpy(b).to(device)).shape == (0, 50)


def test_match_detections_and_map_on_device_vs_oracle(device):
    """eval_map with IoU + greedy matching on the device (v2x_match_detections) = the oracle's from-the-definition AP, on
    random detection sets around jittered ground truth: duplicates on one GT, tied scores, images without GT / detections."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils import postprocess as P
    rng = np.random.default_rng(7)
    for trial in range(4):
        n_img, det_cap = 16, 64
        det = np.zeros((n_img, det_cap, 5), np.float32)
        sc = np.zeros((n_img, det_cap), np.float32)
        cnt = np.zeros((n_img,), np.int32)
        gts, dets_o, gts_o = [], [], []
        for i in range(n_img):
            ng = int(rng.integers(0, 9))
            gt = _rand_boxes(rng, ng, spread=20.0)
            picks = [gt[g] for g in range(ng) for _ in range(int(rng.integers(0, 3)))] + list(_rand_boxes(rng, int(rng.integers(0, 5)), 20.0))
            d = np.asarray(picks, np.float32).reshape(-1, 5).copy()
            d[:, :2] += rng.normal(0, 0.4, (d.shape[0], 2)).astype(np.float32)
            d[:, 2:4] *= rng.uniform(0.8, 1.25, (d.shape[0], 2)).astype(np.float32)
            d[:, 4] += rng.normal(0, 0.15, d.shape[0]).astype(np.float32)
            s = np.round(rng.uniform(0.7, 1.0, d.shape[0]), 2).astype(np.float32)
            order = np.argsort(-s, kind="stable")                  # the device contract: descending score per image
            d, s = d[order], s[order]
            det[i, :d.shape[0]], sc[i, :d.shape[0]], cnt[i] = d, s, d.shape[0]
            gts.append(gt)
            dets_o.append([(float(s[j]), PR.corners_of(tuple(float(v) for v in d[j]))) for j in range(d.shape[0])])
            gts_o.append([PR.corners_of(tuple(float(v) for v in g)) for g in gt])
        for thr in (0.5, 0.7):
            ap_d, info = P.eval_map_device(torch.from_numpy(det).to(device), torch.from_numpy(sc).to(device),
                                           torch.from_numpy(cnt).to(device), gts, thr)
            ap_o, ngt, ndet = PR.eval_map(dets_o, gts_o, thr)
            assert abs(ap_d - ap_o) < 1e-9 and info["num_gt"] == ngt and info["num_det"] == ndet, (trial, thr, ap_d, ap_o)
    # a negative count is the device post-processing's overflow signal: scoring such a map as "no detections" would be silent -> refused
    bad = torch.from_numpy(cnt).to(device).clone()
    bad[1] = -5
    with pytest.raises(ValueError, match="overflowed"):
        P.eval_map_device(torch.from_numpy(det).to(device), torch.from_numpy(sc).to(device), bad, gts, 0.5)
    # flags themselves on a hand case: two detections on one GT -> the better-scored one wins, the other is a false positive
    det = torch.tensor([[[0, 0, 2, 4, 0.0], [0.1, 0, 2, 4, 0.0], [9, 9, 2, 4, 0.0]]], device=device)
    gt = torch.tensor([[[0, 0, 2, 4, 0.0]]], device=device)
    tp, best = ops.match_detections(det, torch.tensor([3], dtype=torch.int32, device=device), gt,
                                    torch.tensor([1], dtype=torch.int32, device=device), 0.5, want_iou=True)
    assert tp[0].tolist() == [1, 0, 0] and abs(float(best[0, 0]) - 1.0) < 1e-6 and float(best[0, 2]) == 0.0


def test_rotated_nms_on_device_vs_oracle(device):
    """v2x_det_postprocess_rotated: suppression on the rotated boxes' polygon IoU (SURVEY row f-1 'rotated-box NMS').  Same kept
    anchors as the oracle's rotated greedy NMS, and strictly more detections survive than with the stand-up boxes (elongated
    boxes at 45 degrees have large stand-up overlaps but small true overlaps)."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils import postprocess as P
    anchors = P.build_anchor_map(Config("test"))
    X, Y, A = anchors.shape[:3]
    cls, loc = _synthetic_logits(2, X, Y, A, n_obj=60, seed=11)
    args = (torch.from_numpy(cls).to(device), torch.from_numpy(loc).to(device), torch.from_numpy(anchors.reshape(-1, 6)).to(device))
    nms_thr = 0.1
    _, _, index_r, count_r = ops.det_postprocess(*args, 0.7, nms_thr, 4096, rotated=True)
    _, _, _, count_s = ops.det_postprocess(*args, 0.7, nms_thr, 4096, rotated=False)
    for i in range(2):
        ref = PR.detect(cls[i], loc[i].reshape(-1, 6), anchors.reshape(-1, 6), 0.7, nms_thr, rotated=True)
        got = index_r[i, :int(count_r[i])].cpu().numpy()
        assert np.array_equal(np.sort(got), np.sort([d["index"] for d in ref])), i
    assert int(count_r.sum()) > int(count_s.sum())


@pytest.mark.parametrize("rotated", [False, True])
def test_full_size_postprocess_properties(device, rotated):
    """Row f-1 at the bench size (320 maps x 393 216 anchors; the python oracle needs minutes per map there) through properties:
    (a) sortedness -- the kept detections of every map come out by descending score;
    (b) idempotence -- running the post-processing again on ONLY the kept anchors (every other score pushed below the threshold) keeps
        exactly the same anchors in the same order: survivors do not suppress each other;
    (c) the batch is a set -- permuting the maps permutes the results bit for bit."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils import postprocess as P
    cfg = Config("test")
    anchors = P.build_anchor_map(cfg)
    X, Y, A = anchors.shape[:3]
    n_small = 8
    cls_s, loc_s = _synthetic_logits(n_small, X, Y, A, n_obj=60, seed=11)
    n = 320
    rep = np.arange(n) % n_small
    cls = torch.from_numpy(cls_s).to(device)[torch.from_numpy(rep).to(device)].contiguous()
    loc = torch.from_numpy(loc_s).to(device)[torch.from_numpy(rep).to(device)].contiguous()
    # make the 40 copies of a map differ: a per-map offset on the foreground logit (changes scores, keeps the objects)
    off = torch.linspace(0.0, 0.5, n, device=device).view(n, 1)
    cls[:, :, 1] += off
    anc = torch.from_numpy(anchors.reshape(-1, 6)).to(device)
    boxes, scores, index, count = ops.det_postprocess(cls, loc, anc, 0.7, 0.01, 4096, rotated=rotated)
    cnt = count.cpu().numpy()
    assert (cnt > 20).all() and (cnt < 4096).all()
    sc = scores.cpu().numpy()
    for i in range(n):
        assert np.all(np.diff(sc[i, :cnt[i]]) <= 0), "map %d: kept scores not sorted" % i
    # (b) only the survivors stay above the threshold
    keep = torch.zeros((n, cls.shape[1]), dtype=torch.bool, device=device)
    for i in range(n):
        keep[i, index[i, :cnt[i]].long()] = True
    cls2 = cls.clone()
    cls2[:, :, 1] = torch.where(keep, cls[:, :, 1], torch.full_like(cls[:, :, 1], -20.0))
    cls2[:, :, 0] = torch.where(keep, cls[:, :, 0], torch.full_like(cls[:, :, 0], 20.0))
    b2, s2, i2, c2 = ops.det_postprocess(cls2, loc, anc, 0.7, 0.01, 4096, rotated=rotated)
    assert torch.equal(c2, count)
    for i in range(n):
        assert torch.equal(i2[i, :cnt[i]], index[i, :cnt[i]]) and torch.equal(b2[i, :cnt[i]], boxes[i, :cnt[i]]), i
    # (c) permutation of the maps
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(device)
    b3, s3, i3, c3 = ops.det_postprocess(cls[perm].contiguous(), loc[perm].contiguous(), anc, 0.7, 0.01, 4096, rotated=rotated)
    assert torch.equal(c3, count[perm])
    pc = perm.cpu().numpy()
    for k in range(0, n, 7):
        m = cnt[pc[k]]
        assert torch.equal(i3[k, :m], index[pc[k], :m]) and torch.equal(s3[k, :m], scores[pc[k], :m])


# ------------------------------------------------------------------ detections without the logits round trip (V2X_EPI_DET)
def _heads_model(device, seed=3, fg_bias=-2.5):
    """FaFNet with synthetic weights whose foreground logit is biased down so that a few hundred anchors per map pass 0.7 (random
    weights put ~half of the 393 216 anchors above it -- far beyond the candidate capacity)."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    m = init_synthetic_weights(FaFNet(Config("train")), seed=seed)
    with torch.no_grad():
        m.classification.conv2.bias[1::2] += fg_bias
    return m.to(device)


@pytest.mark.parametrize("thr", [0.5, 0.99, 0.99995, 0.9999999, 1.0])
def test_fused_detection_heads_candidate_counts_at_thresholds_near_one(device, thr):
    """the fused heads pre-test c1 - c0 against logit(thr) minus a slack before the exact fp32 softmax test.  Near 1 one ulp of the score
    is a large logit step and the score saturates at exactly 1.0f: the slack scales with 1 / (thr (1 - thr)) and the pre-test is capped at 16.
    The candidate COUNTS must equal the logits path's at every threshold, also where scores saturate (logits scaled up for that)."""
    from v2x_sim_amd import ops
    m = _heads_model(device, fg_bias=6.0)
    with torch.no_grad():
        m.classification.conv2.weight *= 4.0          # wide score distribution: many scores within an ulp of 1
    m.repack()
    pk = m.packed(device)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 64, 96, 32, generator=g) * 0.7).relu().to(torch.bfloat16).to(device)
    anchors = torch.randn(64 * 96 * 6, 6, generator=g).to(device)
    anchors[:, 2:4] = anchors[:, 2:4].abs() + 1.0
    cap = 4096
    cls, loc = ops.run_layer(pk["heads"], x)
    cls, loc = cls.reshape(2, -1, 2), loc.reshape(2, -1, 6)
    want = (torch.softmax(cls, -1)[..., 1] >= thr).sum(1)          # informational: torch's softmax may differ from the kernels' formula by an ulp
    keys, codes, counts = ops.conv2d_det(pk["heads"].det, x, thr, cap)
    ref_counts = ops.det_postprocess(cls, loc, anchors, thr, 1.0, cap)[3].abs()      # (overflow: minus the true count)
    assert torch.equal(counts.abs(), ref_counts), (thr, counts.tolist(), ref_counts.tolist(), want.tolist())


@pytest.mark.parametrize("N,H,W", [(3, 256, 256), (2, 64, 96), (5, 8, 32)])
def test_fused_detection_heads_equal_logits_then_postprocess(device, N, H, W):
    """The heads with the score threshold in the epilogue + v2x_det_nms_candidates against the heads' logits + v2x_det_postprocess:
    the same candidates (counts), and bit for bit the same kept boxes, scores and anchor indices, stand-up and rotated NMS; the
    candidate set is the oracle's (softmax(cls)[1] >= thr on the very logits the other launch wrote)."""
    from oracle import postprocess_ref as PR
    from v2x_sim_amd import ops
    m = _heads_model(device)
    pk = m.packed(device)
    g = torch.Generator().manual_seed(N + H)
    x = (torch.randn(N, H, W, 32, generator=g) * 0.7).relu().to(torch.bfloat16).to(device)
    anchors = torch.randn(H * W * 6, 6, generator=g).to(device)
    anchors[:, 2:4] = anchors[:, 2:4].abs() + 1.0
    thr, cap = 0.7, 4096
    cls, loc = ops.run_layer(pk["heads"], x)
    cls, loc = cls.reshape(N, -1, 2), loc.reshape(N, -1, 6)
    keys, codes, counts = ops.conv2d_det(pk["heads"].det, x, thr, cap)
    # the candidate sets agree: with an NMS threshold of 1.0 nothing is suppressed, so v2x_det_postprocess's count IS its candidate count
    assert torch.equal(ops.det_postprocess(cls, loc, anchors, thr, 1.0, cap)[3], counts)
    if H * W <= 64 * 96:                                                  # ... and it is the oracle's (scalar fp64 softmax) up to threshold ties
        n_ref = sum(1 for c in cls[0].cpu().tolist() if PR.fg_score(c[0], c[1]) >= thr)
        assert abs(int(counts[0]) - n_ref) <= 1
    assert 0 < int(counts.min()) and int(counts.max()) <= cap, counts.tolist()
    for rotated in (False, True):
        b0, s0, i0, c0 = ops.det_postprocess(cls, loc, anchors, thr, 0.05, cap, rotated=rotated)
        b1, s1, i1, c1 = ops.det_nms_candidates(keys, codes, counts, anchors, 0.05, rotated=rotated)
        assert torch.equal(c0, c1) and int(c0.min()) > 0
        for i in range(N):
            k = int(c0[i])
            assert torch.equal(i0[i, :k], i1[i, :k]) and torch.equal(s0[i, :k], s1[i, :k]) and torch.equal(b0[i, :k], b1[i, :k]), (rotated, i)
    # every candidate's six codes are the logits' codes of its anchor
    i = 0
    k = int(counts[i])
    anchor = ((keys[i, :k] & 0xffffffff) >> 12).long()
    slot = (keys[i, :k] & 0xfff).long()
    assert torch.equal(codes[i][slot], loc[i][anchor]) and sorted(slot.tolist()) == list(range(k))
    # overflow: more candidates than slots -> the true count is reported, the NMS entry signals -count like v2x_det_postprocess
    keys2, codes2, counts2 = ops.conv2d_det(pk["heads"].det, x, 0.0, 64)
    assert counts2.tolist() == [H * W * 6] * N
    assert ops.det_nms_candidates(keys2, codes2, counts2, anchors, 0.05)[3].tolist() == [-H * W * 6] * N


def test_predict_all_fused_detection_equals_logits_path(device):
    """FaFModule.predict_all with the fused heads (default) and with the logits path: identical detections; a batch that overflows the
    candidate capacity falls back to the logits path (and its host NMS) instead of losing maps."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    from v2x_sim_amd.utils.synthetic import synthetic_points
    from oracle import voxelize_ref as VR
    cfg = Config("train")
    m = _heads_model(device, seed=5, fg_bias=0.0)
    bev = torch.from_numpy(np.stack([VR.voxelize_occupy(p) for p in synthetic_points(2, 20000, seed=4)])[:, None]).to(device)
    data = {"bev_seq": bev}
    with torch.no_grad():                                       # calibrate the foreground bias: ~0.05 % of the anchors above 0.7
        cls = m(bev)["cls"].float()
        margin = (cls[..., 1] - cls[..., 0]).flatten()
        q = torch.quantile(margin[torch.randperm(margin.numel(), device=margin.device)[:2000000]], 0.9995)
        m.classification.conv2.bias[1::2] += float(np.log(0.7 / 0.3) - q)
    mod = FaFModule(m, None, cfg, None, 0)
    mod.nms_thr = 0.05
    fused = mod.predict_all(data, 2, num_agent=1)[3][0]
    assert "det" in mod.last_result and "cls" not in mod.last_result
    mod.fused_detection = False
    plain = mod.predict_all(data, 2, num_agent=1)[3][0]
    assert "cls" in mod.last_result
    assert sum(a["boxes"].shape[0] for a in fused) > 0
    for a, b in zip(fused, plain):
        assert np.array_equal(a["boxes"], b["boxes"]) and np.array_equal(a["scores"], b["scores"])
    mod.fused_detection, mod.cap = True, 64                      # far too few slots: the call must still answer, from the logits
    small = mod.predict_all(data, 2, num_agent=1)[3][0]
    assert "cls" in mod.last_result
    for a, b in zip(small, plain):
        assert np.allclose(a["boxes"], b["boxes"], atol=1e-4) and a["boxes"].shape == b["boxes"].shape


def test_box_axis_second_reading_as_a_switch(device):
    """oracle/ASSUMPTIONS.md row 48, the second reading behind Config.box_wh_axis ("h_along_heading": the extent the codes call h runs along the
    heading): FaFModule's device path (the first reading's kernels on (w, h)-exchanged anchors and codes, columns swapped back), its host path and
    the independent oracle (postprocess_ref.detect(..., wh_axis=...)) keep the same anchors and give the same boxes, scores and corners -- for the
    logits path and for the fused-candidates path -- and the switch is not a no-op: the kept set differs from the first reading's on this fixture."""
    import os
    from oracle import postprocess_ref as PR
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils import postprocess as P
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "postprocess_small.npz"))
    cls, loc, anc = g["cls"], g["loc"].reshape(-1, 6).copy(), g["anchors"].reshape(-1, 6)
    rng = np.random.default_rng(5)
    loc[:, 2] += rng.normal(0, 0.3, loc.shape[0]).astype(np.float32)          # make w and h codes differ, so that exchanging them matters
    kept = {}
    for axis in P.WH_AXES:
        cfg = Config("test")
        cfg.box_wh_axis = axis
        cfg.map_dims = [32, 32, 13]

        class _M(torch.nn.Module):
            pass
        mod = FaFModule(_M(), None, cfg, None, 0)
        mod.anchors = g["anchors"]
        mod.nms_thr = 0.05
        res = {"cls": torch.from_numpy(cls)[None].to(device), "loc": torch.from_numpy(loc)[None].to(device)}
        dev = mod.postprocess(res)[0]
        host = P.apply_nms_det(loc.reshape(32, 32, 6, 6), cls, g["anchors"], mod.score_thr, mod.nms_thr, wh_axis=axis)
        ref = PR.detect(cls, loc, anc, mod.score_thr, mod.nms_thr, wh_axis=axis)
        assert len(ref) == dev["boxes"].shape[0] == host["boxes"].shape[0] >= 4
        assert np.allclose(dev["boxes"], np.asarray([d["box"] for d in ref]), atol=1e-4) and np.allclose(dev["boxes"], host["boxes"], atol=1e-4)
        assert np.allclose(dev["corners"], np.asarray([d["corners"] for d in ref]), atol=1e-4)
        assert np.allclose(dev["scores"], np.asarray([d["score"] for d in ref]), atol=1e-6)
        kept[axis] = [d["index"] for d in ref]
    assert kept["w_along_heading"] != kept["h_along_heading"]
