"""Detection-level parity (the second half of BASELINE.json's metric).

PARITY UNPINNED w.r.t. the reference, and NO trained weights exist here (README.md:46 is an external
download).  With random weights the foreground scores of ~400k anchors sit within a few 1e-2 of each
other, so greedy NMS -- and therefore mAP -- is a chaotic function of rounding noise: the oracle's own
fp32 and bf16-emulating runs differ by 5-10 mAP points (tests/diag_map.py).  What CAN be pinned, and is:

  1. pre-NMS: for every anchor whose oracle score is further than DELTA from the threshold, the HIP path
     makes the same keep/drop decision, and the decoded boxes of kept anchors agree to centimetres;
  2. post-NMS: on a large detection set (hundreds of boxes) the HIP path's mAP@0.5/0.7 lies inside the band spanned
     by the oracle's fp32 and bf16-emulating runs (+-3 points) -- i.e. as close as rounding chaos allows.
The north_star's "+-0.2 mAP" needs a trained detector with separated scores and is NOT evidenced.
"""
import numpy as np
import pytest
import torch

from oracle import coperception_ref as R
from oracle import voxelize_ref as VR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def outputs(device):
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
    A, B = 5, 1
    cfg = Config("test")
    pm = init_synthetic_weights(V2VNet(cfg), seed=0)
    om = R.V2VNet().eval()
    om.load_state_dict(pm.state_dict())
    pm = pm.to(device)
    pts = synthetic_points(A * B, 20000, seed=41)
    bev = torch.from_numpy(np.stack([VR.voxelize_occupy(p) for p in pts])[:, None])
    T = torch.from_numpy(synthetic_poses(B, A, seed=42))
    nat = torch.full((B, A), A)
    with torch.no_grad():
        ref = om(bev, T, nat, batch_size=B)
        om.emulate_bf16 = True
        emu = om(bev, T, nat, batch_size=B)
    data = {"bev_seq": bev.to(device), "trans_matrices": T.to(device), "num_agent": nat}
    return cfg, pm, data, ref, emu, A, B


def test_pre_nms_decisions_and_boxes(outputs, device):
    from v2x_sim_amd.utils import postprocess as P
    cfg, pm, data, ref, emu, A, B = outputs
    with torch.no_grad():
        got = pm(data["bev_seq"], data["trans_matrices"], data["num_agent"], batch_size=B)
    fg_ref = P.softmax_fg(ref["cls"].numpy().reshape(-1, 2))
    fg_hip = P.softmax_fg(got["cls"].float().cpu().numpy().reshape(-1, 2))
    diff = np.abs(fg_hip - fg_ref)
    print("fg score |HIP - oracle|: mean %.5f max %.5f" % (diff.mean(), diff.max()))
    assert diff.max() < 2e-2 and diff.mean() < 3e-3            # scores are probabilities in [0, 1]
    thr = float(np.quantile(fg_ref, 0.95))
    DELTA = 2e-2
    stable = np.abs(fg_ref - thr) > DELTA
    assert np.array_equal((fg_hip >= thr)[stable], (fg_ref >= thr)[stable])   # identical keep/drop decisions
    anchors = P.build_anchor_map(cfg).reshape(-1, 6)
    anchors = np.tile(anchors, (A * B, 1))
    keep = np.nonzero((fg_ref >= thr) & stable)[0]
    b_ref = P.decode_boxes(ref["loc"].numpy().reshape(-1, 6)[keep], anchors[keep])
    b_hip = P.decode_boxes(got["loc"].float().cpu().numpy().reshape(-1, 6)[keep], anchors[keep])
    dxy = np.abs(b_hip[:, :2] - b_ref[:, :2]).max()
    dsz = np.abs(b_hip[:, 2:4] / b_ref[:, 2:4] - 1).max()
    dyaw = np.abs(np.angle(np.exp(1j * (b_hip[:, 4] - b_ref[:, 4]))))
    print("kept %d anchors: centre diff max %.3f m, size ratio diff max %.3f, yaw diff 99.9%% %.3f rad" % (
        keep.size, dxy, dsz, np.quantile(dyaw, 0.999)))
    assert dxy < 0.1 and dsz < 0.1 and np.quantile(dyaw, 0.999) < 0.2


def test_map_large_sample_vs_emulating_oracle(outputs, device):
    from v2x_sim_amd.utils import postprocess as P
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    cfg, pm, data, ref, emu, A, B = outputs
    anchors = P.build_anchor_map(cfg)
    fg_ref = P.softmax_fg(ref["cls"].numpy().reshape(-1, 2))
    thr, nms = float(np.quantile(fg_ref, 0.95)), 0.5
    module = FaFModule(pm, None, cfg, None, 0)
    module.score_thr, module.nms_thr = thr, nms
    _, _, _, seq = module.predict_all(data, B, validation=False, num_agent=A)   # upstream call shape
    det_hip = [seq[k][0] for k in range(A)]
    det_emu = [P.apply_nms_det(emu["loc"][k].numpy(), emu["cls"][k].numpy(), anchors, thr, nms) for k in range(A)]
    det_ref = [P.apply_nms_det(ref["loc"][k].numpy(), ref["cls"][k].numpy(), anchors, thr, nms) for k in range(A)]
    rng = np.random.default_rng(7)
    gts = []
    for d in det_ref:   # synthetic GT: every other fp32-oracle detection, jittered
        b = d["boxes"][::2].copy()
        b[:, :2] += rng.normal(0, 0.15, (b.shape[0], 2))
        b[:, 2:4] *= rng.uniform(0.9, 1.1, (b.shape[0], 2))
        b[:, 4] += rng.normal(0, 0.05, b.shape[0])
        gts.append(P.box_corners(b))
    n_det = sum(d["scores"].shape[0] for d in det_hip)
    assert n_det > 150
    for iou in (0.5, 0.7):
        ap_ref, info = P.eval_map(det_ref, gts, iou)
        ap_emu, _ = P.eval_map(det_emu, gts, iou)
        ap_hip, _ = P.eval_map(det_hip, gts, iou)
        print("mAP@%.1f  oracle-fp32 %.2f  oracle-bf16emu %.2f  HIP %.2f   (gt %d, det %d)" % (
            iou, 100 * ap_ref, 100 * ap_emu, 100 * ap_hip, info["num_gt"], n_det))
        # the oracle's own two precisions bracket what rounding chaos can do to this metric; the HIP path must land
        # inside that band (+-3 points).  Measured across kernel revisions: HIP 45.4 / 46.6 / 48.7 with the band at
        # [44.7, 54.2] for mAP@0.5.
        lo, hi = min(ap_ref, ap_emu), max(ap_ref, ap_emu)
        assert 100 * lo - 3.0 <= 100 * ap_hip <= 100 * hi + 3.0, (iou, ap_ref, ap_emu, ap_hip)
