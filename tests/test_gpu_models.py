"""End-to-end GPU parity: the coperception-shaped models (FaFNet lower/upperbound, V2VNet,
when2com/who2com, V2VNet-seg) running on the HIP kernels vs the build-owned CPU oracle, on the
same seeded synthetic inputs and weights.  PARITY UNPINNED w.r.t. the reference itself.

Tolerances (bf16 storage / fp32 accumulate through ~25 layers), relative to max|ref|:
    max|diff| <= 3e-2 * max|ref|   and   mean|diff| <= 3e-3 * max|ref|
against BOTH the fp32 oracle (the spec) and the bf16-emulating oracle.  Measured on MI355X
(round 1): max 1.3-1.8e-2, mean 1.2-2.5e-3 for the models of round 1; round 2 measured every fusion baseline too (worst:
MeanFusion loc 2.6e-2 / 2.0e-3 against the fp32 oracle): the bar is the worst measured + ~15 % (round 1 allowed 4e-2 / 5e-3).  Per stage the emulating oracle is
matched to <= 1 bf16 ulp (tests/test_gpu_stages.py; the MFMA accumulation is as accurate as
torch-CPU fp32 vs fp64, 0.005-0.025 % of outputs round differently per layer), but end to end each
differently-rounded activation perturbs 9*Cout downstream sums and flips further roundings, so
the two bf16 pipelines decorrelate to the bf16 noise floor -- hence one tolerance for both.
"""
import os

import numpy as np
import pytest
import torch

from oracle import coperception_ref as R
from oracle import voxelize_ref as VR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

TOL_EMU = (3e-2, 3e-3)
TOL_FP32 = (3e-2, 3e-3)


def check(got, ref, tol, what):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(ref.abs().max())
    d = (got - ref).abs()
    mx, mean = float(d.max()) / scale, float(d.mean()) / scale
    print("%-28s max %.3e  mean %.3e  (rel. to max|ref|=%.3f)" % (what, mx, mean, scale))
    assert mx <= tol[0] and mean <= tol[1], (what, mx, mean)


def make_inputs(A, B, n_pts=20000, seed=1):
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    pts = synthetic_points(A * B, n_pts, seed=seed)
    bev = np.stack([VR.voxelize_occupy(p) for p in pts])[:, None]
    return pts, torch.from_numpy(bev), torch.from_numpy(synthetic_poses(B, A, seed=seed + 1))


def build(P, O, dev, seed=0, pkw=None, okw=None):
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    pm = init_synthetic_weights(P(Config("train"), **(pkw or {})), seed=seed)
    om = O(**(okw or {})).eval()
    om.load_state_dict(pm.state_dict())
    return pm.to(dev), om


def test_fafnet_lowerbound(device):
    from v2x_sim_amd.models.det import FaFNet
    pm, om = build(FaFNet, R.FaFNet, device)
    _, bev, _ = make_inputs(2, 1)
    with torch.no_grad():
        got = pm(bev.to(device))
        for emu, tol in ((True, TOL_EMU), (False, TOL_FP32)):
            om.emulate_bf16 = emu
            ref = om(bev)
            check(got["cls"], ref["cls"], tol, "fafnet cls emu=%s" % emu)
            check(got["loc"], ref["loc"], tol, "fafnet loc emu=%s" % emu)
    assert got["cls"].shape == (2, 256 * 256 * 6, 2) and got["loc"].shape == (2, 256, 256, 6, 1, 6)


def test_upperbound_full_size_points_to_logits(device):
    """BASELINE.json config 1 at its full size, end to end: 5 agents x 65 536 points, every ego grid = the union of all five
    sweeps moved into the ego frame (25 transform + scatter jobs, ONE launch) -> FaFNet on the HIP path.  The occupancy
    is bit-exact against the oracle's early fusion and the logits are inside the end-to-end tolerance of both oracles."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.models.det.base import LidarDecoder, LidarEncoder
    from v2x_sim_amd.utils.synthetic import synthetic_poses
    A, n = 5, 65536
    clouds = [VR.synthetic_points(n, seed=500 + a, n_edge=64) for a in range(A)]
    Tgeo = synthetic_poses(1, A, seed=17)[0]                                   # T[i, j]: agent j's frame -> agent i's frame
    pm, om = build(FaFNet, R.FaFNet, device, seed=2)
    pts = torch.from_numpy(np.stack(clouds)).to(device)
    cnt = torch.full((A,), n, dtype=torch.int32, device=device)
    xf = torch.tensor(np.stack([Tgeo[i, j][:3] for i in range(A) for j in range(A)]), dtype=torch.float32, device=device)
    src = torch.tensor([j for i in range(A) for j in range(A)], dtype=torch.int32, device=device)
    dst = torch.tensor([i for i in range(A) for j in range(A)], dtype=torch.int32, device=device)
    grid = ops.VoxelGrid()
    with torch.no_grad():
        bits = ops.voxelize_fused_bits(pts, cnt, xf, src, dst, A, grid)
        pk = pm.packed(device)
        feats = LidarEncoder.run(pk["enc"], bits, zbits=grid.dims[2])            # conv_pre_1 reads the bit grid directly
        got = pm.get_cls_loc_result(LidarDecoder.run(pk["dec"], *feats), pk["heads"])
    ref_bev = np.stack([VR.voxelize_early_fusion(clouds, [Tgeo[i, j] for j in range(A)]) for i in range(A)])
    assert np.array_equal(ops.bits_to_dense(bits, grid.dims[2]).cpu().numpy(), ref_bev), "merged occupancy is not bit-exact"
    merged_occ = ref_bev.reshape(A, -1).sum(1)
    single_occ = np.array([VR.voxelize_occupy(c).sum() for c in clouds])
    assert (merged_occ > 2.5 * single_occ).all()                                 # the union really holds several sweeps
    bev = torch.from_numpy(ref_bev)[:, None]
    with torch.no_grad():
        for emu, tol in ((True, TOL_EMU), (False, TOL_FP32)):
            om.emulate_bf16 = emu
            ref = om(bev)
            check(got["cls"], ref["cls"], tol, "upperbound cls emu=%s" % emu)
            check(got["loc"], ref["loc"], tol, "upperbound loc emu=%s" % emu)


@pytest.mark.parametrize("gnn_iter,source", [(1, "initial"), (2, "updated"), (3, "initial"), (3, "updated"), (3, "frozen")])
def test_v2vnet(device, gnn_iter, source):
    from v2x_sim_amd.models.det import V2VNet
    A, B = 5, 1
    pm, om = build(V2VNet, R.V2VNet, device, pkw=dict(gnn_iter_times=gnn_iter, neighbor_source=source),
                   okw=dict(gnn_iter_times=gnn_iter, neighbor_source=source))
    _, bev, T = make_inputs(A, B)
    nat = torch.full((B, A), A)
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat.to(device), batch_size=B)
        for emu, tol in ((True, TOL_EMU), (False, TOL_FP32)):
            om.emulate_bf16 = emu
            ref = om(bev, T, nat, batch_size=B)
            check(got["cls"], ref["cls"], tol, "v2vnet cls emu=%s" % emu)
            check(got["loc"], ref["loc"], tol, "v2vnet loc emu=%s" % emu)


@pytest.mark.parametrize("name,layer", [("V2VNet", 2), ("V2VNet", 4), ("MeanFusion", 2), ("MeanFusion", 4), ("MaxFusion", 2), ("CatFusion", 4)])
def test_fusion_at_layer_2_and_4(device, name, layer):
    """The drivers' --layer flag (SURVEY section 5 flag surface): fusion at the 128-channel 64x64 map (layer 2) and at the 512-channel
    16x16 map (layer 4) instead of the default layer 3 -- V2VNet (ConvGRU of that width: layer_channel follows the layer) and the
    FusionBase family, ragged batch, against both precisions of the oracle at the end-to-end tolerance of layer 3."""
    from v2x_sim_amd.models import det
    from v2x_sim_amd.models.det.base import LAYER_SHAPES
    A, B = 5, 2
    kw = dict(layer=layer)
    if name == "V2VNet":
        kw["layer_channel"] = LAYER_SHAPES[layer][0]
    pm, om = build(getattr(det, name), getattr(R, name), device, seed=6 + layer, pkw=kw, okw=kw)
    _, bev, T = make_inputs(A, B, n_pts=8000, seed=20 + layer)
    nat = torch.tensor([[5] * A, [4] * A])
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat, batch_size=B)
        for emu, tol in ((True, TOL_EMU), (False, TOL_FP32)):
            om.emulate_bf16 = emu
            ref = om(bev, T, nat, batch_size=B)
            check(got["cls"], ref["cls"], tol, "%s layer %d cls emu=%s" % (name, layer, emu))
            check(got["loc"], ref["loc"], tol, "%s layer %d loc emu=%s" % (name, layer, emu))
    # the flag must matter: the oracle fusing at layer 3 with the same weights gives logits far outside the tolerance just met
    if name in ("MeanFusion", "MaxFusion"):                      # (no layer-dependent parameters)
        o3 = getattr(R, name)(layer=3).eval()
        o3.load_state_dict(om.state_dict())
        with torch.no_grad():
            other = o3(bev, T, nat, batch_size=B)
        assert float((other["cls"] - ref["cls"]).abs().max()) > 3 * TOL_FP32[0] * float(ref["cls"].abs().max())


@pytest.mark.parametrize("name", ["SumFusion", "MeanFusion", "MaxFusion", "CatFusion", "DiscoNet"])
def test_simple_fusion_baselines(device, name):
    """Row f-4: sum / mean / max / cat / DiscoNet (pixel-weighted) intermediate fusion (one warp_fuse launch + 1x1 convs),
    ragged batch (second frame has 4 real agents)."""
    from v2x_sim_amd.models import det
    A, B = 5, 2
    pm, om = build(getattr(det, name), getattr(R, name), device, seed=4)
    _, bev, T = make_inputs(A, B, n_pts=8000, seed=9)
    nat = torch.tensor([[5] * A, [4] * A])
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat, batch_size=B)
        for emu, tol in ((True, TOL_EMU), (False, TOL_FP32)):
            om.emulate_bf16 = emu
            ref = om(bev, T, nat, batch_size=B)
            check(got["cls"], ref["cls"], tol, "%s cls emu=%s" % (name, emu))
            check(got["loc"], ref["loc"], tol, "%s loc emu=%s" % (name, emu))


def test_v2vnet_ragged_agents_and_batch(device):
    """B=2 frames, the second with only 3 real agents: padding agents keep their own features."""
    from v2x_sim_amd.models.det import V2VNet
    A, B = 5, 2
    pm, om = build(V2VNet, R.V2VNet, device)
    _, bev, T = make_inputs(A, B, n_pts=8000, seed=5)
    nat = torch.tensor([[5] * A, [3] * A])
    om.emulate_bf16 = True
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat, batch_size=B)
        ref = om(bev, T, nat, batch_size=B)
    check(got["cls"], ref["cls"], TOL_EMU, "v2vnet ragged cls")
    check(got["loc"], ref["loc"], TOL_EMU, "v2vnet ragged loc")


def test_v2vnet_golden_small(device):
    """64x64 grid, 3 agents: the committed oracle outputs (tests/golden/v2vnet_small.npz)."""
    import math
    from v2x_sim_amd.models.det import V2VNet
    g = np.load(os.path.join(GOLD, "v2vnet_small.npz"))
    A = 3
    pm, _ = build(V2VNet, R.V2VNet, device, seed=3, pkw=dict(num_agent=A), okw=dict(num_agent=A))
    wsum = float(sum(p.double().abs().sum() for p in pm.state_dict().values()))
    assert math.isclose(wsum, float(g["weight_abs_sum"]), rel_tol=1e-9), (
        "the seeded synthetic weights differ from the ones tests/golden/v2vnet_small.npz was generated with (torch RNG "
        "stream or init_synthetic_weights changed): regenerate with `python tests/golden/make_golden.py g_v2vnet_small` "
        "and commit the fixture -- a silent skip would un-pin the end-to-end known-answer test")
    shape = tuple(int(v) for v in g["bev_shape"])
    bev = torch.from_numpy(np.unpackbits(g["bev"])[: int(np.prod(shape))].reshape(shape).astype(np.float32))
    with torch.no_grad():
        got = pm(bev.to(device), torch.from_numpy(g["T"]).to(device), torch.full((1, A), A), batch_size=1)
    cls = got["cls"].view(A, 64, 64, 12)[:, ::4, ::4]
    loc = got["loc"].reshape(A, 64, 64, 36)[:, ::4, ::4]
    check(cls, torch.from_numpy(g["cls_emu"]), TOL_EMU, "golden cls (emu)")
    check(loc, torch.from_numpy(g["loc_emu"]), TOL_EMU, "golden loc (emu)")
    check(cls, torch.from_numpy(g["cls_fp32"]), TOL_FP32, "golden cls (fp32)")


def _separate_attention_scores(pm, om, bev, B):
    """With random weights the 5-way softmax is ~0.2 everywhere = the 'activated' threshold, so which links are selected
    is decided by bf16 noise and a comparison of the fused logits would be vacuous (VERDICT r1 'weak' #3).  Here the
    attention layer (W, b of MIMOGeneralDotProductAttention.linear -- the same parameters on both sides) is SOLVED so
    that, on this input, key . (W query + b) equals a designed logit matrix: per query the scores are a rotation of
    (0.45, 0.33, 0.073, 0.073, 0.073) -- 0.12 away from the 0.2 threshold and from the runner-up, five times the
    measured bf16 noise of the scores (<= 2e-2).  Keys / queries come from the fp32 oracle tower; frame 0 only."""
    A = om.agent_num
    with torch.no_grad():
        qk = om.query_key_net(bev.permute(0, 1, 4, 2, 3), False)
        K = torch.stack([om.key_net(qk, False)[B * i] for i in range(A)]).double()       # (A, 1024)
        Q = torch.stack([om.query_net(qk, False)[B * i] for i in range(A)]).double()     # (A, 32)
        base = torch.tensor([0.45, 0.33, 0.22 / 3, 0.22 / 3, 0.22 / 3], dtype=torch.float64)
        target = torch.stack([torch.roll(base, q) for q in range(A)], 1).log()            # [key k][query q]
        W = torch.linalg.pinv(K) @ target @ torch.linalg.pinv(Q.T)                         # K W Q^T == target exactly
        for m in (pm, om):
            lin = m.attention_net.linear
            lin.weight.copy_(W.float().to(lin.weight.device))
            lin.bias.zero_()
    return target.softmax(0)


@pytest.mark.parametrize("inference", ["softmax", "activated", "argmax_test"])
def test_when2com(device, inference):
    """BASELINE.json config 3 end to end: tower -> keys/queries -> handshake -> selection -> weighted warp -> decoder ->
    heads.  The scores are separated by construction, so the selection is identical on both sides and the logits are
    ALWAYS compared."""
    from v2x_sim_amd.models.det import When2com
    A, B = 5, 1
    pm, om = build(When2com, R.When2com, device)
    _, bev, T = make_inputs(A, B)
    nat = torch.full((B, A), A)
    want = _separate_attention_scores(pm, om, bev, B)
    om.emulate_bf16 = True
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat, training=False, inference=inference, batch_size=B)
        ref = om(bev, T, nat, training=False, inference=inference, batch_size=B)
    dp = (got["prob_action"].cpu() - ref["prob_action"]).abs().max()
    print("when2com prob (%s): max |HIP - oracle| %.3e, max |oracle - designed| %.3e"
          % (inference, float(dp), float((ref["prob_action"][0].double() - want).abs().max())))
    assert float(dp) <= 2e-2, "attention scores differ by %.3e" % float(dp)
    assert torch.equal(got["coef"].cpu() != 0, ref["coef"] != 0), "HIP and oracle selected different links"
    n_links = int((ref["coef"][0] != 0).sum())
    assert n_links == {"softmax": 25, "activated": 10, "argmax_test": 5}[inference], ref["coef"]
    check(got["coef"], ref["coef"], (5e-2, 1e-2), "when2com coef (%s)" % inference)
    check(got["cls"], ref["cls"], TOL_EMU, "when2com cls (%s)" % inference)
    check(got["loc"], ref["loc"], TOL_EMU, "when2com loc (%s)" % inference)
    assert abs(got["num_connect"] - ref["num_connect"]) < 1e-9


@pytest.mark.parametrize("attn_index,renormalize", [("qk", False), ("kq", True), ("qk", True)])
def test_when2com_open_readings_as_switches(device, attn_index, renormalize):
    """oracle/ASSUMPTIONS.md rows 30 and 31, the second readings behind constructor flags in oracle AND product (as row 25's three readings are):
    attn_index = "qk" reads the attention matrix transposed in the weighted sum; renormalize divides the 'activated' coefficients by their sum
    over the keys.  Same end-to-end bars as the default reading; the designed scores make the two attn_index readings select DIFFERENT links
    for a target (the matrix is not symmetric), so a flag that did nothing would fail."""
    from v2x_sim_amd.models.det import When2com
    A, B = 5, 1
    kw = dict(attn_index=attn_index, renormalize=renormalize)
    pm, om = build(When2com, R.When2com, device, pkw=kw, okw=kw)
    _, bev, T = make_inputs(A, B)
    nat = torch.full((B, A), A)
    _separate_attention_scores(pm, om, bev, B)
    om.emulate_bf16 = True
    with torch.no_grad():
        got = pm(bev.to(device), T.to(device), nat, training=False, inference="activated", batch_size=B)
        ref = om(bev, T, nat, training=False, inference="activated", batch_size=B)
        base = R.When2com().eval()
        base.load_state_dict(om.state_dict())
        base.emulate_bf16 = True
        ref_default = base(bev, T, nat, training=False, inference="activated", batch_size=B)
    assert torch.equal(got["coef"].cpu() != 0, ref["coef"] != 0)
    check(got["coef"], ref["coef"], (5e-2, 1e-2), "when2com coef (%s, renorm %s)" % (attn_index, renormalize))
    if renormalize:
        tot = got["coef"].sum(1)
        assert torch.allclose(tot, torch.ones_like(tot), atol=1e-5)
    check(got["cls"], ref["cls"], TOL_EMU, "when2com cls (%s, renorm %s)" % (attn_index, renormalize))
    check(got["loc"], ref["loc"], TOL_EMU, "when2com loc (%s, renorm %s)" % (attn_index, renormalize))
    # the switch is not a no-op: the second reading's logits differ from the first reading's by far more than the tolerance
    assert float((ref["cls"] - ref_default["cls"]).abs().max()) > 5 * TOL_EMU[0] * float(ref_default["cls"].abs().max()) or renormalize


def test_v2vnet_seg(device):
    from v2x_sim_amd import ops
    from v2x_sim_amd.models.seg import V2VNetSeg
    A, B = 5, 1
    pm, om = build(V2VNetSeg, R.V2VNetSeg, device)
    _, bev, T = make_inputs(A, B)
    nat = torch.full((B, A), A)
    om.emulate_bf16 = True
    with torch.no_grad():
        got = pm.forward_nhwc(pm._input_nhwc(bev.to(device)), T.to(device), nat, batch_size=B)
        ref = om(bev, T, nat, batch_size=B).permute(0, 2, 3, 1).contiguous()
    check(got, ref, TOL_EMU, "seg logits")
    # IoU parity: confusion matrix of the HIP argmax vs the oracle's labels; exact integer kernel,
    # and pixels whose top-2 logit gap exceeds the logit tolerance must agree exactly.
    label = ref.argmax(-1).to(torch.uint8)
    pred, conf = ops.seg_argmax_confusion(got, label.to(device))
    assert torch.equal(conf.cpu(), R.confusion_matrix(pred.cpu().long(), label))
    top2 = ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * TOL_EMU[0] * float(ref.abs().max())
    assert torch.equal(pred.cpu().long()[safe], label.long()[safe])
    agree = float((pred.cpu() == label).float().mean())
    print("seg argmax agreement with oracle: %.5f" % agree)
    assert agree > 0.97


@pytest.mark.parametrize("cls_name,N,H,W", [("V2VNetSeg", 5, 256, 256), ("FaFNetSeg", 3, 64, 96), ("FaFNetSeg", 2, 48, 80)])
def test_seg_fused_head_equals_two_layers(device, tune, cls_name, N, H, W):
    """conv8_2 chained with the 1x1 class head in one halo launch (SEG_FUSE = 1, the default) against the two layers (SEG_FUSE = 0): the hidden map is
    rounded to bf16 as the stand-alone layer stores it, so the fp32 logits agree bit for bit; a map that does not tile by 8 x 32 takes the two-layer
    path either way; the bit-grid input (zbits) equals the expanded NHWC input."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models import seg as S
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_poses
    pm = init_synthetic_weights(getattr(S, cls_name)(Config("test"), **({"num_agent": N} if cls_name == "V2VNetSeg" else {})), seed=5).to(device)
    g = torch.Generator().manual_seed(H)
    bits = ((torch.rand(N, H, W, generator=g) < 0.3).to(torch.int32) * torch.randint(0, 1 << 13, (N, H, W), generator=g, dtype=torch.int32)).to(device)
    x0 = ops.bits_to_nhwc(bits, 13, 32)
    if cls_name == "V2VNetSeg":
        T = torch.from_numpy(synthetic_poses(1, N, seed=3)).to(device)
        nat = torch.full((1, N), N)
        run = lambda inp, **kw: pm.forward_nhwc(inp, T, nat, batch_size=1, **kw)     # noqa: E731
    else:
        run = lambda inp, **kw: pm.forward_nhwc(inp, **kw)                            # noqa: E731
    with torch.no_grad():
        tune("SEG_FUSE", 0)
        two = run(x0)
        tune("SEG_FUSE", 1)
        one = run(x0)
        from_bits = run(bits, zbits=13) if (H % 8 == 0 and W % 32 == 0) else one
    assert one.shape == (N, H, W, 8) and one.dtype == torch.float32
    assert torch.equal(one, two) and torch.equal(from_bits, one)
    assert float(one.abs().max()) > 0


@pytest.mark.parametrize("n_classes,fused", [(6, False), (10, False), (12, True), (4, True)])
def test_seg_head_class_counts_other_than_8(device, tune, n_classes, fused):
    """the chained halo epilogue stores whole float4s per k-slot quarter, so only class counts that are multiples of 4 (<= 16) take the
    fused conv8_2 + head launch; any other count keeps the two-layer path instead of failing in v2x_conv2d -- and both give the SEG_FUSE = 0 logits."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models import seg as S
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    pm = init_synthetic_weights(S.FaFNetSeg(Config("test"), n_classes=n_classes), seed=n_classes).to(device)
    g = torch.Generator().manual_seed(n_classes)
    bits = ((torch.rand(2, 64, 96, generator=g) < 0.3).to(torch.int32) * torch.randint(0, 1 << 13, (2, 64, 96), generator=g, dtype=torch.int32)).to(device)
    x0 = ops.bits_to_nhwc(bits, 13, 32)
    assert (pm.packed(device).get("seg_fused") is not None) == fused
    with torch.no_grad():
        one = pm.forward_nhwc(x0)
        tune("SEG_FUSE", 0)
        two = pm.forward_nhwc(x0)
    assert one.shape == (2, 64, 96, n_classes) and torch.equal(one, two) and float(one.abs().max()) > 0


def test_points_to_logits_path(device):
    """a1 -> a7 without the dense fp32 BEV: voxelize on the GPU and feed the network directly."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.models.det import FaFNet
    pm, om = build(FaFNet, R.FaFNet, device)
    pts, bev, _ = make_inputs(2, 1, n_pts=30000, seed=9)
    grid = ops.VoxelGrid()
    cnt = torch.full((2,), pts.shape[1], dtype=torch.int32, device=device)
    bits = ops.voxelize_bits(torch.from_numpy(pts).to(device), cnt, grid)
    x0 = ops.bits_to_nhwc(bits, 13, 32)
    with torch.no_grad():
        a = pm.forward_nhwc(x0)
        b = pm(bev.to(device))
    assert torch.equal(a["cls"], b["cls"]) and torch.equal(a["loc"], b["loc"])


@pytest.mark.parametrize("world", [2, 5])
def test_sharded_equals_unsharded_bitwise(device, world):
    """SURVEY.md 8(e) oracle: R-rank output == 1-rank output, bit for bit.  One GPU plays every
    rank in turn; the all-gather is emulated by concatenating the ranks' encoder outputs."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.models.det.base import LidarEncoder
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    A, Bt = 5, 2
    pm, _ = build(V2VNet, R.V2VNet, device, pkw=dict(gnn_iter_times=2, neighbor_source="updated"))
    pts = torch.from_numpy(synthetic_points(A * Bt, 16384, seed=11)).to(device)
    cnt = torch.full((A * Bt,), 16384, dtype=torch.int32, device=device)
    trans = torch.from_numpy(synthetic_poses(Bt, A, seed=12)).to(device)
    nat = torch.tensor([[5] * A, [4] * A])
    with torch.no_grad():
        one = AgentShard(A, Bt, 0, 1)
        ref = ShardedV2VNet(pm, one).forward_points(pts, cnt, trans, one.fusion_plan(nat, device))
        shards = [AgentShard(A, Bt, r, world) for r in range(world)]
        pk = pm.packed(device)
        # round 0 of the exchange: every rank's encoder output, in rank order
        runners = [ShardedV2VNet(pm, s, exchange=None) for s in shards]
        enc = [rn.encode_points(pts[s.lo:s.hi], cnt[s.lo:s.hi], pk) for rn, s in zip(runners, shards)]
        gathered0 = torch.cat([e[pm.layer] for e in enc])
        # GNN round 1 on every rank, then the second exchange, then round 2 (neighbor_source='updated')
        plans = [s.fusion_plan(nat, device) for s in shards]
        state = {"g": gathered0}
        outs = []
        for phase in range(2):
            nxt = []
            for rn, s, e, p in zip(runners, shards, enc, plans):
                rn.exchange = lambda t, st=state: st["g"]
                m = pm
                saved = m.gnn_iter_num
                m.gnn_iter_num = 1
                feats = list(e)
                if phase == 1:
                    feats[m.layer] = state["cur"][s.lo:s.hi]
                cur = rn.fuse_local(feats, trans, p, pk)
                m.gnn_iter_num = saved
                nxt.append(cur)
            state["cur"] = torch.cat(nxt)
            state["g"] = state["cur"]
        from v2x_sim_amd.models.det.base import LidarDecoder
        for s, e, in zip(shards, enc):
            feats = list(e)
            feats[pm.layer] = state["cur"][s.lo:s.hi]
            outs.append(pm.get_cls_loc_result(LidarDecoder.run(pk["dec"], *feats), pk["heads"]))
    assert torch.equal(torch.cat([o["cls"] for o in outs]), ref["cls"])
    assert torch.equal(torch.cat([o["loc"] for o in outs]), ref["loc"])


def test_begin_finish_interleaved_equals_forward_points(device):
    """bench.py's step: two half-batches, begin(A) begin(B) finish(A) finish(B).  Interleaving must not change a bit
    of either half (separate buffers, no shared state between begin and finish)."""
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    A, Bt = 5, 2
    pm, _ = build(V2VNet, R.V2VNet, device)
    sh = AgentShard(A, Bt, 0, 1)
    rn = ShardedV2VNet(pm, sh)
    nat = torch.full((Bt, A), A)
    plan = sh.fusion_plan(nat, device)
    halves = []
    for h in range(2):
        pts = torch.from_numpy(synthetic_points(A * Bt, 16384, seed=70 + h)).to(device)
        cnt = torch.full((A * Bt,), 16384, dtype=torch.int32, device=device)
        trans = torch.from_numpy(synthetic_poses(Bt, A, seed=80 + h)).to(device)
        halves.append((pts, cnt, trans))
    with torch.no_grad():
        ref = [rn.forward_points(p, c, t, plan) for p, c, t in halves]
        fa = rn.begin(halves[0][0], halves[0][1])
        fb = rn.begin(halves[1][0], halves[1][1])
        got = [rn.finish(*fa, halves[0][2], plan), rn.finish(*fb, halves[1][2], plan)]
    for g, r in zip(got, ref):
        assert torch.equal(g["cls"], r["cls"]) and torch.equal(g["loc"], r["loc"])


def test_when2com_fused_key_query_mlp_equals_separate_mlps(device):
    """KmGenerator.pack_pair: the key and the query MLP as one chain of three launches (stacked first layer, block-diagonal second and third) --
    bit for bit the two separate three-launch chains (the zero blocks only add exact zeros in whole K chunks)."""
    from v2x_sim_amd.models.det import When2com
    from v2x_sim_amd.models.det.When2com import KmGenerator
    pm, _ = build(When2com, R.When2com, device)
    pk = pm.packed(device)
    g = torch.Generator().manual_seed(11)
    y = torch.relu(torch.randn(37, 4, 4, 256, generator=g)).to(torch.bfloat16).to(device)
    keys, querys = KmGenerator.run_pair(pk["keyquery"], y)
    k_ref, q_ref = KmGenerator.run(pk["key"], y), KmGenerator.run(pk["query"], y)
    assert keys.shape == (37, 1024) and querys.shape == (37, 32) and keys.is_contiguous() and querys.is_contiguous()
    assert torch.equal(keys, k_ref) and torch.equal(querys, q_ref)


def test_sharded_when2com_equals_unsharded_bitwise(device, tune):
    """BASELINE.json config 4: when2com agent-sharded.  Two virtual ranks on one GPU (the all-gathers are emulated by
    concatenating the ranks' tensors) must reproduce the unsharded model bit for bit, incl. a ragged frame.  (The plain class's
    forward_nhwc declares latency launches under the default SMALL_BATCH = 2, the runners never do: the reference is taken with the
    throughput forms, SMALL_BATCH = 0 -- the same kernels as the runners'.)"""
    tune("SMALL_BATCH", 0)
    from v2x_sim_amd import ops
    from v2x_sim_amd.models.det import When2com
    from v2x_sim_amd.parallel import AgentShard, ShardedWhen2com
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    A, Bt, world = 5, 2, 2
    pm, _ = build(When2com, R.When2com, device)
    grid = ops.VoxelGrid()
    pts = torch.from_numpy(synthetic_points(A * Bt, 16384, seed=21)).to(device)
    cnt = torch.full((A * Bt,), 16384, dtype=torch.int32, device=device)
    bits = ops.voxelize_bits(pts, cnt, grid)
    trans = torch.from_numpy(synthetic_poses(Bt, A, seed=22)).to(device)
    nat = torch.tensor([[5] * A, [4] * A])
    with torch.no_grad():
        ref = pm.forward_nhwc(ops.bits_to_nhwc(bits, 13, 32), trans, nat, training=False, inference="activated", batch_size=Bt)
        shards = [AgentShard(A, Bt, r, world) for r in range(world)]
        # pass 1: every rank's local tensors, recorded in exchange order (keys, querys, features)
        recorded = [[] for _ in range(world)]
        outs = []
        for phase in range(2):
            for r, s in enumerate(shards):
                k = {"i": 0}

                def exch(t, r=r, k=k):
                    i = k["i"]
                    k["i"] += 1
                    if phase == 0:
                        recorded[r].append(t)
                        # placeholder of the right shape so the pass can finish
                        return torch.cat([t] * world)
                    return torch.cat([recorded[q][i] for q in range(world)])
                rn = ShardedWhen2com(pm, s, exchange=exch)
                res = rn.forward_bits(bits[s.lo:s.hi].contiguous(), 13, trans, rn.plan(nat, device), training=False,
                                      inference="activated")
                if phase == 1:
                    outs.append(res)
    assert torch.equal(torch.cat([o["cls"] for o in outs]), ref["cls"])
    assert torch.equal(torch.cat([o["loc"] for o in outs]), ref["loc"])
    assert torch.equal(outs[0]["coef"], ref["coef"])


def test_full_size_end_to_end_properties(device):
    """The bench's half-batch (64 frames x 5 agents x 65 536 points -> logits) through size-independent properties, bit for bit:
    (a) a sweep is a SET of points: shuffling every cloud's points changes nothing (occupancy and therefore logits);
    (b) the frames of a batch are independent: permuting the frames (sweeps, poses, agent table together) permutes the logits.
    Both involve every kernel of the path (voxeliser, pair kernel, all conv kernels, warp + ConvGRU, heads) at the size the
    headline number is measured on."""
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
    from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
    A, Bt, n = 5, 64, 65536
    pm, _ = build(V2VNet, R.V2VNet, device)
    sh = AgentShard(A, Bt, 0, 1)
    rn = ShardedV2VNet(pm, sh)
    nat = torch.full((Bt, A), A)
    plan = sh.fusion_plan(nat, device)
    g = torch.Generator().manual_seed(5)
    pts = torch.from_numpy(synthetic_points(A * Bt, n, seed=123)).to(device)          # item = agent * Bt + frame
    cnt = torch.full((A * Bt,), n, dtype=torch.int32, device=device)
    trans = torch.from_numpy(synthetic_poses(Bt, A, seed=124)).to(device)
    with torch.no_grad():
        ref = rn.forward_points(pts, cnt, trans, plan)
        cls_ref, loc_ref = ref["cls"].clone(), ref["loc"].clone()
        del ref
        # (a) shuffle the points inside every cloud (one permutation per cloud)
        order = torch.argsort(torch.rand(A * Bt, n, generator=g), dim=1).to(device)
        pts_s = torch.gather(pts, 1, order[:, :, None].expand(-1, -1, pts.shape[2]))
        got = rn.forward_points(pts_s, cnt, trans, plan)
        assert torch.equal(got["cls"], cls_ref) and torch.equal(got["loc"], loc_ref), "the logits depend on the order of the points in a sweep"
        del got, pts_s, order
        # (b) permute the frames
        perm = torch.randperm(Bt, generator=g)
        item = (torch.arange(A)[:, None] * Bt + perm[None, :]).reshape(-1).to(device)   # new item a * Bt + f' <- old item a * Bt + perm[f']
        got = rn.forward_points(pts[item].contiguous(), cnt, trans[perm.to(device)].contiguous(), plan)
        assert torch.equal(got["cls"], cls_ref[item]) and torch.equal(got["loc"], loc_ref[item]), "the logits of a frame depend on its position in the batch"
