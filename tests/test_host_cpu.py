"""CPU tests of the host logic: configs, weight packing layouts, frame plans, error behaviour
(the product path must refuse to run without the MI355X -- no CPU fallback)."""
import numpy as np
import pytest
import torch

from v2x_sim_amd import packing
from v2x_sim_amd.configs import Config
from v2x_sim_amd.models.det import FaFNet, V2VNet, When2com
from v2x_sim_amd.models.det.base import IntermediateModelBase
from v2x_sim_amd.models.seg import V2VNetSeg
from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_poses


def test_config_values():
    cfg = Config("train")
    assert cfg.map_dims == [256, 256, 13]
    assert len(cfg.anchor_size) == 6 and cfg.category_num == 2 and cfg.box_code_size == 6
    assert Config("train", is_cross_road=True).map_dims == [256, 256, 13]


def test_param_counts_and_state_dict_names():
    cfg = Config("train")
    m = FaFNet(cfg)
    assert sum(p.numel() for p in m.parameters()) == 7896656  # 7.89 M
    keys = set(V2VNet(cfg).state_dict().keys())
    for k in ("u_encoder.conv_pre_1.weight", "u_encoder.conv3d_1.conv3d.weight", "u_encoder.bn4_2.running_var",
              "decoder.conv5_1.weight", "classification.conv2.bias", "regression.box_prediction.3.weight",
              "convgru.weight_ih_l0", "convgru.bias_hh_l0"):
        assert k in keys, k
    w = When2com(cfg).state_dict()
    assert w["key_net.fc.0.weight"].shape == (256, 4096)
    assert w["attention_net.linear.weight"].shape == (1024, 32)
    assert w["query_key_net.conv1.cbr_unit.0.weight"].shape == (512, 512, 3, 3)


def test_models_refuse_cpu():
    cfg = Config("train")
    m = init_synthetic_weights(V2VNet(cfg))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(5, 1, 256, 256, 13), torch.zeros(1, 5, 5, 4, 4), torch.full((1, 5), 5))
    with pytest.raises(RuntimeError):
        m.packed("cpu")
    with pytest.raises(RuntimeError, match="parameter container"):
        m.u_encoder(torch.zeros(1))
    with pytest.raises(NotImplementedError):
        FaFNet(cfg, kd_flag=1)


def test_pack_conv_layout():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 13, 3, 3, generator=g)
    pc = packing.pack_conv("t", w, torch.ones(24), torch.zeros(24), cin_pad=16, device="cpu")
    assert pc.w_rows == 32 and pc.w_kpad == 192 and pc.C0 == 16 and pc.Cout == 24
    wp = pc.weight.float()
    # k = (ky*3 + kx)*16 + c
    for (co, c, ky, kx) in ((0, 0, 0, 0), (5, 12, 1, 2), (23, 7, 2, 2)):
        assert wp[co, (ky * 3 + kx) * 16 + c] == w[co, c, ky, kx].bfloat16().float()
    assert wp[:, 9 * 16:].abs().sum() == 0 and wp[24:].abs().sum() == 0
    assert wp.view(32, 12, 16)[:, :9, 13:].abs().sum() == 0  # padded channels


def test_pack_gru_layout():
    g = torch.Generator().manual_seed(1)
    hid, cin = 32, 64
    w = torch.randn(3 * hid, cin, 3, 3, generator=g)
    bi, bh = torch.randn(3 * hid, generator=g), torch.randn(3 * hid, generator=g)
    pc = packing.pack_gru("g", w, bi, bh, C0=32, C1=32, device="cpu")
    assert pc.w_rows == 96 and pc.w_kpad == 576 and pc.Cout == hid
    wp = pc.weight.float()
    for gate in range(3):
        for hc in (0, 15, 16, 31):
            row = (hc // 16) * 48 + gate * 16 + hc % 16
            assert wp[row, (1 * 3 + 2) * cin + 5] == w[gate * hid + hc, 5, 1, 2].bfloat16().float()
    b4 = pc.scale
    assert torch.allclose(b4[:, 0], bi[:hid] + bh[:hid]) and torch.allclose(b4[:, 3], bh[2 * hid:])
    with pytest.raises(ValueError):
        packing.pack_gru("g", torch.randn(3 * 24, 64, 3, 3), torch.zeros(72), torch.zeros(72), C0=32, C1=32, device="cpu")


def test_fold_bn_matches_torch():
    conv = torch.nn.Conv2d(4, 6, 3, padding=1)
    bn = torch.nn.BatchNorm2d(6).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 4, 5, 5)
    s, t = packing.fold_bn(conv.bias, bn, 6)
    with torch.no_grad():
        ref = bn(conv(x))
        got = torch.nn.functional.conv2d(x, conv.weight, None, 1, 1) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    assert torch.allclose(ref, got, atol=1e-5)


def test_frame_plan():
    nat = torch.tensor([[5] * 5, [3] * 5])
    counts, items, rows = IntermediateModelBase.frame_plan(nat, 2, 5)
    assert counts == [5, 3]
    assert items == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (4, 0)]
    assert rows == [0, 1, 2, 3, 4, 5, 6, 8]
    with pytest.raises(ValueError):
        IntermediateModelBase.frame_plan(torch.tensor([[6] * 5]), 1, 5)
    m = V2VNet(Config("train"))
    with pytest.raises(RuntimeError, match="non-empty"):
        m.make_plan(torch.tensor([[1] * 5]), 1, "cpu")
    plan = m.make_plan(nat, 2, "cpu")
    assert plan["rows"].tolist() == rows and plan["coef"].shape == (8, 5)
    assert plan["coef"][0].tolist() == [0, 1, 1, 1, 1] and plan["coef"][1].tolist() == [0, 1, 1, 0, 0]
    assert m.make_plan(torch.full((2, 5), 5), 2, "cpu")["rows"] is None


def test_synthetic_poses_are_consistent():
    T = synthetic_poses(2, 4, seed=0)
    assert T.shape == (2, 4, 4, 4, 4)
    for i in range(4):
        assert np.allclose(T[0, i, i], np.eye(4), atol=1e-5)
        for j in range(4):
            assert np.allclose(T[0, i, j] @ T[0, j, i], np.eye(4), atol=1e-4)


def test_seg_model_has_head():
    m = V2VNetSeg(Config("train"), n_classes=8)
    assert m.outc.conv.weight.shape == (8, 32, 1, 1)


def test_voxelize_wrapper_validates_the_count_vector():
    """ops.voxelize_bits / voxelize_fused_bits check n_pts against the number of clouds before any launch."""
    import pytest
    import torch
    from v2x_sim_amd import ops
    grid = ops.VoxelGrid()
    pts = torch.zeros((3, 16, 4))
    with pytest.raises(ValueError, match="one count per cloud"):
        ops.voxelize_bits(pts, torch.zeros(2, dtype=torch.int32), grid)
    with pytest.raises(ValueError, match="one count per cloud"):
        ops.voxelize_fused_bits(pts, torch.zeros(4, dtype=torch.int32), torch.zeros((1, 3, 4)), torch.zeros(1, dtype=torch.int32),
                                torch.zeros(1, dtype=torch.int32), 1, grid)


def test_param_version_sees_fused_optimizer_steps():
    """A fused optimizer step leaves Tensor._version alone; packing.watch_optimizer's post-step hook is what the packed-weight caches see."""
    import torch
    from v2x_sim_amd import packing
    p = torch.nn.Parameter(torch.randn(4, 4))
    for fused in (False, True):
        opt = packing.watch_optimizer(torch.optim.Adam([p], lr=1e-3, fused=fused))
        assert packing.watch_optimizer(opt) is opt          # idempotent: one hook
        p.grad = torch.ones_like(p)
        v0 = packing.param_version(p)
        opt.step()
        v1 = packing.param_version(p)
        assert v1 != v0 and v1[1] == v0[1] + 1, (fused, v0, v1)


def test_unwatched_optimizer_is_stamped_by_the_global_hook():
    """a caller who builds his own (fused) optimizer and never calls watch_optimizer must not train on stale packed weights: the
    process-wide post-step hook packing installs at import stamps the parameters of ANY optimizer; a watched one is not stamped twice."""
    import torch
    from v2x_sim_amd import packing
    assert packing.GLOBAL_OPTIMIZER_HOOK
    p = torch.nn.Parameter(torch.randn(4, 4))
    for fused in (False, True):
        opt = torch.optim.Adam([p], lr=1e-3, fused=fused)           # never handed to packing
        p.grad = torch.ones_like(p)
        v0 = packing.param_version(p)
        opt.step()
        v1 = packing.param_version(p)
        assert v1 != v0 and v1[1] == v0[1] + 1, (fused, v0, v1)
    sgd = torch.optim.SGD([p], lr=0.1)
    p.grad = torch.ones_like(p)
    e0 = packing.param_version(p)[1]
    sgd.step()
    assert packing.param_version(p)[1] == e0 + 1


def test_unhookable_optimizer_is_stamped_by_stepped():
    """An optimizer object without register_step_post_hook cannot be watched; the training loops call packing.stepped(opt) after opt.step(), which
    stamps the parameters itself (and does nothing for a watched optimizer, whose hook already did)."""
    import torch
    from v2x_sim_amd import packing

    class Bare:                                  # the minimum the loops use
        def __init__(self, params):
            self.param_groups = [{"params": list(params)}]

        def step(self):
            pass
    p = torch.nn.Parameter(torch.zeros(3))
    opt = packing.watch_optimizer(Bare([p]))
    assert not opt.__dict__.get("_v2x_watched")
    v0 = packing.param_version(p)
    opt.step()
    packing.stepped(opt)
    assert packing.param_version(p)[1] == v0[1] + 1
    real = packing.watch_optimizer(torch.optim.SGD([p], lr=0.1))
    p.grad = torch.ones(3)
    v1 = packing.param_version(p)
    real.step()
    packing.stepped(real)
    assert packing.param_version(p)[1] == v1[1] + 1       # once, by the hook


def test_latency_dispatch_is_per_thread():
    """the declared-latency state is thread-local -- a forward in one thread inside `with ops.latency_dispatch()` must not switch
    launches of another thread (a sharded runner, a training step) to the split-K / 1-tap forms."""
    import threading
    from v2x_sim_amd import ops, tuning
    old = tuning.set("SMALL_BATCH", 2)
    try:
        seen = {}
        inside, release = threading.Event(), threading.Event()

        def worker():
            with ops.latency_dispatch():
                seen["worker_in"] = ops.latency_launches()
                inside.set()
                release.wait(5)
            seen["worker_out"] = ops.latency_launches()

        t = threading.Thread(target=worker)
        t.start()
        assert inside.wait(5)
        seen["main_while_worker_inside"] = ops.latency_launches()
        with ops.latency_dispatch():
            with ops.latency_dispatch():
                seen["main_nested"] = ops.latency_launches()
            seen["main_in"] = ops.latency_launches()
        seen["main_out"] = ops.latency_launches()
        release.set()
        t.join()
        assert seen == {"worker_in": True, "main_while_worker_inside": False, "main_nested": True, "main_in": True, "main_out": False,
                        "worker_out": False}, seen
    finally:
        tuning.set("SMALL_BATCH", old)


def test_use_hip_adam_leaves_cpu_and_foreign_optimizers_alone():
    """train/optim.py::use_hip_adam switches only a plain torch.optim.Adam whose parameters are CUDA fp32 tensors (the kernel has no CPU form)."""
    from v2x_sim_amd.train.optim import use_hip_adam
    p = torch.nn.Parameter(torch.randn(5))
    for opt in (torch.optim.Adam([p], lr=1e-3), torch.optim.SGD([p], lr=1e-3), torch.optim.AdamW([p], lr=1e-3)):
        cls = type(opt)
        assert use_hip_adam(opt) is opt and type(opt) is cls
    assert use_hip_adam(None) is None
