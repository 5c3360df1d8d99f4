"""Row f-3, first hand-written backward kernels: forward / data gradient / weight gradient of the 3x3 stride-1 convolutions
on libv2x_amd.so (v2x_conv2d on flipped weights, v2x_conv3x3_wgrad) behind torch.autograd.Function, against
torch.autograd (MIOpen / CPU fp32) on the SAME bf16-rounded operands.  Tolerances: bf16 storage of y / dx (one rounding of
the fp32 sums); dW is an fp32 sum of bf16 x bf16 products over N*H*W pixels -> 1e-3 of its scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 32, 64, 64), (3, 32, 64, 128, 128), (1, 8, 32, 32, 64), (5, 64, 64, 64, 128),
                                            (2, 24, 96, 192, 64)])
def test_wgrad_kernel_vs_autograd(device, N, H, W, Cin, Cout):
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(N + H + W + Cin + Cout)
    x = bf16r(torch.randn(N, Cin, H, W, generator=g))
    dy = bf16r(torch.randn(N, Cout, H, W, generator=g))
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dy)                       # CPU fp32 reference of the weight gradient
    got = ops.conv3x3_wgrad(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device),
                            dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device))
    assert got.shape == w.grad.shape and got.dtype == torch.float32
    err = float((got.cpu() - w.grad).abs().max()) / float(w.grad.abs().max())
    print("wgrad %s: max |diff| / max |ref| = %.2e" % ((N, H, W, Cin, Cout), err))
    assert err < 1e-3
    again = ops.conv3x3_wgrad(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device),
                              dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device))
    assert torch.equal(got, again), "the fixed-order reduction must be bit-reproducible"


def test_wgrad_borders_and_single_tap(device):
    """A delta image and a delta gradient pick single taps out: dW[co][ci][ky][kx] must be 1 exactly where the geometry
    says (zero padding at the image border, tap orientation), 0 elsewhere."""
    from v2x_sim_amd import ops
    N, H, W, Cin, Cout = 1, 8, 32, 32, 64
    x = torch.zeros(N, H, W, Cin)
    dy = torch.zeros(N, H, W, Cout)
    x[0, 0, 0, 3] = 1.0            # corner pixel
    dy[0, 0, 0, 5] = 1.0           # centre tap:     output (0,0) <- input (0,0)  => dW[5][3][1][1]
    dy[0, 1, 1, 6] = 1.0           # top-left tap:   output (1,1) <- input (0,0)  => dW[6][3][0][0]
    dy[0, 0, 1, 7] = 1.0           # left tap:       output (0,1) <- input (0,0)  => dW[7][3][1][0]
    got = ops.conv3x3_wgrad(x.to(torch.bfloat16).to(device), dy.to(torch.bfloat16).to(device)).cpu()
    want = torch.zeros(Cout, Cin, 3, 3)
    want[5, 3, 1, 1] = want[6, 3, 0, 0] = want[7, 3, 1, 0] = 1.0
    assert torch.equal(got, want)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 32, 64, 64), (2, 32, 32, 128, 256), (1, 8, 32, 32, 64)])
def test_hip_conv_function_forward_dgrad_wgrad(device, N, H, W, Cin, Cout):
    """The autograd Function end to end: y, dL/dx, dL/dW, dL/db of sum(conv(x) * g) vs torch on the same bf16 operands."""
    from v2x_sim_amd.train import hip_conv
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = bf16r(torch.randn(N, Cin, H, W, generator=gen))
    w = bf16r(torch.randn(Cout, Cin, 3, 3, generator=gen) * (2.0 / (9 * Cin)) ** 0.5)
    b = torch.randn(Cout, generator=gen) * 0.1
    gy = bf16r(torch.randn(N, Cout, H, W, generator=gen))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, 1)
    (yr * gy).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device).requires_grad_(True)
    wd, bd = w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    yd = hip_conv.conv3x3_nhwc(xd, wd, bd)
    (yd.float() * gy.permute(0, 2, 3, 1).to(device)).sum().backward()
    ulp = 2 ** -7
    y = yd.detach().float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(y, yr.detach(), atol=2e-3, rtol=ulp)
    dx = xd.grad.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(dx, xr.grad, atol=2e-3 * float(xr.grad.abs().max()), rtol=ulp), float((dx - xr.grad).abs().max())
    assert float((wd.grad.cpu() - wr.grad).abs().max()) < 1e-3 * float(wr.grad.abs().max())
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-3)
    assert not hip_conv.eligible(torch.zeros(64, 13, 3, 3), (1, 1), (1, 1), 256, 256)       # 13 input channels: MIOpen keeps it
    assert not hip_conv.eligible(torch.zeros(64, 32, 3, 3), (2, 2), (1, 1), 256, 256)       # stride 2: MIOpen keeps it


def test_training_step_on_hip_conv_kernels(device, tune):
    """V2X_TRAIN_HIP_CONV=1: one FaFNet training step with every eligible 3x3 layer (conv1_2 ... conv7_2: 13 layers) on the HIP
    forward / dgrad / wgrad kernels.  Loss within 1 % and every parameter gradient within 5 % (of the tensor's scale) of the
    all-MIOpen step -- bf16 activations and gradients through ~20 layers, batch-statistics BN in between."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import synthetic_batch_on_device
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    cfg = Config("train")
    model = init_synthetic_weights(FaFNet(cfg, kd_flag=0, num_agent=2), seed=3).to(device)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=5, device=device)
    model.eval()   # running-statistics BN: batch statistics amplify ReLU flips chaotically (see test_gpu_train.py)
    grads, losses = {}, {}
    for flag in ("0", "1"):
        tune("TRAIN_HIP_CONV", flag)
        model.zero_grad()
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        losses[flag] = float(loss.detach())
        grads[flag] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    tune.reset("TRAIN_HIP_CONV")
    print("loss MIOpen %.5f, HIP conv kernels %.5f" % (losses["0"], losses["1"]))
    assert abs(losses["1"] - losses["0"]) <= 1e-2 * abs(losses["0"])
    worst, worst_k = 0.0, ""
    for k, g0 in grads["0"].items():
        d = float((grads["1"][k] - g0).abs().max()) / max(float(g0.abs().max()), 1e-12)
        if d > worst:
            worst, worst_k = d, k
    print("worst relative gradient difference %.3e (%s)" % (worst, worst_k))
    assert worst < 5e-2, (worst, worst_k)


# ------------------------------------------------------------------ batch-statistics BN + ReLU kernels (bn_train.hip)
@pytest.mark.parametrize("layout", [1, 0])
@pytest.mark.parametrize("M,C,relu", [(2 * 256 * 256, 32, True), (3 * 64 * 64, 128, True), (5 * 16 * 16, 512, True), (1000, 64, False),
                                      (7, 8, True), (40 * 16 * 16, 512, True)])
def test_bn_train_kernels_vs_autograd(device, tune, M, C, relu, layout):
    """v2x_bn_train_forward / backward against F.batch_norm(training=True) (+ relu) in fp32 on the same bf16 inputs: statistics and
    parameter gradients to 1e-5 / 2e-4 of their scale, y and dx within one bf16 rounding, running statistics as nn.BatchNorm updates
    them, and bit-identical results on a second run (no atomics).  Both layouts of the per-workgroup partial sums (BN_PARTIAL_T: round 6's
    [kind][channel][workgroup], which the finish kernels read as contiguous floats, and the earlier [workgroup][kind][channel])."""
    from v2x_sim_amd import ops
    tune("BN_PARTIAL_T", layout)
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + torch.randn(C, generator=g)).to(torch.bfloat16)
    dy = torch.randn(M, C, generator=g).to(torch.bfloat16)
    gamma = (0.5 + torch.rand(C, generator=g)).requires_grad_(True)
    beta = (0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    rm0, rv0 = torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    # reference: fp32 on the CPU
    xr = x.float().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.batch_norm(xr, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.float())
    # device
    xd, dyd = x.to(device), dy.to(device)
    gd, bd = gamma.detach().to(device), beta.detach().to(device)
    rmd, rvd = rm0.to(device), rv0.to(device)
    y, mean, invstd = ops.bn_train_forward(xd, gd, bd, rmd, rvd, 1e-5, 0.1, relu)
    dx, dgamma, dbeta = ops.bn_train_backward(xd, dyd, gd, bd, mean, invstd, relu)
    mu_ref = x.float().mean(0)
    var_ref = x.float().var(0, unbiased=False)
    assert torch.allclose(mean.cpu(), mu_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(invstd.cpu(), 1.0 / torch.sqrt(var_ref + 1e-5), rtol=2e-5)
    assert torch.allclose(rmd.cpu(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv, rtol=2e-5, atol=1e-6)
    ybf = yr.detach().to(torch.bfloat16).float()
    tol_y = 2.0 ** -7 * ybf.abs().clamp(min=2.0 ** -6)            # one bf16 rounding step of the value
    assert bool(((y.cpu().float() - ybf).abs() <= tol_y).all())
    sg, sb = float(gamma.grad.abs().max()), float(beta.grad.abs().max())
    assert float((dgamma.cpu() - gamma.grad).abs().max()) <= 2e-4 * sg + 1e-4, (dgamma.cpu() - gamma.grad).abs().max()
    assert float((dbeta.cpu() - beta.grad).abs().max()) <= 2e-4 * sb + 1e-4
    dxr = xr.grad
    err = (dx.cpu().float() - dxr).abs()
    # a ReLU decided on the fp32 value vs on the value recomputed in another order can differ only where y ~ 0
    near_kink = yr.detach().abs() < 1e-5 if relu else torch.zeros_like(err, dtype=torch.bool)
    tol_dx = 2.0 ** -7 * dxr.abs() + 2.0 ** -8 * float(dxr.abs().max()) * 2.0 ** -4 + 1e-6
    assert bool(((err <= tol_dx) | near_kink).all()), float((err - tol_dx).max())
    y2, mean2, invstd2 = ops.bn_train_forward(xd, gd, bd, None, None, 1e-5, 0.1, relu)
    dx2, dgamma2, dbeta2 = ops.bn_train_backward(xd, dyd, gd, bd, mean2, invstd2, relu)
    assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(dx2, dx) and torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 64, 64, 32, 32), (1, 32, 64, 96, 32), (3, 8, 32, 64, 96)])
def test_wgrad_32_row_form_vs_autograd(device, N, H, W, Cin, Cout):
    """Cout % 64 == 32: the 32-row form of conv3x3_wgrad_kernel (wave pairs split the tile's rows, two workspace slots per block)."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(N * H + Cout)
    x = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).float()
    dy = torch.randn(N, Cout, H, W, generator=g).to(torch.bfloat16).float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    got = ops.conv3x3_wgrad(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device),
                            dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(device)).cpu()
    err = float((got - w.grad).abs().max()) / float(w.grad.abs().max())
    print("wgrad32 %s: %.2e" % ((N, H, W, Cin, Cout), err))
    assert err < 2e-5


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,cin_store", [(2, 64, 64, 32, 64, 2, 32), (1, 32, 32, 256, 512, 2, 256), (2, 64, 64, 13, 32, 1, 32),
                                                             (1, 64, 64, 96, 32, 1, 96), (2, 32, 32, 64, 128, 1, 64), (2, 16, 16, 64, 64, 1, 64)])
def test_hip_graph_conv_function_vs_autograd(device, N, H, W, Cin, Cout, stride, cin_store):
    """train/hip_graph.py::_Conv3x3 (forward, data gradient, weight gradient, bias gradient) for stride 1 and 2, 32-channel layers and the
    13-channel first layer stored as 32, against torch.autograd in fp32 on the same bf16-rounded operands."""
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(Cin + Cout + stride)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
    conv_r = torch.nn.Conv2d(Cin, Cout, 3, stride, 1)
    with torch.no_grad():
        conv_r.weight.copy_((conv_r.weight * 2).to(torch.bfloat16).float())
    conv_d = torch.nn.Conv2d(Cin, Cout, 3, stride, 1).to(device)
    conv_d.load_state_dict(conv_r.state_dict())
    Ho, Wo = H // stride, W // stride
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = conv_r(xr)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    xd = F.pad(x, (0, cin_store - Cin)).to(device).requires_grad_(cin_store == Cin)   # the padded first layer's input takes no gradient
    assert hip_graph.hip_eligible(conv_d.weight, stride, H, W) or W == 16      # 16-wide maps run zero-widened to 32
    yd = hip_graph.conv3x3(xd, conv_d)
    yd.backward(dy.to(device))
    ybf = yr.detach().permute(0, 2, 3, 1)
    assert float((yd.detach().cpu().float() - ybf).abs().max()) <= 2.0 ** -7 * float(ybf.abs().max())
    if cin_store == Cin:
        dxr = xr.grad.permute(0, 2, 3, 1)
        assert float((xd.grad.cpu().float() - dxr).abs().max()) <= 2.0 ** -7 * float(dxr.abs().max())
    assert float((conv_d.weight.grad.cpu() - conv_r.weight.grad).abs().max()) <= 2e-5 * float(conv_r.weight.grad.abs().max())
    assert torch.allclose(conv_d.bias.grad.cpu(), conv_r.bias.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("H,W,Cin,Cout,stride", [(24, 96, 64, 128, 2), (8, 32, 64, 128, 2), (40, 96, 128, 256, 2)])
def test_hip_graph_conv_extents_without_a_kernel_take_the_torch_path(device, H, W, Cin, Cout, stride):
    """pack_conv_device packs ONE layout and no gather fallback, so hip_eligible must be exactly run_layer's conditions.  These maps
    tile by 8 x 32 (the old test) but have no stride-2 kernel (24 x 96, 8 x 32, 40 x 96: neither 8 x 64 nor 16 x 32 input tiles):
    conv3x3 must route them through F.conv2d (they raised IndexError on the empty fallback list) and still match autograd."""
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(H + W + stride)
    conv_r = torch.nn.Conv2d(Cin, Cout, 3, stride, 1)
    conv_d = torch.nn.Conv2d(Cin, Cout, 3, stride, 1).to(device)
    conv_d.load_state_dict(conv_r.state_dict())
    assert not hip_graph.hip_eligible(conv_d.weight, stride, H, W, Cin)
    x = torch.randn(2, H, W, Cin, generator=g).to(torch.bfloat16)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = conv_r(xr)
    dy = torch.randn(2, H // stride, W // stride, Cout, generator=g).to(torch.bfloat16)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    xd = x.to(device).requires_grad_(True)
    yd = hip_graph.conv3x3(xd, conv_d)
    yd.backward(dy.to(device))
    ybf = yr.detach().permute(0, 2, 3, 1)
    assert float((yd.detach().cpu().float() - ybf).abs().max()) <= 2.0 ** -6 * float(ybf.abs().max())
    dxr = xr.grad.permute(0, 2, 3, 1)
    assert float((xd.grad.cpu().float() - dxr).abs().max()) <= 2.0 ** -6 * float(dxr.abs().max())
    assert float((conv_d.weight.grad.cpu() - conv_r.weight.grad).abs().max()) <= 1e-2 * float(conv_r.weight.grad.abs().max())


def test_hip_graph_training_step_vs_fp32_graph(device, monkeypatch, tune):
    """V2X_TRAIN_HIP=1: a FaFNet training step (batch-statistics BN) on the bf16 NHWC HIP graph against the fp32 MIOpen graph: the
    loss within 2 %, the running statistics of every BN within 2 % of their scale, and parameter gradients that point the same way:
    cosine over all parameters together > 0.95 and no worse than 0.02 below a CONTROL -- the fp32 graph itself with its weights rounded to
    bf16 (measured 0.973 vs 0.978: individual ReLU flips under batch statistics make per-element comparisons meaningless in train
    mode, see tests/test_gpu_train.py).  Then 30 SGD steps on either graph from the same start: both reduce the loss, to within 15 % of each other."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import synthetic_batch_on_device
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    cfg = Config("train")
    base = init_synthetic_weights(FaFNet(cfg, kd_flag=0, num_agent=2), seed=3).to(device)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=5, device=device)
    out = {}
    for flag in ("0", "1", "0r"):
        tune("TRAIN_HIP", int(flag[0]))
        model = copy.deepcopy(base)
        model.train()
        if flag == "0r":   # control: the fp32 graph with its weights rounded to bf16 -- the noise floor of this comparison
            with torch.no_grad():
                for p in model.parameters():
                    p.copy_(p.to(torch.bfloat16).float())
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        out[flag] = (float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None},
                     {k: b.detach().float().clone() for k, b in model.named_buffers() if "running" in k})
    l0, g0, b0 = out["0"]
    l1, g1, b1 = out["1"]
    print("loss: fp32 graph %.5f, HIP graph %.5f, fp32 graph on bf16-rounded weights %.5f" % (l0, l1, out["0r"][0]))
    assert abs(l1 - l0) <= 2e-2 * abs(l0)
    assert set(g0) == set(g1)
    for k in b0:
        assert float((b1[k] - b0[k]).abs().max()) <= 2e-2 * max(float(b0[k].abs().max()), 1e-3), k

    def cosine(ga, gb):
        dot = sum(float((ga[k] * gb[k]).sum()) for k in ga)
        na = sum(float((ga[k] ** 2).sum()) for k in ga) ** 0.5
        nb = sum(float((gb[k] ** 2).sum()) for k in ga) ** 0.5
        return dot / (na * nb), na, nb
    c_hip, n0, n1 = cosine(g0, g1)
    c_ctl, _, _ = cosine(g0, out["0r"][1])
    print("gradient cosine vs the fp32 graph: HIP graph %.4f, control (bf16-rounded weights) %.4f; norms %.4e / %.4e" % (c_hip, c_ctl, n0, n1))
    assert c_hip > 0.95 and c_hip >= c_ctl - 0.02 and abs(n1 - n0) <= 0.1 * n0
    finals = {}
    for flag in ("0", "1"):
        tune("TRAIN_HIP", flag)
        model = copy.deepcopy(base)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=2e-3, momentum=0.9)
        first = None
        for _ in range(30):
            res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
            loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            first = float(loss.detach()) if first is None else first
        finals[flag] = (first, float(loss.detach()))
    tune.reset("TRAIN_HIP")
    print("30 SGD steps: fp32 graph %.4f -> %.4f, HIP graph %.4f -> %.4f" % (finals["0"] + finals["1"]))
    assert finals["0"][1] < 0.7 * finals["0"][0] and finals["1"][1] < 0.7 * finals["1"][0]
    assert abs(finals["1"][1] - finals["0"][1]) <= 0.15 * finals["0"][1]


def test_hip_graph_v2vnet_step(device, tune):
    """V2VNet on the HIP training graph (encoder / decoder / heads on the kernels, warp + ConvGRU fusion on the fp32 graph in between):
    loss within 2 % of the fp32 graph's, every parameter that is in the graph (the ConvGRU's input weights included) receives a finite gradient,
    and FaFModule.step runs."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import synthetic_batch_on_device
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights
    cfg = Config("train")
    base = init_synthetic_weights(V2VNet(cfg, num_agent=2), seed=4).to(device)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=6, device=device)
    losses = {}
    for flag in ("0", "1"):
        tune("TRAIN_HIP", flag)
        model = copy.deepcopy(base)
        model.train()
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        losses[flag] = float(loss.detach())
        for k, p in model.named_parameters():
            if "weight_hh" in k:      # h0 = 0: W_hh never enters the graph, on either path
                assert p.grad is None
                continue
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    print("V2VNet loss: fp32 graph %.5f, HIP graph %.5f" % (losses["0"], losses["1"]))
    assert abs(losses["1"] - losses["0"]) <= 2e-2 * abs(losses["0"])
    model = copy.deepcopy(base)
    module = FaFModule(model, None, cfg, torch.optim.Adam(model.parameters(), lr=1e-4), 0)
    first = module.step(data, 1, num_agent=2)[0]
    for _ in range(9):
        last = module.step(data, 1, num_agent=2)[0]
    tune.reset("TRAIN_HIP")
    print("FaFModule.step on the HIP graph: loss %.4f -> %.4f in 10 steps" % (first, last))
    assert np.isfinite(last) and last < first


@pytest.mark.parametrize("N,H,W,Cin,Cout,f32_out", [(2, 64, 64, 64, 64, False), (1, 64, 64, 32, 12, True), (2, 32, 32, 32, 36, True),
                                                     (1, 32, 64, 128, 128, False)])
def test_hip_graph_conv1x1_vs_autograd(device, N, H, W, Cin, Cout, f32_out):
    """1x1 layers of the training graph (conv3d_1/2, the heads' last layers): forward / dgrad on v2x_conv2d, weight gradient = centre tap
    of v2x_conv3x3_wgrad, against F.linear autograd in fp32 on the same bf16-rounded operands."""
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(torch.bfloat16).float()
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.linear(xr, wr.view(Cout, Cin), br)
    yr.backward(dy.float())
    xd, wd, bd = x.to(device).requires_grad_(True), w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    yd = hip_graph.conv1x1(xd, wd, bd, f32_out=f32_out)
    assert yd.dtype == (torch.float32 if f32_out else torch.bfloat16)
    yd.backward(dy.to(device).float() if f32_out else dy.to(device))
    tol = 1e-5 if f32_out else 2.0 ** -7
    assert float((yd.detach().cpu().float() - yr.detach()).abs().max()) <= tol * float(yr.abs().max()) + 1e-5
    assert float((xd.grad.cpu().float() - xr.grad).abs().max()) <= 2.0 ** -7 * float(xr.grad.abs().max())
    assert float((wd.grad.cpu() - wr.grad).abs().max()) <= 2e-5 * float(wr.grad.abs().max())
    assert torch.allclose(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-3)


def test_graphed_training_step_equals_eager(device, tune):
    """train/graph_step.py: the whole HIP-graph training step (forward, loss, backward, optimizer) captured as one hipGraph.  Five replays
    on five different batches reproduce five eager steps from the same start: losses to 1e-6 and final parameters to 1e-6 of each
    tensor's scale (measured: identical digits -- a FaFNet step on the HIP graph has no atomics and no library-chosen algorithm in it,
    and the constructor's warm-up steps leave no trace).  Then a capturable Adam whose device-tensor learning rate changes after capture."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    cfg = Config("train")
    base = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=1).to(device)
    batches = [synthetic_batch_on_device(cfg, 1, 2, seed=10 + i, device=device) for i in range(5)]
    eager = copy.deepcopy(base).train()
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-3)
    losses_e = []
    for d in batches:
        res = train_forward(eager, d["bev_seq"], d["trans_matrices"], d["num_agent"], 1)
        loss = detection_loss(res, d["labels"], d["reg_targets"], d["reg_loss_mask"])[0]
        opt_e.zero_grad(set_to_none=True)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    graphed = copy.deepcopy(base).train()
    opt_g = torch.optim.SGD(graphed.parameters(), lr=1e-3)
    step = GraphedTrainStep(graphed, opt_g, batches[0], 1)
    losses_g = [float(step(d)[0]) for d in batches]
    print("eager  ", ["%.5f" % v for v in losses_e])
    print("graphed", ["%.5f" % v for v in losses_g])
    assert np.allclose(losses_g, losses_e, rtol=1e-6)
    for (k, pe), (_, pg) in zip(eager.state_dict().items(), graphed.state_dict().items()):
        if "num_batches_tracked" in k:
            assert int(pe) == int(pg) == 5, k
            continue
        assert float((pe.float() - pg.float()).abs().max()) <= 1e-6 * max(float(pe.float().abs().max()), 1e-3), k
    # capturable Adam with a device-tensor learning rate that changes after capture
    adam_model = copy.deepcopy(base).train()
    opt_a = torch.optim.Adam(adam_model.parameters(), lr=torch.tensor(1e-3, device=device), capturable=True)
    astep = GraphedTrainStep(adam_model, opt_a, batches[0], 1)
    first = float(astep(batches[0])[0])
    astep.set_lr(3e-4)
    for i in range(9):
        last = float(astep(batches[i % 5])[0])
    print("graphed Adam: loss %.4f -> %.4f in 10 replays" % (first, last))
    assert abs(first - losses_e[0]) <= 1e-5 * abs(losses_e[0]) and np.isfinite(last) and last < first
    import time
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        step(batches[0])
    torch.cuda.synchronize()
    print("graphed step (2 maps): %.2f ms" % ((time.time() - t0) / 20 * 1e3))


def test_graphed_step_keeps_optimizer_state_and_survives_epochs(device, tune):
    """the captured step is built on an optimizer that ALREADY carries moments and step counters (a later epoch, the partial last
    batch, --resume).  The constructor's warm-up steps must leave that state as they found it: 3 eager Adam steps + 3 graphed ones ==
    6 eager ones (losses and parameters), and the run's loops -- which make a new FaFModule per call around the run's optimizer --
    reuse the captured step instead of re-capturing (and re-warming) it every epoch."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, synthetic_batch_on_device, train_synthetic
    tune("TRAIN_HIP", 1)
    cfg = Config("train")
    base = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=5).to(device)
    batches = [synthetic_batch_on_device(cfg, 1, 2, seed=30 + i, device=device) for i in range(6)]

    def eager_step(model, opt, d):
        res = train_forward(model, d["bev_seq"], d["trans_matrices"], d["num_agent"], 1)
        loss = detection_loss(res, d["labels"], d["reg_targets"], d["reg_loss_mask"])[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss.detach())

    from v2x_sim_amd.train.optim import use_hip_adam
    # (both optimizers on the library's step from the start: GraphedTrainStep switches a plain torch.optim.Adam anyway, and torch's own step differs from it
    #  by fp32 rounding, which the bf16 weight packing turns into 1e-4-level loss differences within two steps)
    ref = copy.deepcopy(base).train()
    opt_r = use_hip_adam(torch.optim.Adam(ref.parameters(), lr=torch.tensor(1e-3, device=device), capturable=True))
    losses_r = [eager_step(ref, opt_r, d) for d in batches]
    mixed = copy.deepcopy(base).train()
    opt_m = use_hip_adam(torch.optim.Adam(mixed.parameters(), lr=torch.tensor(1e-3, device=device), capturable=True))
    losses_m = [eager_step(mixed, opt_m, d) for d in batches[:3]]
    p0 = next(iter(opt_m.state))
    before = {k: v.clone() for k, v in opt_m.state[p0].items() if torch.is_tensor(v)}
    assert float(before["step"]) == 3 and float(before["exp_avg_sq"].abs().max()) > 0
    step = GraphedTrainStep(mixed, opt_m, batches[3], 1)                 # warm-up runs 3 real steps, then must restore
    for k, v in before.items():
        assert torch.equal(opt_m.state[p0][k], v), "the warm-up changed the optimizer's %s" % k
    losses_m += [float(step(d)[0]) for d in batches[3:]]
    print("6 eager Adam steps  ", ["%.5f" % v for v in losses_r])
    print("3 eager + 3 graphed ", ["%.5f" % v for v in losses_m])
    assert np.allclose(losses_m, losses_r, rtol=2e-5)
    assert float(opt_m.state[p0]["step"]) == 6
    for (k, pe), (_, pg) in zip(ref.state_dict().items(), mixed.state_dict().items()):
        if "num_batches_tracked" not in k:
            assert float((pe.float() - pg.float()).abs().max()) <= 2e-5 * max(float(pe.float().abs().max()), 1e-3), k
    # two "epochs" through the loop API: one optimizer, a new FaFModule per call, ONE captured step
    tune("TRAIN_GRAPH", 1)
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=6).to(device)
    opt, sched = make_optimizer(model, 1e-3, 8)
    train_synthetic(model, cfg, 4, frames_per_step=1, seed=7, device=device, agents=2, opt=opt, sched=sched)
    cache = opt.__dict__["_v2x_graphed_steps"]
    assert len(cache) == 1
    first = next(iter(cache.values()))
    train_synthetic(model, cfg, 4, frames_per_step=1, seed=8, device=device, agents=2, opt=opt, sched=sched)
    assert len(cache) == 1 and next(iter(cache.values())) is first
    p1 = next(iter(opt.state))
    assert float(opt.state[p1]["step"]) == 8                               # 8 real steps, no warm-up step counted, none lost


def test_fafmodule_step_graphed_with_scheduler(device, tune):
    """V2X_TRAIN_HIP=1 V2X_TRAIN_GRAPH=1: FaFModule.step replays the captured step; make_optimizer's device-tensor learning rate is updated
    IN PLACE by the scheduler (the captured Adam reads the same tensor), and training on synthetic scenes reduces the loss."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, train_synthetic
    tune("TRAIN_HIP", 1)
    tune("TRAIN_GRAPH", 1)
    cfg = Config("train")
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=2).to(device)
    opt, sched = make_optimizer(model, 1e-3, 40)
    lr_tensor = opt.param_groups[0]["lr"]
    assert torch.is_tensor(lr_tensor) and lr_tensor.is_cuda
    hist = train_synthetic(model, cfg, 40, frames_per_step=1, seed=3, device=device, agents=2, opt=opt, sched=sched)
    assert opt.param_groups[0]["lr"] is lr_tensor and abs(float(lr_tensor) - 1e-3 * 0.09) < 1e-9      # two decays of x0.3, in place
    first, last = np.mean([h[0] for h in hist[:5]]), np.mean([h[0] for h in hist[-5:]])
    print("graphed FaFModule.step: mean loss of the first / last 5 of 40 steps: %.4f / %.4f" % (first, last))
    assert np.isfinite(last) and last < 0.8 * first
    # the inference engine must see the weights the replays produced (they do not bump the parameters' version counters): a model that
    # was used for inference BEFORE training, then trained by replays, must answer like a fresh copy holding the same state_dict
    from v2x_sim_amd.train.loop import synthetic_batch_on_device
    data = synthetic_batch_on_device(cfg, 1, 2, seed=41, device=device, with_targets=False)
    model2 = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=2).to(device).eval()
    with torch.no_grad():
        before = model2(data["bev_seq"], batch_size=1)["cls"].clone()           # packs the initial weights
    opt2, _ = make_optimizer(model2, 1e-3, 10)
    train_synthetic(model2, cfg, 10, frames_per_step=1, seed=4, device=device, agents=2, opt=opt2, sched=None)   # ends with model.eval()
    fresh = FaFNet(cfg, kd_flag=0, num_agent=2).to(device).eval()
    fresh.load_state_dict(model2.state_dict())
    with torch.no_grad():
        after = model2(data["bev_seq"], batch_size=1)["cls"]
        ref = fresh(data["bev_seq"], batch_size=1)["cls"]
    assert torch.equal(after, ref) and not torch.equal(after, before)


@pytest.mark.parametrize("family", ["when2com", "max", "cat", "disco", "v2v_seg", "faf_seg"])
def test_hip_graph_other_baselines_loss_parity(device, family, tune):
    """V2X_TRAIN_HIP=1 for the other detection baselines and the segmentation variants: encoder / decoder / heads on the kernels, their
    cross-agent fusion on the fp32 graph at the fusion layer.  One training-mode step: loss within 2 % of the all-fp32 graph's, finite
    gradients for the same set of parameters, and 15 Adam steps reduce the loss."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models import det as D
    from v2x_sim_amd.models import seg as S
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    cfg = Config("train")
    A = 3
    seg = family.endswith("_seg")
    if seg:
        cls = {"v2v_seg": S.V2VNetSeg, "faf_seg": S.FaFNetSeg}[family]
        base = cls(cfg, num_agent=A) if family == "v2v_seg" else cls(cfg)
        from v2x_sim_amd.utils.synthetic import init_synthetic_weights
        base = init_synthetic_weights(base, seed=2).to(device)
    else:
        cls = {"when2com": D.When2com, "max": D.MaxFusion, "cat": D.CatFusion, "disco": D.DiscoNet}[family]
        base = init_for_training(cls(cfg, num_agent=A), seed=2).to(device)
    data = synthetic_batch_on_device(cfg, 1, A, seed=8, device=device)
    if seg:
        g = torch.Generator().manual_seed(3)
        target = torch.randint(0, 8, (A, 256, 256), generator=g).to(device)

    def loss_of(model):
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        if seg:
            return torch.nn.functional.cross_entropy(res.reshape(-1, res.shape[-1]), target.reshape(-1))
        return detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
    out = {}
    for flag in ("0", "1"):
        tune("TRAIN_HIP", flag)
        model = copy.deepcopy(base).train()
        loss = loss_of(model)
        loss.backward()
        out[flag] = (float(loss.detach()), {k for k, p in model.named_parameters() if p.grad is not None})
        for k, p in model.named_parameters():
            assert p.grad is None or bool(torch.isfinite(p.grad).all()), k
    print("%s: loss fp32 graph %.5f, HIP graph %.5f" % (family, out["0"][0], out["1"][0]))
    assert abs(out["1"][0] - out["0"][0]) <= 2e-2 * abs(out["0"][0])
    assert out["0"][1] == out["1"][1]
    tune("TRAIN_HIP", 1)
    model = copy.deepcopy(base).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    first = None
    for _ in range(15):
        loss = loss_of(model)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
    tune.reset("TRAIN_HIP")
    print("%s: 15 Adam steps on the HIP graph %.4f -> %.4f" % (family, first, float(loss.detach())))
    assert float(loss.detach()) < first


def test_graphed_v2vnet_step(device, tune):
    """The captured step for V2VNet (fixed agent table): the frame plan's index tensors come from the warm-up calls' cache, so the capture
    contains no host -> device copy.  Five replays track five eager steps (the fp32 fusion's index_add uses atomics: 1e-3, not bitwise),
    a batch with another agent table is refused, and FaFModule.step falls back to eager for it."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    cfg = Config("train")
    base = init_for_training(V2VNet(cfg, num_agent=3), seed=1).to(device)
    batches = [synthetic_batch_on_device(cfg, 1, 3, seed=20 + i, device=device) for i in range(5)]
    eager = copy.deepcopy(base).train()
    opt_e = torch.optim.SGD(eager.parameters(), lr=1e-3)
    losses_e = []
    for d in batches:
        res = train_forward(eager, d["bev_seq"], d["trans_matrices"], d["num_agent"], 1)
        loss = detection_loss(res, d["labels"], d["reg_targets"], d["reg_loss_mask"])[0]
        opt_e.zero_grad(set_to_none=True)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    graphed = copy.deepcopy(base).train()
    step = GraphedTrainStep(graphed, torch.optim.SGD(graphed.parameters(), lr=1e-3), batches[0], 1)
    losses_g = [float(step(d)[0]) for d in batches]
    print("eager  ", ["%.5f" % v for v in losses_e])
    print("graphed", ["%.5f" % v for v in losses_g])
    assert np.allclose(losses_g, losses_e, rtol=2e-3)
    other = dict(batches[0])
    other["num_agent"] = batches[0]["num_agent"].clone()
    other["num_agent"][0, :] = 2
    with pytest.raises(ValueError):
        step(other)


@pytest.mark.parametrize("cout,cin,stride,cin_pad", [(32, 13, 1, 32), (32, 32, 1, None), (64, 64, 1, None), (128, 64, 2, None), (128, 384, 1, None),
                                                      (32, 96, 1, None), (512, 256, 2, None), (64, 32, 2, None)])
def test_pack_conv_device_equals_torch_packing(device, cout, cin, stride, cin_pad):
    """v2x_pack_conv_device (one launch per layer, row f-3) against the torch-op packing of packing.layer_conv_bn it replaces in the training
    graph: the same bytes, for the forward layer and for the data-gradient layer (flipped, transposed weights), in every layout the backbone's
    layers take (halo, streamed stride 1 / 2, gather)."""
    import types
    from v2x_sim_amd import packing
    g = torch.Generator().manual_seed(cout + cin)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(device)
    b = torch.randn(cout, generator=g).to(device)

    def ref_layer(weight, bias, s, cp):
        conv = types.SimpleNamespace(weight=weight, bias=bias, kernel_size=(3, 3), stride=(s, s), padding=(1, 1), out_channels=weight.shape[0])
        with packing.on_device(device):
            return packing.layer_conv_bn("ref", conv, None, device=device, relu=False, cin_pad=cp)
    for dgrad in (False, True):
        if dgrad and cin_pad:
            continue                               # (the first layer needs no data gradient)
        got = packing.pack_conv_device("t", w, b, stride=stride, cin_pad=cin_pad, dgrad=dgrad)
        ref = ref_layer(w.flip(2, 3).transpose(0, 1).contiguous(), None, 1, None) if dgrad else ref_layer(w, b, stride, cin_pad)
        gp = got.halo if got.halo is not None else got.fallback[0]
        rp = ref.halo if got.halo is not None else ref.fallback[0]
        assert rp is not None and (gp.w_layout or 0) == (rp.w_layout or 0), (dgrad, gp.w_layout, None if rp is None else rp.w_layout)
        assert gp.w_rows == rp.w_rows and gp.w_kpad == rp.w_kpad and (gp.C0, gp.C1, gp.Cout, gp.stride) == (rp.C0, rp.C1, rp.Cout, rp.stride)
        assert torch.equal(gp.weight.view(torch.int16).flatten(), rp.weight.view(torch.int16).flatten()), dgrad
        assert torch.equal(gp.scale, rp.scale) and torch.equal(gp.shift, rp.shift)


@pytest.mark.parametrize("shape", [(10, 256, 256, 32), (3, 64, 64, 128), (2, 16, 16, 512), (1, 8, 32, 64), (5, 7, 3, 256)])
def test_channel_sum_vs_torch(device, shape):
    """v2x_channel_sum_bf16 (the convolutions' bias gradient): against torch's fp64 sum of the same bf16 values (fp32-accumulation error
    only), and bit-reproducible."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) + 0.3).to(torch.bfloat16).to(device)
    got = ops.channel_sum(x)
    ref = x.double().reshape(-1, shape[-1]).sum(0)
    scale = float(x.double().abs().reshape(-1, shape[-1]).sum(0).max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale, float((got.double() - ref).abs().max())
    assert torch.equal(ops.channel_sum(x), got)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (1, 8, 32, 32, 32), (3, 24, 32, 96, 32), (2, 32, 32, 256, 128), (10, 128, 128, 64, 64)])
def test_wgrad_transpose_read_form_equals_first_form_bitwise(device, N, H, W, Cin, Cout, tune):
    """conv3x3_wgrad_tr_kernel (LDS-DMA tiles in their natural layout + ds_read_b64_tr_b16 fragments; default) against the first form (VALU
    transposes into LDS): the same products summed in the same order -> bit-identical partials and gradients; borders, ragged tile shares,
    both channel-tile forms (Cout % 64 == 0 and == 32)."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(N + H + Cin + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16).to(device)
    new = ops.conv3x3_wgrad(x, dy)
    tune("WGRAD_TR", 0)
    old = ops.conv3x3_wgrad(x, dy)
    tune.reset("WGRAD_TR")
    assert torch.equal(new, old), float((new - old).abs().max())
    assert torch.equal(ops.conv3x3_wgrad(x, dy), new)


def test_hip_engine_learns_under_a_fused_optimizer(device, tune):
    """torch.optim.Adam(fused=True) -- what train.loop.make_optimizer creates -- updates the parameters WITHOUT bumping their version
    counters, and the packed-weight caches of the HIP training graph used to key on those alone: every step then ran on the packed weights
    of step 0 and nothing but the BatchNorm parameters learned (loss 3.0 -> 2.8 in 250 steps against 0.13 on the fp32 graph).
    packing.watch_optimizer stamps the parameters after each step; 40 steps must bring the loss well under its initial value, the inference
    engine must see the trained weights, and a detector trained this way must not fire everywhere."""
    from v2x_sim_amd import packing
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, synthetic_batch_on_device, train_synthetic
    tune("TRAIN_HIP", 1)
    tune("TRAIN_GRAPH", 0)
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(FaFNet(cfg, num_agent=2), seed=0).to(device)
    opt, sched = make_optimizer(model, 1e-3, 60)
    assert opt.defaults.get("fused") and opt.__dict__.get("_v2x_watched")
    w = model.u_encoder.conv3_1.weight if hasattr(model, "u_encoder") else next(model.parameters())
    v0 = packing.param_version(w)
    hist = train_synthetic(model, cfg, 60, frames_per_step=1, lr=1e-3, seed=3, device=device, agents=2, opt=opt, sched=sched)
    assert packing.param_version(w) != v0
    first, last = np.mean([h[0] for h in hist[:5]]), np.mean([h[0] for h in hist[-10:]])
    print("fused Adam on the HIP graph: loss %.3f -> %.3f" % (first, last))
    assert last < 0.5 * first, (first, last)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=99, device=device, with_targets=False)
    model.eval()
    with torch.no_grad():
        cls = model(data["bev_seq"])["cls"].float()
    frac = float((torch.softmax(cls.reshape(-1, 2), -1)[:, 1] >= 0.7).float().mean())
    assert frac < 0.05, frac


@pytest.mark.parametrize("P,C,H,W", [(7, 32, 32, 32), (3, 20, 16, 48), (2, 256, 32, 32)])
def test_warp_affine_forward_and_transpose_vs_grid_sample(device, P, C, H, W):
    """v2x_warp_affine_f32 vs F.affine_grid + F.grid_sample (bilinear, zeros, align_corners=False) and v2x_warp_affine_bwd_f32 vs autograd's
    grid_sampler backward, for rotations, translations (the two steps of feature_transformation), a general affine map with shear / scale,
    the identity and a map that throws the whole image out of range; fp32 coordinate rounding only.  The backward is bit-reproducible."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(P * 1000 + C)
    x = torch.randn(P, C, H, W, generator=g).to(device).requires_grad_(True)
    ang = torch.rand(P, generator=g) * 6.283
    th = torch.zeros(P, 2, 3)
    th[:, 0, 0], th[:, 0, 1], th[:, 1, 0], th[:, 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)   # rotations
    th[0] = torch.tensor([[1.0, 0.0, 0.37], [0.0, 1.0, -0.81]])          # a translation
    th[1] = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])             # the identity: sample positions ON the pixel centres
    if P > 2:
        th[2] = torch.tensor([[0.7, 0.45, 0.1], [-0.3, 1.4, -0.2]])      # shear + anisotropic scale (|det| != 1: more candidates per pixel)
    if P > 3:
        th[3] = torch.tensor([[1.0, 0.0, 5.0], [0.0, 1.0, 0.0]])         # everything out of range -> zeros
    if P > 4:
        th[4] = torch.tensor([[0.25, 0.0, 0.0], [0.0, 0.25, 0.0]])       # zoom in x4 (16 output pixels per input pixel)
    if P > 5:
        th[5] = torch.tensor([[0.0, 0.0, 0.1], [0.0, 0.0, -0.2]])        # singular: every output pixel samples the same point
    th = th.to(device)
    ref = F.grid_sample(x, F.affine_grid(th, x.shape, align_corners=False), mode="bilinear", padding_mode="zeros", align_corners=False)
    got = ops.warp_affine(x.detach(), th)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), float((got - ref).abs().max())
    if P > 3:
        assert float(got[3].abs().max()) == 0.0
    dy = torch.randn(P, C, H, W, generator=g).to(device)
    gref, = torch.autograd.grad(ref, x, dy)
    ggot = ops.warp_affine(dy, th, backward=True)
    scale = float(gref.abs().max())
    assert torch.allclose(ggot, gref, atol=2e-5 * max(1.0, scale), rtol=1e-4), float((ggot - gref).abs().max())
    for _ in range(3):
        assert torch.equal(ops.warp_affine(dy, th, backward=True), ggot)
    # <dy, A x> == <A^T dy, x> (the transpose property itself, in fp64 sums)
    lhs = float((dy.double() * got.double()).sum())
    rhs = float((ggot.double() * x.detach().double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


def test_v2vnet_hip_graph_warp_on_kernels_matches_grid_sample_path(device, tune):
    """A V2VNet training step with the fusion stage's warp on v2x_warp_affine_f32 / _bwd_f32 against the same step with F.grid_sample and its
    atomic backward.  (a) On the fp32 graph (override installed by hand): same loss, gradients within 1e-2 of the whole gradient's norm -- the
    operator itself agrees to 2e-6 (test above); a randomly initialised deep net with batch-statistics BN amplifies that (measured 1.5e-3).
    (b) On the bf16 HIP graph (the default engine; WARP_HIP = 1 | 0): one-ulp bf16 flips amplify further (measured 8e-2 of the norm; the loss
    agrees to 1e-4), so only sanity bounds are asserted there -- and that the kernel path is bit-reproducible from run to run."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.train import detection_loss, graph, train_forward
    from v2x_sim_amd.train.hip_graph import _AffineSample
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(V2VNet(cfg, num_agent=3), seed=2).to(device).train()
    data = synthetic_batch_on_device(cfg, 1, 3, seed=5, device=device)

    def run():
        model.zero_grad(set_to_none=True)
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    def rel(ga, gb):
        den = sum(float(gb[n].double().pow(2).sum()) for n in gb) ** 0.5
        return sum(float((ga[n].double() - gb[n].double()).pow(2).sum()) for n in gb) ** 0.5 / den
    # (a) fp32 graph
    tune("TRAIN_HIP", 0)
    l0, g0 = run()
    graph._affine_sample_override = _AffineSample.apply
    try:
        l1, g1 = run()
    finally:
        graph._affine_sample_override = None
    print("fp32 graph, warp on the kernels vs grid_sample: loss %.6f vs %.6f, gradient difference %.2e of the norm" % (l1, l0, rel(g1, g0)))
    assert set(g1) == set(g0) and abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0)) and rel(g1, g0) < 1e-2
    # (b) HIP graph (the fp32 fusion stage: round 6's NHWC stage has its own tests below)
    tune("TRAIN_HIP", 1)
    tune("TRAIN_V2V_NHWC", 0)
    tune("WARP_HIP", 1)
    l1, g1 = run()
    l1b, g1b = run()
    tune("WARP_HIP", 0)
    l0, g0 = run()
    assert l1 == l1b and all(torch.equal(g1[n], g1b[n]) for n in g1)
    print("HIP graph, warp on the kernels vs grid_sample: loss %.6f vs %.6f, gradient difference %.2e of the norm" % (l1, l0, rel(g1, g0)))
    assert set(g1) == set(g0) and abs(l1 - l0) <= 2e-3 * max(1.0, abs(l0)) and rel(g1, g0) < 0.25


@pytest.mark.parametrize("M,C,relu", [(2 * 64 * 64, 32, True), (3 * 32 * 32, 128, True), (10 * 16 * 16, 512, False), (777, 64, True)])
def test_bn_backward_accumulates_the_conv_bias_gradient(device, M, C, relu):
    """v2x_bn_train_backward_dxsum: the same dx, dgamma, dbeta as v2x_bn_train_backward, bit for bit, plus the per-channel sum of dx AS STORED
    (what ops.channel_sum(dx) -- the bias gradient of the convolution in front of the BN -- would reduce in a second pass); fixed order."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(M, C, generator=g).to(torch.bfloat16).to(device)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(device), (torch.randn(C, generator=g) * 0.3).to(device)
    _, mean, invstd = ops.bn_train_forward(x, gamma, beta, None, None, 1e-5, 0.1, relu)
    dx0, dg0, db0 = ops.bn_train_backward(x, dy, gamma, beta, mean, invstd, relu)
    dx1, dg1, db1, dsum = ops.bn_train_backward(x, dy, gamma, beta, mean, invstd, relu, dx_sum=True)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    ref = dx1.double().sum(0)
    tol = 1e-6 * float(dx1.double().abs().sum(0).max()) + 1e-7
    assert float((dsum.double() - ref).abs().max()) <= tol, (float((dsum.double() - ref).abs().max()), tol)
    assert torch.allclose(dsum, ops.channel_sum(dx1), atol=tol, rtol=0)
    for _ in range(3):
        assert torch.equal(ops.bn_train_backward(x, dy, gamma, beta, mean, invstd, relu, dx_sum=True)[3], dsum)


def test_conv_bias_gradients_come_from_the_bn_backward_kernel(device, tune, monkeypatch):
    """In the HIP training graph every conv + BN pair takes its bias gradient from the BN backward kernel's accumulated sums (attached to the
    gradient tensor, hip_graph._attached_channel_sum) instead of a second pass over dx: on a FaFNet step ops.channel_sum is left with the
    layers whose gradient reaches them another way (the width-padded 16-pixel layer), and the bias gradients equal the reduced ones."""
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, hip_graph, train_forward
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    tune("TRAIN_BN_BIAS_ZERO", 0)       # (round 6's default returns these gradients as exact zeros: test_bias_gradient_in_front_of_a_batchnorm_is_zero)
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=4).to(device).train()
    data = synthetic_batch_on_device(cfg, 1, 2, seed=6, device=device)
    calls = []
    real = ops.channel_sum
    monkeypatch.setattr(ops, "channel_sum", lambda t: (calls.append(tuple(t.shape)), real(t))[1])

    def step():
        model.zero_grad(set_to_none=True)
        res = train_forward(model, data["bev_seq"], None, None, 1)
        detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0].backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if n.endswith("bias") and p.grad is not None}
    g1 = step()
    n_attached = len(calls)
    monkeypatch.setattr(hip_graph, "_attached_channel_sum", lambda dy: None)     # every layer reduces its own gradient
    calls.clear()
    g0 = step()
    print("ops.channel_sum calls per FaFNet step: %d with the BN kernel's sums, %d without" % (n_attached, len(calls)))
    assert n_attached <= 2 < len(calls) - 10
    for n in g0:
        scale = float(g0[n].abs().max())
        # (a bias in front of a batch-statistics BN has a zero true gradient: both values are cancelling sums of ~1e5 terms, equal up to summation order)
        assert torch.allclose(g1[n], g0[n], atol=1e-3 * max(scale, 1e-3), rtol=1e-4), (n, float((g1[n] - g0[n]).abs().max()), scale)


@pytest.mark.parametrize("N,H,W,C0,C1", [(2, 16, 16, 512, 256), (3, 8, 24, 64, 32), (1, 128, 128, 64, 32), (5, 5, 7, 8, 24)])
def test_upcat_forward_and_backward_equal_the_torch_ops_bitwise(device, N, H, W, C0, C1):
    """v2x_upcat_bf16 = cat(nearest x2 upsample, skip) (pure data movement: identical bits); v2x_upcat_bwd_bf16 = autograd's backward of the torch
    expression on bf16 tensors (slice of the skip part; the 2x2 sum of four bf16 values in fp32 rounded once -- torch's bf16 sum does the same,
    and with four addends any fp32 order that differs could change the last bit, so the comparison allows one bf16 ulp and checks that almost
    every element is identical)."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(N * H + C0)
    lo = torch.randn(N, H, W, C0, generator=g).to(torch.bfloat16).to(device).requires_grad_(True)
    skip = torch.randn(N, 2 * H, 2 * W, C1, generator=g).to(torch.bfloat16).to(device).requires_grad_(True)
    up = lo[:, :, None, :, None, :].expand(N, H, 2, W, 2, C0).reshape(N, 2 * H, 2 * W, C0)
    ref = torch.cat((up, skip), dim=3)
    got = ops.upcat(lo.detach(), skip.detach())
    assert torch.equal(got, ref)
    dcat = torch.randn(N, 2 * H, 2 * W, C0 + C1, generator=g).to(torch.bfloat16).to(device)
    rlo, rskip = torch.autograd.grad(ref, (lo, skip), dcat)
    dlo, dskip = ops.upcat_backward(dcat, C0)
    assert torch.equal(dskip, rskip)
    exact = dcat[..., :C0].float().reshape(N, H, 2, W, 2, C0).sum((2, 4))
    assert torch.equal(dlo, (dcat[:, 0::2, 0::2, :C0].float() + dcat[:, 0::2, 1::2, :C0].float() + dcat[:, 1::2, 0::2, :C0].float()
                             + dcat[:, 1::2, 1::2, :C0].float()).to(torch.bfloat16))          # the kernel's own order, bit for bit
    assert float((dlo.float() - exact).abs().max()) <= 2 ** -7 * float(exact.abs().max())
    assert float((dlo != rlo).float().mean()) < 0.02 and torch.allclose(dlo.float(), rlo.float(), atol=2 ** -6 * float(exact.abs().max()), rtol=2 ** -7)


@pytest.mark.parametrize("cout,cin,f32_out", [(64, 64, False), (128, 128, False), (12, 32, True), (36, 32, True), (8, 64, True)])
def test_pack_conv1x1_device_equals_torch_packing(device, cout, cin, f32_out):
    """packing.pack_conv1x1_device (one launch) against the torch-op packing the training graph used for its 1x1 layers: the same bytes and
    PackedConv fields, forward layer and data-gradient layer (transposed weights, the gradient's channels padded to a multiple of 32)."""
    from v2x_sim_amd import packing
    from v2x_sim_amd._lib import V2X_EPI_BF16, V2X_EPI_F32
    g = torch.Generator().manual_seed(cout * 7 + cin)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).to(device)
    b = torch.randn(cout, generator=g).to(device)
    cp = (cout + 31) // 32 * 32
    w2 = w.reshape(cout, cin)
    with packing.on_device(device):
        sc, sh = packing.fold_bn(b, None, cout)
        ref_f = packing.pack_conv("r", w2[:, :, None, None], sc, sh, stride=1, pad=0, relu=False, epilogue=V2X_EPI_F32 if f32_out else V2X_EPI_BF16, device=device)
        sc, sh = packing.fold_bn(None, None, cin)
        ref_d = packing.pack_conv("r", F.pad(w2.t(), (0, cp - cout))[:, :, None, None], sc, sh, stride=1, pad=0, relu=False, epilogue=V2X_EPI_BF16, device=device)
    got_f = packing.pack_conv1x1_device("t", w, b, f32_out=f32_out)
    got_d = packing.pack_conv1x1_device("t", w, None, dgrad=True, cout_pad=cp)
    for got, ref in ((got_f, ref_f), (got_d, ref_d)):
        assert (got.C0, got.C1, got.Cout, got.ksize, got.stride, got.pad, got.epilogue, got.relu, got.w_rows, got.w_kpad) == \
               (ref.C0, ref.C1, ref.Cout, ref.ksize, ref.stride, ref.pad, ref.epilogue, ref.relu, ref.w_rows, ref.w_kpad)
        assert torch.equal(got.weight.view(torch.int16).flatten(), ref.weight.view(torch.int16).flatten())
        assert torch.equal(got.scale, ref.scale) and torch.equal(got.shift, ref.shift)


def test_eager_step_after_graph_replays_sees_the_replayed_weights(device, tune, monkeypatch):
    """hipGraph replays update the parameters without touching Tensor._version.  An EAGER HIP-graph forward between / after replays (another batch
    shape, the partial last batch of an epoch) must re-pack: its loss equals that of a fresh copy of the model (new parameter objects, nothing
    cached).  Negative control: with the stamping of GraphedTrainStep.__call__ disabled the eager forward really does run on stale packed weights."""
    import copy
    from v2x_sim_amd import packing
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    tune("TRAIN_GRAPH", 1)
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=1).to(device).train()
    opt, _ = make_optimizer(model, 1e-2, 100)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=3, device=device)

    def eager_loss(m):
        with torch.no_grad():
            res = train_forward(m, data["bev_seq"], None, None, 1)
            return float(detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0])
    step = GraphedTrainStep(model, opt, data, 1)
    for _ in range(2):
        step(data)
    l_mid = eager_loss(model)                          # fills the eager packing cache with the weights after two replays
    real = packing.note_params_changed
    monkeypatch.setattr(packing, "note_params_changed", lambda params: None)
    for _ in range(3):
        step(data)                                     # replays WITHOUT the stamp
    torch.cuda.synchronize()
    l_stale = eager_loss(model)
    l_fresh_then = eager_loss(copy.deepcopy(model))
    monkeypatch.setattr(packing, "note_params_changed", real)
    step(data)                                         # one more replay, stamped
    torch.cuda.synchronize()
    l_after = eager_loss(model)
    l_fresh = eager_loss(copy.deepcopy(model))
    print("eager loss after 2 replays %.5f; after 3 more, unstamped: %.5f (a fresh copy of those weights: %.5f); after a stamped replay %.5f (fresh copy %.5f)"
          % (l_mid, l_stale, l_fresh_then, l_after, l_fresh))
    assert l_after == l_fresh
    assert l_stale != l_fresh_then            # the control: without the stamp the convolutions ran on the packings of two replays ago


@pytest.mark.parametrize("family", ["FaFNet", "V2VNet"])
def test_batched_repack_changes_no_bit(device, tune, family):
    """TRAIN_PACK_BATCH (all stale packings rebuilt in place by ONE launch at the start of a step) only changes in HOW MANY launches the same
    packing runs: four SGD steps from the same start give bit-identical losses and parameters with it on and off."""
    import copy
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet, V2VNet
    from v2x_sim_amd.train import detection_loss, train_forward, hip_graph
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    cfg = Config("train")
    cls, kw = (FaFNet, dict(kd_flag=0, num_agent=2)) if family == "FaFNet" else (V2VNet, dict(num_agent=2))
    base = init_for_training(cls(cfg, **kw), seed=3).to(device)
    batches = [synthetic_batch_on_device(cfg, 1, 2, seed=20 + i, device=device) for i in range(4)]

    def run(batch):
        tune("TRAIN_PACK_BATCH", batch)
        hip_graph._CACHE.clear()
        hip_graph._PLANS.clear()
        m = copy.deepcopy(base).train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9)
        losses = []
        for d in batches:
            res = train_forward(m, d["bev_seq"], d["trans_matrices"], d["num_agent"], 1)
            loss = detection_loss(res, d["labels"], d["reg_targets"], d["reg_loss_mask"])[0]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return torch.stack(losses), {k: v.detach().clone() for k, v in m.state_dict().items()}, len(hip_graph._PLANS)

    ref_l, ref_p, n_plans = run(0)
    assert n_plans == 0
    l, p, n_plans = run(1)
    assert n_plans >= 1, "the batched re-pack must really have run"
    assert torch.equal(l, ref_l), (l, ref_l)
    for k in ref_p:
        assert torch.equal(p[k], ref_p[k]), k


def test_discarded_models_leave_no_packings_behind(device, tune):
    """ the training-graph cache holds its packed buffers strongly and the parameters weakly: when a model is dropped, the next
    train_forward prunes its cache entries and every re-pack plan that names them -- the device memory of a discarded model is released."""
    import gc
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward, hip_graph
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    tune("TRAIN_PACK_BATCH", 1)
    cfg = Config("train")
    hip_graph._CACHE.clear()
    hip_graph._PLANS.clear()
    d = synthetic_batch_on_device(cfg, 1, 2, seed=31, device=device)

    def steps(m, n):
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
        for _ in range(n):
            res = train_forward(m, d["bev_seq"], d["trans_matrices"], d["num_agent"], 1)
            loss = detection_loss(res, d["labels"], d["reg_targets"], d["reg_loss_mask"])[0]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

    first = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=1).to(device).train()
    steps(first, 3)
    n_first, plans_first = len(hip_graph._CACHE), len(hip_graph._PLANS)
    assert n_first > 20 and plans_first >= 1
    del first
    gc.collect()
    second = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=2).to(device).train()
    steps(second, 3)
    torch.cuda.synchronize()
    alive = [k for k, ent in hip_graph._CACHE.items() if ent[2]() is not None]
    assert len(hip_graph._CACHE) == len(alive) == n_first, (len(hip_graph._CACHE), len(alive), n_first)    # the first model's entries are gone
    assert all(set(pk) <= set(hip_graph._CACHE) for pk in hip_graph._PLANS) and len(hip_graph._PLANS) <= plans_first + 1


def test_repack_plan_equals_the_per_layer_device_packer(device):
    """packing.RepackPlan (v2x_pack_conv_device_job / _batch): every layout the training graph packs (gather, halo, streamed; forward and
    data-gradient; 3x3 and 1x1; a padded bias), rebuilt in place after the parameters changed == a fresh per-layer packing, bit for bit."""
    from v2x_sim_amd import packing
    g = torch.Generator().manual_seed(5)
    specs = [(32, 13, 1, 32, False), (64, 32, 2, None, False), (128, 64, 1, None, False), (256, 128, 2, None, False), (512, 256, 1, None, False),
             (64, 128, 1, None, True), (32, 64, 1, None, True), (256, 512, 1, None, True)]
    params, packs = [], []
    for cout, cin, stride, cin_pad, dgrad in specs:
        co_w, ci_w = (cin, cout) if dgrad else (cout, cin)
        w = torch.randn(co_w, ci_w, 3, 3, generator=g).to(device)
        b = None if dgrad else torch.randn(cout, generator=g).to(device)
        layer = packing.pack_conv_device("t", w, b, stride=stride, cin_pad=cin_pad, dgrad=dgrad)
        params.append((w, b, dict(stride=stride, cin_pad=cin_pad, dgrad=dgrad), "3x3"))
        packs += ([layer.halo] if layer.halo is not None else []) + list(layer.fallback)
    for cout, cin, dgrad, f32 in ((12, 32, False, True), (36, 32, False, True), (64, 64, False, False), (64, 64, True, False), (12, 32, True, False)):
        w = torch.randn(cout, cin, 1, 1, generator=g).to(device)
        b = None if dgrad else torch.randn(cout, generator=g).to(device)
        kw = dict(dgrad=dgrad, cout_pad=32 if dgrad else 0, f32_out=f32)
        packs.append(packing.pack_conv1x1_device("t1", w, b, **kw))
        params.append((w, b, kw, "1x1"))
    plan = packing.RepackPlan(packs)
    assert plan.valid() and plan.n_jobs == len(packs)
    with torch.no_grad():
        for w, b, _, _ in params:
            w.mul_(-0.5).add_(0.125)
            if b is not None:
                b.add_(1.0)
    plan.launch()
    torch.cuda.synchronize()
    for (w, b, kw, kind), pc in zip(params, packs):
        if kind == "3x3":
            layer = packing.pack_conv_device("t", w, b, **kw)
            fresh = layer.halo if layer.halo is not None else layer.fallback[0]
        else:
            fresh = packing.pack_conv1x1_device("t1", w, b, **kw)
        assert torch.equal(pc.weight.view(-1).view(torch.int16), fresh.weight.view(-1).view(torch.int16)), (kind, kw)
        assert torch.equal(pc.shift, fresh.shift) and torch.equal(pc.scale, fresh.scale), (kind, kw)
    # a parameter that moved invalidates the plan
    params[0][0].data = params[0][0].data.clone()
    assert not plan.valid()


@pytest.mark.parametrize("normalizer", ["positives", "batch"])
@pytest.mark.parametrize("case", ["scene", "no_positive", "all_masked_in", "soft_and_empty_labels"])
def test_fused_detection_loss_equals_the_torch_ops(device, tune, case, normalizer):
    """csrc/det_loss.hip (v2x_det_loss_forward / _backward behind train/loss.py::detection_loss) against the PyTorch-op specification in the same
    file: the three losses and both gradients, for a synthetic scene's targets, a batch without a positive anchor (n clamps to 1), every anchor
    selected by the regression mask, and label pairs that are not one-hot (0, 0) / (0.3, 0.7) -- the gradient formulas hold for any pair.
    Also the incoming gradients of the two partial losses (a caller that logs or weights them) and bit-reproducibility.  normalizer = "batch":
    the second reading of oracle/ASSUMPTIONS.md row 49 (Config.loss_normalizer) -- the kernels' sums rescaled by n_pos / maps on the device."""
    from v2x_sim_amd.train.loss import detection_loss
    g = torch.Generator().manual_seed(7)
    N, X, Y, A = 3, 32, 64, 6
    cls = (torch.randn(N, X * Y * A, 2, generator=g) * 3.0).to(device).requires_grad_(True)
    loc = (torch.randn(N, X, Y, A, 1, 6, generator=g) * 0.5).to(device).requires_grad_(True)
    lab = torch.zeros(N, X, Y, A, 2)
    pos = torch.rand(N, X, Y, A, generator=g) < 0.02
    lab[..., 1] = pos.float()
    lab[..., 0] = 1.0 - lab[..., 1]
    tgt = torch.where(pos[..., None, None], torch.randn(N, X, Y, A, 1, 6, generator=g) * 0.4, torch.zeros(()))
    mask = pos[..., None].clone()
    if case == "no_positive":
        lab[..., 1], lab[..., 0] = 0.0, 1.0
        mask[:] = False
    elif case == "all_masked_in":
        mask[:] = True
        tgt = torch.randn(N, X, Y, A, 1, 6, generator=g) * 0.4
    elif case == "soft_and_empty_labels":
        lab[0, :8] = 0.0
        lab[1, :8, :, :, 0], lab[1, :8, :, :, 1] = 0.3, 0.7
    lab, tgt, mask = lab.to(device), tgt.to(device), mask.to(device)
    w = torch.tensor([1.0, 0.25, -0.5], device=device)

    def run(flag):
        tune("TRAIN_HIP", 1)
        tune("TRAIN_LOSS_HIP", flag)
        cls.grad = loc.grad = None
        out = detection_loss({"cls": cls, "loc": loc}, lab, tgt, mask, normalizer=normalizer)
        (out[0] * w[0] + out[1] * w[1] + out[2] * w[2]).backward()
        return [o.detach().clone() for o in out], cls.grad.clone(), loc.grad.clone()

    ref, ref_dc, ref_dl = run(0)
    got, dc, dl = run(1)
    for a, b in zip(got, ref):
        assert abs(float(a) - float(b)) <= 2e-6 * max(abs(float(b)), 1e-3), (case, float(a), float(b))
    for a, b in ((dc, ref_dc), (dl, ref_dl)):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-12) + 1e-9, (case, float((a - b).abs().max()), float(b.abs().max()))
    if case == "no_positive":
        assert float(got[2]) == 0.0 and float(dl.abs().max()) == 0.0
    again, dc2, dl2 = run(1)
    assert all(torch.equal(a, b) for a, b in zip(again, got)) and torch.equal(dc, dc2) and torch.equal(dl, dl2), "fixed-order sums: bit-reproducible"
    # only `loss` used (the training loop): the partial losses' gradients arrive as None
    cls.grad = loc.grad = None
    detection_loss({"cls": cls, "loc": loc}, lab, tgt, mask, normalizer=normalizer)[0].backward()
    tune("TRAIN_LOSS_HIP", 0)
    g1, g2 = cls.grad.clone(), loc.grad.clone()
    cls.grad = loc.grad = None
    detection_loss({"cls": cls, "loc": loc}, lab, tgt, mask, normalizer=normalizer)[0].backward()
    assert float((g1 - cls.grad).abs().max()) <= 2e-6 * float(cls.grad.abs().max()) + 1e-9
    assert float((g2 - loc.grad).abs().max()) <= 2e-6 * max(float(loc.grad.abs().max()), 1e-12) + 1e-9


@pytest.mark.parametrize("shape", [(3, 16, 32, 64), (1, 8, 8, 8), (2, 5, 7, 136)])
def test_zero_insert_equals_the_torch_ops_bitwise(device, shape):
    """v2x_zero_insert_bf16 (the operand of a stride-2 layer's gradients) == torch.zeros + strided copy."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    dy = torch.randn(*shape, generator=g).to(torch.bfloat16).to(device)
    N, Ho, Wo, C = shape
    ref = torch.zeros((N, 2 * Ho, 2 * Wo, C), dtype=torch.bfloat16, device=device)
    ref[:, ::2, ::2] = dy
    assert torch.equal(ops.zero_insert(dy), ref)


@pytest.mark.parametrize("shape", [(10, 256, 32, 32), (3, 16, 8, 4), (1, 64, 16, 16)])
def test_fused_gru_gates_equal_the_torch_ops(device, tune, shape):
    """csrc/gru_train.hip (v2x_gru_gates_f32 / _bwd_f32 behind train/graph.py::_gru_step) against the PyTorch ops of the same function: h, the
    gradient of the pre-activations and the gradient of bias_hh (its n part needs d pre_n * r summed per channel), to fp32 rounding; large
    pre-activations saturate the gates without NaNs."""
    import types
    from v2x_sim_amd.train import graph
    P, C, H, W = shape
    g = torch.Generator().manual_seed(P + C)
    gi0 = (torch.randn(P, 3 * C, H, W, generator=g) * 2.0)
    gi0[0, :, 0, 0] = 60.0
    gi0[0, :, 0, 1] = -60.0
    bhh = (torch.randn(3 * C, generator=g) * 0.5).to(device).requires_grad_(True)
    dh = torch.randn(P, C, H, W, generator=g).to(device)
    cell = types.SimpleNamespace(weight_ih_l0=None, bias_ih_l0=None, bias_hh_l0=bhh, kernel_size=3)

    def run(flag):
        tune("TRAIN_GATES_HIP", flag)
        gi = gi0.to(device).requires_grad_(True)
        bhh.grad = None
        h = graph._gru_step(cell, None, conv=lambda x, w, b: gi)
        h.backward(dh)
        return h.detach(), gi.grad.clone(), bhh.grad.clone()

    ref = run(0)
    got = run(1)
    for name, a, b in zip(("h", "dgi", "dbias_hh"), got, ref):
        assert torch.isfinite(a).all(), name
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 3e-6 * scale + 1e-7, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("M_shape,Cin,Cout,f32_out,relu", [((2, 64, 64), 64, 64, False, True), ((1, 64, 64), 32, 12, True, False), ((2, 32, 32), 32, 36, True, False),
                                                             ((1, 32, 64), 128, 128, False, True), ((3, 24, 40), 64, 32, False, False), ((1, 5, 7), 32, 32, False, True),
                                                             ((10, 256, 256), 32, 12, True, False), ((4, 128, 128), 64, 64, False, True), ((2, 16, 16), 128, 64, False, False)])
def test_streaming_conv1x1_equals_the_gather_kernel_bitwise(device, tune, M_shape, Cin, Cout, f32_out, relu):
    """conv1x1.hip (round 6: 1x1 layers as a register-resident-weights stream, no LDS) against the gather kernel it replaces for these shapes (switch CONV1X1 = 0):
    the same K order, one MFMA per 32-channel chunk, the same epilogue arithmetic -> the same bits, for bf16 and fp32 outputs, Cout not a multiple of 16 (12, 36: the
    heads), pixel counts that are not a multiple of 16, and against torch fp32 on the same bf16 operands."""
    from v2x_sim_amd import ops, packing
    g = torch.Generator().manual_seed(Cin * 7 + Cout + M_shape[1])
    N, H, W = M_shape
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    pc = packing.pack_conv1x1_device("t", w.to(device), b.to(device), f32_out=f32_out)
    pc.relu = relu
    xd = x.to(device)
    ops.PROFILE = []
    new = ops.conv2d(pc, xd)
    torch.cuda.synchronize()
    ops.PROFILE = None
    tune("CONV1X1", 0)
    old = ops.conv2d(pc, xd)
    tune.reset("CONV1X1")
    assert new.dtype == old.dtype == (torch.float32 if f32_out else torch.bfloat16) and new.shape == (N, H, W, Cout)
    assert torch.equal(new, old), float((new.float() - old.float()).abs().max())
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b).permute(0, 2, 3, 1)
    if relu:
        ref = torch.relu(ref)
    tol = 2e-4 if f32_out else 2.0 ** -7
    assert torch.allclose(new.float().cpu(), ref, atol=2e-3 if not f32_out else 2e-4, rtol=tol), float((new.float().cpu() - ref).abs().max())
    assert torch.equal(ops.conv2d(pc, xd), new)


@pytest.mark.parametrize("shape,cp", [((2, 64, 64, 12), 32), ((1, 32, 32, 36), 64), ((10, 256, 256, 12), 32), ((3, 8, 32, 36), 64), ((1, 5, 3, 32), 32)])
def test_cast_pad_chsum_equals_the_torch_ops(device, shape, cp):
    """v2x_cast_pad_chsum_f32 (round 6: a 1x1 head's fp32 logit gradients -> bf16, channels zero-padded for the gradient kernels, bias gradient on the side) against
    F.pad + .to(bfloat16) (bit-equal) and an fp64 sum (fp32-accumulation error only); bit-reproducible."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + cp)
    x = (torch.randn(shape, generator=g) * 0.1 + 0.01).to(device)
    out, sums = ops.cast_pad_chsum(x, cp)
    ref = F.pad(x, (0, cp - shape[-1])).to(torch.bfloat16)
    assert out.shape == ref.shape and torch.equal(out, ref)
    ref_s = x.double().reshape(-1, shape[-1]).sum(0)
    scale = float(x.double().abs().reshape(-1, shape[-1]).sum(0).max())
    assert sums.shape == (shape[-1],) and float((sums.double() - ref_s).abs().max()) <= 2e-6 * scale
    out2, sums2 = ops.cast_pad_chsum(x, cp)
    assert torch.equal(out2, out) and torch.equal(sums2, sums)
    assert ops.cast_pad_chsum(x[..., :shape[-1] - 1].contiguous(), cp) is None if (shape[-1] - 1) % 4 else True      # C % 4 != 0: the caller's torch path


def test_head_gradient_pack_changes_the_step_only_in_the_last_bits(device, tune):
    """The heads' 1x1 backward with TRAIN_HEAD_PACK = 1 (one pass) and 0 (pad / cast / sum): same data gradient and weight gradient bits (the bf16 values are
    bit-equal), bias gradients equal to fp32 summation order."""
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 64, 64, 32, 36
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(device)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2).to(device)
    b = torch.randn(Cout, generator=g).to(device)
    dy = (torch.randn(N, H, W, Cout, generator=g) * 0.05).to(device)
    grads = {}
    for flag in (1, 0):
        tune("TRAIN_HEAD_PACK", flag)
        xd, wd, bd = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        hip_graph.conv1x1(xd, wd, bd, f32_out=True).backward(dy)
        grads[flag] = (xd.grad.clone(), wd.grad.clone(), bd.grad.clone())
    assert torch.equal(grads[1][0], grads[0][0]) and torch.equal(grads[1][1], grads[0][1])
    assert torch.allclose(grads[1][2], grads[0][2], rtol=1e-5, atol=1e-5)


def test_bias_gradient_in_front_of_a_batchnorm_is_zero(device, tune):
    """Round 6 (TRAIN_BN_BIAS_ZERO = 1, the default): a conv + batch-statistics BN pair returns its bias gradient as the exact value, zero.  Checked on a
    FaFNet step: (a) the COMPUTED values (switch 0: what autograd's sum gives up to order) are rounding residue -- below 5 % of a live bias gradient of the same
    step (the heads' last layers), although they sum 1e5-1e6 terms each; (b) the logits do not depend on those biases beyond the bf16 noise floor (shifting them all by 0.02 -- the BN subtracts the batch mean, whatever
    the bias added to it); (c) with the switch on those gradients are exact zeros and every OTHER gradient of the step is bit-identical."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    tune("TRAIN_HIP", 1)
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=4).to(device).train()
    data = synthetic_batch_on_device(cfg, 1, 2, seed=6, device=device)

    def step():
        model.zero_grad(set_to_none=True)
        res = train_forward(model, data["bev_seq"], None, None, 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    tune("TRAIN_BN_BIAS_ZERO", 0)
    l0, g0 = step()
    tune("TRAIN_BN_BIAS_ZERO", 1)
    l1, g1 = step()
    assert l0 == l1 and set(g0) == set(g1)
    bn_biases = [n for n in g1 if n.endswith("bias") and float(g1[n].abs().max()) == 0.0 and "bn" not in n]
    assert len(bn_biases) >= 20, bn_biases                      # every 3x3 / 1x1x1 convolution in front of a BatchNorm (22 in FaFNet, minus the width-padded layer)
    residue = max(float(g0[n].abs().max()) for n in bn_biases)
    signal = max(float(g0[n].abs().max()) for n in g0 if n.endswith("bias") and n not in bn_biases and "bn" not in n)     # the heads' last layers: real gradients
    print("bias gradients in front of a BatchNorm: computed residue <= %.3e; a live bias gradient of the same step: %.3e" % (residue, signal))
    assert residue <= 0.05 * signal, (residue, signal)
    for n in g0:
        if n not in bn_biases:
            assert torch.equal(g0[n], g1[n]), n
    # the biases are dead parameters in train mode: the BN subtracts whatever they add
    with torch.no_grad():
        ref = train_forward(model, data["bev_seq"], None, None, 1)
        for n, p in model.named_parameters():
            if n in bn_biases:
                p.add_(0.02)     # (small: the convolution's output is STORED in bf16 before the BN, so a large bias costs the map precision -- exact arithmetic only is bias-blind)
        got = train_forward(model, data["bev_seq"], None, None, 1)
    for k in ("cls", "loc"):
        # (the shifted maps are rounded to bf16 at other points: the two passes are two draws of the storage noise -- compared in the mean)
        d, scale = (got[k] - ref[k]).abs(), float(ref[k].abs().max())
        print("logits after shifting every such bias by 0.02: %s mean |diff| %.2e, max %.2e of max|ref|" % (k, float(d.mean()) / scale, float(d.max()) / scale))
        assert float(d.mean()) <= 1.5e-2 * scale, k      # measured 2e-3 (cls) ... 5.6e-3 (loc) of max|ref|: the train-mode bf16 noise floor of 22 BN-normalised layers


@pytest.mark.parametrize("N,H,W", [(2, 32, 32), (1, 128, 128), (3, 8, 16)])
def test_upcat_conv8_on_the_halo_kernels_vs_autograd_and_the_gather_path(device, tune, N, H, W):
    """Round 6 (TRAIN_UPCAT_CONV): conv8_1 = conv3x3(cat(up(lo 64 ch), skip 32 ch)) -> 32 -- forward on the two-source halo kernel, the data gradient as two
    halo launches (32 -> 64 | 32 -> 32, packed from row slices of the transposed weights: v2x_pack_spec.src_rows / src_row0) into one 96-channel map, then
    the upsample / concat backward.  Against fp32 autograd on the same bf16 operands (1 bf16 ulp of the largest value, as the other layers of the graph are
    held) and against the gather-kernel path it replaces (same operands, different K walk: 1 ulp); the weight gradient is the same launch either way: bit-equal."""
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(N * 100 + H)
    lo = torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16)
    skip = torch.randn(N, 2 * H, 2 * W, 32, generator=g).to(torch.bfloat16)
    conv = torch.nn.Conv2d(96, 32, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(32, 96, 3, 3, generator=g) * (2.0 / (96 * 9)) ** 0.5)
        conv.bias.copy_(torch.randn(32, generator=g) * 0.1)
    dy = torch.randn(N, 2 * H, 2 * W, 32, generator=g).to(torch.bfloat16)
    # fp32 autograd reference on the bf16-rounded operands
    lor, skr = lo.float().requires_grad_(True), skip.float().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    up = F.interpolate(lor.permute(0, 3, 1, 2), scale_factor=2)
    yr = F.conv2d(torch.cat((up, skr.permute(0, 3, 1, 2)), 1), wr, conv.bias.detach(), 1, 1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    res = {}
    convd = conv.to(device)
    for flag in (1, 0):
        tune("TRAIN_UPCAT_CONV", flag)
        hip_graph._CACHE.clear()
        lod, skd = lo.to(device).requires_grad_(True), skip.to(device).requires_grad_(True)
        convd.zero_grad(set_to_none=True)
        y = hip_graph.upcat_conv3x3(lod, skd, convd)
        y.backward(dy.to(device))
        res[flag] = (y.detach().float().cpu(), lod.grad.float().cpu(), skd.grad.float().cpu(), convd.weight.grad.detach().cpu().clone(), convd.bias.grad.detach().cpu().clone())
    ulp = 2.0 ** -7
    refs = (yr.detach().permute(0, 2, 3, 1), lor.grad, skr.grad)
    for k, name in enumerate(("y", "d_lo", "d_skip")):
        scale = float(refs[k].abs().max())
        for flag in (1, 0):
            assert float((res[flag][k] - refs[k]).abs().max()) <= ulp * scale, (name, flag, float((res[flag][k] - refs[k]).abs().max()), scale)
        assert float((res[1][k] - res[0][k]).abs().max()) <= ulp * scale, name
    assert torch.equal(res[1][3], res[0][3]) and torch.equal(res[1][4], res[0][4])
    assert float((res[1][3] - wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max())


@pytest.mark.parametrize("N,H,W,Cin,Cout,cin_out", [(2, 16, 64, 64, 64, None), (1, 8, 32, 32, 32, 13), (3, 24, 32, 96, 32, None), (2, 32, 32, 256, 128, None), (10, 128, 128, 64, 64, None),
                                                    (4, 16, 32, 512, 512, None), (2, 64, 64, 32, 32, 12)])
def test_vectorised_wgrad_reduce_equals_the_scalar_form_bitwise(device, tune, N, H, W, Cin, Cout, cin_out):
    """wgrad_reduce4_kernel (round 6: four consecutive ci per thread, 16-byte loads, eight partials in flight) against the scalar reduce: the same association (a slice's
    partials in slot order, the eight slice sums in slice order) -> the same bits; cin_out % 4 != 0 (the 13-channel first layer) keeps the scalar form."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(N + H + Cin + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16).to(device)
    new = ops.conv3x3_wgrad(x, dy, cin_out=cin_out)
    tune("WGRAD_REDUCE4", 0)
    old = ops.conv3x3_wgrad(x, dy, cin_out=cin_out)
    tune.reset("WGRAD_REDUCE4")
    assert new.shape == old.shape == (Cout, cin_out or Cin, 3, 3) and torch.equal(new, old), float((new - old).abs().max())


# ------------------------------------------------------------------ V2VNet's fusion stage on bf16 NHWC (csrc/v2v_train.hip, round 6)
def _v2v_case(A, B, C, H, W, seed, device, shrink=False):
    """Agent-major maps, poses (small rotations + translations of a few cells, one pair pushed mostly off the map) and the plan tables of hip_graph._v2v_plan."""
    from v2x_sim_amd.models.det.base import IntermediateModelBase
    from v2x_sim_amd.train import hip_graph
    g = torch.Generator().manual_seed(seed)
    N = A * B
    feat = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    ang = (torch.rand(B, A, A, generator=g) - 0.5) * 1.2
    T = torch.zeros(B, A, A, 4, 4)
    T[..., 0, 0], T[..., 0, 1], T[..., 1, 0], T[..., 1, 1] = torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)
    T[..., 0, 3] = (torch.rand(B, A, A, generator=g) - 0.5) * 12.0
    T[..., 1, 3] = (torch.rand(B, A, A, generator=g) - 0.5) * 12.0
    T[0, 0, 1, 0, 3] = 40.0          # one neighbour almost entirely outside the ego's map
    T[..., 2, 2] = T[..., 3, 3] = 1.0
    if shrink:                       # a strongly shrinking pose: many output pixels per input pixel -- the backward kernel's direct loops
        T[0, 1, 0, :2, :2] *= 0.3
    nat = torch.full((B, A), A)
    counts, items, rows = IntermediateModelBase.frame_plan(nat, B, A)

    class _M:        # (the plan cache lives in the model's __dict__)
        pass
    plan = hip_graph._v2v_plan(_M(), counts, items, rows, B, A, T, N, device)
    assert plan is not None and plan["identity"]
    return feat, T, plan, counts, items


def _v2v_reference(base32, cur32, T, counts, items, B):
    """graph.py::v2v_fuse's message + concat with torch ops (F.grid_sample twice per pair, mean, cat) on fp32 NCHW maps."""
    from v2x_sim_amd.train import graph
    pairs = [(m, j * B + f, f, a, j) for m, (a, f) in enumerate(items) for j in range(counts[f]) if j != a]
    src = torch.tensor([p[1] for p in pairs])
    Tp = torch.stack([T[f, a, j] for (_, _, f, a, j) in pairs])
    warped = graph.warp_batch(base32.index_select(0, src), Tp)
    K = counts[0] - 1
    mean = warped.view((len(items), K) + tuple(base32.shape[1:])).mean(1)
    return torch.cat([cur32, mean], 1)


@pytest.mark.parametrize("A,B,C,H,W,two,shrink", [(3, 2, 32, 16, 32, False, False), (5, 1, 64, 32, 32, False, False), (4, 2, 32, 8, 32, True, False),
                                                  (2, 3, 256, 32, 32, False, False), (3, 1, 512, 16, 16, False, False), (3, 1, 32, 16, 32, False, True),
                                                  (5, 1, 96, 12, 20, True, False), (6, 1, 32, 16, 32, False, False), (10, 1, 32, 8, 32, False, False)])
def test_v2v_message_forward_and_backward_vs_torch(device, A, B, C, H, W, two, shrink):
    """v2x_v2v_message_bf16 / _bwd_bf16 against F.grid_sample o F.grid_sample, mean, cat and their autograd backward in fp32 on the same bf16 maps: the
    forward to one bf16 rounding of the value (+ the coordinate arithmetic's fp32 noise), the backward likewise and as the exact transpose of the forward
    (<d, F(x)> == <F^T d, x> to fp32 summation noise); bit-identical from run to run."""
    from v2x_sim_amd import ops
    feat, T, plan, counts, items = _v2v_case(A, B, C, H, W, 100 * A + C, device, shrink)
    g = torch.Generator().manual_seed(7)
    other = torch.randn(feat.shape, generator=g).to(torch.bfloat16)
    cur, base = (other, feat) if two else (feat, feat)
    x = base.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    c = cur.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = _v2v_reference(x, c if two else x, T, counts, items, B)
    Td = T.to(device)
    got = ops.v2v_message(cur.to(device), base.to(device) if two else None, Td, plan)
    got2 = ops.v2v_message(cur.to(device), base.to(device) if two else None, Td, plan)
    assert torch.equal(got, got2)
    refn = ref.detach().permute(0, 2, 3, 1)
    err = (got.cpu().float() - refn).abs()
    tol = 2.0 ** -8 * refn.abs() + 2e-5 * float(refn.abs().max())
    assert bool((err <= tol).all()), float((err - tol).max())
    assert torch.equal(got.cpu()[..., :C], cur)                                   # the ego half is a copy
    d = torch.randn(ref.shape, generator=g).to(torch.bfloat16)                    # NCHW values
    ref.backward(d.float())
    dn = d.permute(0, 2, 3, 1).contiguous().to(device)
    dbase, dcur = ops.v2v_message_backward(dn, Td, plan, feat.shape[0], two)
    dbase2, _ = ops.v2v_message_backward(dn, Td, plan, feat.shape[0], two)
    assert torch.equal(dbase, dbase2)
    gx = x.grad.permute(0, 2, 3, 1)
    e = (dbase.cpu().float() - gx).abs()
    t = 2.0 ** -8 * gx.abs() + 5e-5 * float(gx.abs().max())
    assert bool((e <= t).all()), float((e - t).max())
    if two:
        assert torch.equal(dcur.cpu(), d.permute(0, 2, 3, 1)[..., :C].contiguous())
    # transpose identity on the message half (fp64 inner products of the kernels' own outputs; bf16 rounding of both sides bounds the mismatch)
    msg = got.cpu().double()[..., C:]
    lhs = float((d.permute(0, 2, 3, 1).double()[..., C:] * msg).sum())
    dmsg_only = dn.clone()
    dmsg_only[..., :C] = 0
    db_only, _ = ops.v2v_message_backward(dmsg_only, Td, plan, feat.shape[0], two)
    rhs = float((db_only.cpu().double() * base.double()).sum())
    scale = float((d.double().abs().permute(0, 2, 3, 1)[..., C:] * msg.abs()).sum())
    # (both sides carry one bf16 rounding per element, random in sign: ~2^-9 scale / sqrt(n) ~ 1e-5 scale; unrelated operators would differ by ~4e-3 scale)
    assert abs(lhs - rhs) <= 1e-4 * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("P,C", [(2 * 32 * 32, 256), (3 * 16 * 32, 64), (1001, 32)])
def test_gru_gates_nhwc_vs_torch(device, P, C):
    """v2x_gru_gates_nhwc_bf16 / _bwd_bf16 against the PyTorch ops of graph.py::_gru_step in fp32 on the same bf16 pre-activations: h and dgi to one bf16
    rounding, the six channel-sum vectors against sums of the stored dgi (d bias_ih) and of the fp32 dpre_n * r (d bias_hh); fixed order."""
    from v2x_sim_amd import ops
    g = torch.Generator().manual_seed(P + C)
    gi = (torch.randn(P, 3 * C, generator=g) * 1.5).to(torch.bfloat16)
    bhh = torch.randn(3 * C, generator=g) * 0.5
    dh = torch.randn(P, C, generator=g).to(torch.bfloat16)
    x = gi.float().requires_grad_(True)
    b = bhh.clone().requires_grad_(True)
    i_r, i_z, i_n = x.chunk(3, 1)
    h_r, h_z, h_n = b.view(1, -1).chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    h = n - z * n
    h.backward(dh.float())
    got = ops.gru_gates_nhwc(gi.to(device), bhh.to(device))
    hb = h.detach().to(torch.bfloat16).float()
    assert bool(((got.cpu().float() - hb).abs() <= 2.0 ** -7 * hb.abs() + 1e-6).all())
    dgi, sums = ops.gru_gates_nhwc_backward(gi.to(device), bhh.to(device), dh.to(device))
    dgi2, sums2 = ops.gru_gates_nhwc_backward(gi.to(device), bhh.to(device), dh.to(device))
    assert torch.equal(dgi, dgi2) and torch.equal(sums, sums2)
    gb = x.grad.to(torch.bfloat16).float()
    assert bool(((dgi.cpu().float() - gb).abs() <= 2.0 ** -7 * gb.abs() + 1e-6 * float(gb.abs().max())).all())
    stored = dgi.cpu().double().sum(0)
    s = sums.cpu().double()
    tol = 1e-6 * float(dgi.cpu().double().abs().sum(0).max()) + 1e-7
    assert float((s[:3 * C] - stored).abs().max()) <= tol
    assert torch.equal(sums[3 * C:5 * C], sums[:2 * C])
    ref_bhh = b.grad.double()
    assert float((s[5 * C:] - ref_bhh[2 * C:]).abs().max()) <= 1e-4 * float(ref_bhh.abs().max()) + 1e-5
    assert float((s[3 * C:5 * C] - ref_bhh[:2 * C]).abs().max()) <= 2e-3 * float(ref_bhh.abs().max()) + 1e-4      # (sums of bf16-rounded terms vs fp32 terms)


@pytest.mark.parametrize("rounds,source", [(1, "initial"), (2, "initial"), (2, "updated")])
def test_v2vnet_training_step_nhwc_fusion_stage_vs_fp32_stage(device, tune, rounds, source):
    """A V2VNet training step with the message-passing rounds on bf16 NHWC (TRAIN_V2V_NHWC 1) against the same step with the fp32 NCHW stage around the warp /
    gates kernels (0): same loss to 2e-3, gradients within the bf16 graph's own amplification (the bound of the WARP_HIP test), every parameter has a
    gradient either way, bit-reproducible run to run -- and no PyTorch-op launch is left in the stage (asserted on the op names torch.profiler records)."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    cfg = Config("train", binary=True, only_det=True)
    model = init_for_training(V2VNet(cfg, num_agent=3, gnn_iter_times=rounds, neighbor_source=source), seed=2).to(device).train()
    data = synthetic_batch_on_device(cfg, 2, 3, seed=5, device=device)
    tune("TRAIN_HIP", 1)

    def run():
        model.zero_grad(set_to_none=True)
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    def rel(ga, gb):
        den = sum(float(gb[n].double().pow(2).sum()) for n in gb) ** 0.5
        return sum(float((ga[n].double() - gb[n].double()).pow(2).sum()) for n in gb) ** 0.5 / den
    tune("TRAIN_V2V_NHWC", 1)
    l1, g1 = run()
    l1b, g1b = run()
    tune("TRAIN_V2V_NHWC", 0)
    l0, g0 = run()
    assert l1 == l1b and all(torch.equal(g1[n], g1b[n]) for n in g1)
    gru = [n for n in g0 if "convgru" in n]
    print("V2VNet step, %d round(s), neighbours '%s': loss %.6f (NHWC stage) vs %.6f (fp32 stage); gradient difference %.2e of the norm (ConvGRU parameters alone: %.2e)"
          % (rounds, source, l1, l0, rel(g1, g0), rel({n: g1[n] for n in gru}, {n: g0[n] for n in gru})))
    assert set(g1) == set(g0) and abs(l1 - l0) <= 2e-3 * max(1.0, abs(l0)) and rel(g1, g0) < 0.25


# ------------------------------------------------------------------ the optimizer step (csrc/adam.hip, train/optim.py; round 6)
@pytest.mark.parametrize("mode,wd", [("capturable", 0.0), ("fused", 0.0), ("plain", 0.0), ("capturable", 0.01), ("plain", 0.01)])
def test_hip_adam_matches_torch_adam(device, tune, mode, wd):
    """train/optim.py::HipAdam (v2x_adam_step_f32) against torch.optim.Adam on the same parameters and gradients, 6 steps: parameters and both moments agree
    to fp32 rounding of the update; 150 tensors of odd sizes (three launches, scalar tails, a zero-element tensor, one tensor that joins late in the
    non-capturable modes' fallback); the state dicts are interchangeable both ways."""
    from v2x_sim_amd.train.optim import HipAdam, use_hip_adam
    g = torch.Generator().manual_seed(11)
    sizes = [1, 7, 4096, 5000, 100003, 16384, 0] + [int(s) for s in torch.randint(1, 3000, (143,), generator=g)]
    base = [torch.randn(n, generator=g) for n in sizes]
    pa = [torch.nn.Parameter(b.clone().to(device)) for b in base]
    pb = [torch.nn.Parameter(b.clone().to(device)) for b in base]
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    if mode == "capturable":
        kw.update(lr=torch.tensor(3e-3, device=device), capturable=True)
    elif mode == "fused":
        kw.update(lr=3e-3, fused=True)
    else:
        kw.update(lr=3e-3)
    ref = torch.optim.Adam(pa, **{k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}, **({} if mode == "fused" else {"foreach": True}))
    opt = use_hip_adam(torch.optim.Adam(pb, **kw))
    assert isinstance(opt, HipAdam)
    tune("TRAIN_ADAM_HIP", 1)
    late = 1 if mode == "plain" else -1          # host step counters: a tensor that sits out the first step lags by one -> torch's own step from then on
    for it in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g).to(device) * (0.1 + it)
            a.grad, b.grad = (None, None) if (i == late and it == 0) else (gr.clone(), gr.clone())
        ref.step()
        opt.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (i, sizes[i], float((a - b).abs().max()))
        if a.numel():
            sa, sb = ref.state[a], opt.state[b]
            if i == late:
                assert float(sa["step"]) == float(sb["step"]) == 5.0
                continue
            # (fp32 rounding of the two forms of the moving averages, relative to the tensor's scale: an element of m near zero is a cancelling sum)
            assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6 * float(sa["exp_avg"].abs().max()))
            assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-6 * float(sa["exp_avg_sq"].abs().max()))
            assert float(sa["step"]) == float(sb["step"]) == 6.0
    # interchangeable state
    import copy
    ref.load_state_dict(copy.deepcopy(opt.state_dict()))          # (load_state_dict keeps same-device tensors by reference: copies, or the two would share moments)
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))
    for a, b in zip(pa, pb):
        gr = torch.randn(a.shape, generator=g).to(device)
        a.grad, b.grad = gr.clone(), gr.clone()
    ref.step()
    opt.step()
    assert all(torch.allclose(a, b, rtol=2e-6, atol=2e-7) for a, b in zip(pa, pb))
    # the switch off: torch's own step inside the same object
    tune("TRAIN_ADAM_HIP", 0)
    opt.step()
    ref.step()
    assert all(torch.allclose(a, b, rtol=2e-6, atol=2e-7) for a, b in zip(pa, pb))


def test_hip_adam_step_is_captured_with_the_training_step(device, tune):
    """GraphedTrainStep with a plain torch.optim.Adam(capturable=True): the optimizer is switched to the library's kernel, the captured step contains no
    multi_tensor_apply launch of torch's fused Adam, and three replays equal three eager steps of a twin model on torch's own Adam to 1e-5 of the weights' scale."""
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    from v2x_sim_amd.train.optim import HipAdam
    import copy
    tune("TRAIN_HIP", 1)
    cfg = Config("train", binary=True, only_det=True)
    m1 = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=2), seed=3).to(device).train()
    m0 = copy.deepcopy(m1)
    data = synthetic_batch_on_device(cfg, 1, 2, seed=4, device=device)
    o1 = torch.optim.Adam(m1.parameters(), lr=torch.tensor(1e-3, device=device), capturable=True)
    step1 = GraphedTrainStep(m1, o1, data, 1)
    assert isinstance(o1, HipAdam)
    tune("TRAIN_ADAM_HIP", 0)
    o0 = torch.optim.Adam(m0.parameters(), lr=torch.tensor(1e-3, device=device), capturable=True)
    step0 = GraphedTrainStep(m0, o0, data, 1)
    assert type(o0) is torch.optim.Adam
    for _ in range(3):
        l1 = step1(data)[0]
        l0 = step0(data)[0]
    torch.cuda.synchronize()
    assert abs(float(l1) - float(l0)) <= 5e-3 * max(1.0, abs(float(l0)))
    # Adam normalises the gradient: where it is near zero, bf16-level differences between the two runs decide its sign, and each step moves the parameter by up
    # to lr either way -- the bound is 2 x 3 steps x lr absolute, plus 1 % of the tensor's scale
    worst = 0.0
    for (n, a), (_, b) in zip(m1.named_parameters(), m0.named_parameters()):
        worst = max(worst, float((a - b).abs().max()) / (6 * 1e-3 + 1e-2 * float(b.abs().max())))
    print("three captured steps, HIP Adam vs torch Adam: loss %.6f vs %.6f, worst parameter difference %.2f of its bound" % (float(l1), float(l0), worst))
    assert worst <= 1.0
