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


def test_training_step_on_hip_conv_kernels(device, monkeypatch):
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
        monkeypatch.setenv("V2X_TRAIN_HIP_CONV", flag)
        model.zero_grad()
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 1)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        loss.backward()
        losses[flag] = float(loss.detach())
        grads[flag] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    monkeypatch.delenv("V2X_TRAIN_HIP_CONV")
    print("loss MIOpen %.5f, HIP conv kernels %.5f" % (losses["0"], losses["1"]))
    assert abs(losses["1"] - losses["0"]) <= 1e-2 * abs(losses["0"])
    worst, worst_k = 0.0, ""
    for k, g0 in grads["0"].items():
        d = float((grads["1"][k] - g0).abs().max()) / max(float(g0.abs().max()), 1e-12)
        if d > worst:
            worst, worst_k = d, k
    print("worst relative gradient difference %.3e (%s)" % (worst, worst_k))
    assert worst < 5e-2, (worst, worst_k)
