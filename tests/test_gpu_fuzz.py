"""Randomised shape sweep of the patch-based conv kernels (streamed 4-wave / 8-wave persistent / wide / stride-2 / halo)
against torch-CPU fp32 on the same bf16 operands.  The hand-picked cases of the other test files pin the layer shapes of
the network; this sweep hunts for tiling edge cases: tile counts that are not a multiple of the persistent grid, one tile
per map, several channel tiles, upsampled sources on the smallest legal maps, batch sizes around the 256-workgroup
boundary.  Every case also re-launches twice and demands identical bits (the kernels' counted-vmcnt pipelines must not
race).  Seeded: the same 40 cases every run."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)


def back(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _cases():
    rnd = random.Random(20261001)
    cases = []
    for _ in range(14):      # stride 1, streamed family (4-wave, 8-wave persistent, wide): Cout in 64 / 128 / 256
        cout = rnd.choice([64, 64, 128, 128, 256])
        up = rnd.random() < 0.4
        c = rnd.choice([64, 128, 256]) if not up else rnd.choice([64, 128])
        cup = rnd.choice([64, 128, 256]) if up else 0
        if cup + c < 128:
            c = 128
        t16 = rnd.random() < 0.25
        if t16:
            H, W = 16 * rnd.randint(1, 2), 16
        else:
            H, W = 8 * rnd.randint(1, 6), 32 * rnd.randint(1, 2)
        N = rnd.choice([1, 2, 3, 5, 9, 40])
        cases.append(("s1", cup, c, cout, N, H, W))
    for _ in range(6):       # enough tiles for the persistent 8-wave grid to wrap unevenly (n_tiles > 256)
        cout = rnd.choice([128, 256])
        c = rnd.choice([128, 256])
        H, W = 16 * rnd.randint(1, 2), 32 * rnd.randint(1, 2)
        tiles_per_map = (H // 16) * (W // 32) * (cout // 128)
        N = 256 // tiles_per_map + rnd.randint(1, 40)
        cases.append(("s1", 0, c, cout, N, H, W))
    for _ in range(10):      # stride 2
        c = rnd.choice([32, 64, 128])
        cout = rnd.choice([64, 128, 256])
        H, W = 8 * rnd.randint(1, 5), 64 * rnd.randint(1, 2)
        N = rnd.choice([1, 2, 3, 7])
        cases.append(("s2", 0, c, cout, N, H, W))
    for _ in range(8):       # stride 2, 8-wave three-tap form (forced: S2_G = 2): 128-row tiles, >= 2 chunks, outputs tile by 8 x 32 or 16 x 16
        c = rnd.choice([64, 96, 128, 256])
        cout = rnd.choice([128, 256])
        if rnd.random() < 0.5:
            H, W = 16 * rnd.randint(1, 3), 64 * rnd.randint(1, 2)      # 8 x 32 output tiles
        else:
            H, W = 32 * rnd.randint(1, 2), 32 * rnd.choice([1, 3])      # 16 x 16 output tiles
        N = rnd.choice([1, 2, 5, 70, 133])                              # 70, 133: the persistent grid wraps (unevenly)
        cases.append(("s2g", 0, c, cout, N, H, W))
    for _ in range(10):      # halo family: the instantiations that exist
        key = rnd.choice([(0, 32, 32), (64, 32, 32), (0, 64, 64)])
        H, W = 8 * rnd.randint(1, 5), 32 * rnd.randint(1, 3)
        N = rnd.choice([1, 2, 5, 33])
        cases.append(("halo",) + key + (N, H, W))
    return cases


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "-".join(str(v) for v in c))
def test_conv_kernels_random_shapes(device, case, tune):
    from v2x_sim_amd import ops, packing
    kind, cup, c, cout, N, H, W = case
    if kind == "s2g":
        tune("S2_G", 2)
    g = torch.Generator().manual_seed(sum(case[1:]) * 7919 + len(case[0]))
    x = bf16r(torch.randn(N, c, H, W, generator=g))
    x_up = bf16r(torch.randn(N, cup, H // 2, W // 2, generator=g)) if cup else None
    w = torch.randn(cout, cup + c, 3, 3, generator=g) * (2.0 / ((cup + c) * 9)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    stride = 2 if kind in ("s2", "s2g") else 1
    xin = torch.cat((F.interpolate(x_up, scale_factor=(2, 2)), x), 1) if cup else x
    ref = F.relu(F.conv2d(xin, bf16r(w), None, stride, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    if kind == "halo":
        pc = packing.pack_conv_halo("h", w, scale, shift, C0=cup if cup else c, C1=c if cup else 0, relu=True, device=device)
    else:
        pc = packing.pack_conv_stream("t", w, scale, shift, C0=cup if cup else c, C1=c if cup else 0, up0=1 if cup else 0,
                                      stride=stride, device=device)
    run = (lambda: ops.conv2d(pc, nhwc(x_up, device), nhwc(x, device))) if cup else (lambda: ops.conv2d(pc, nhwc(x, device)))
    if kind == "s2g":
        assert ops.conv_kernel_name(pc, H, W, False, N).startswith("conv3x3_s2g_kernel"), case
    y = run()
    got = back(y)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, atol=2e-3, rtol=2 ** -7), float((got - ref).abs().max())
    for _ in range(2):
        assert torch.equal(run(), y)
