#!/usr/bin/env python3
"""CPU study for the next kernel round: the four decoder `_1` layers convolve cat(nearest-x2-upsample(lo), skip).
For the upsampled operand a 3x3 tap window of an output pixel (y, x) touches only 2 x 2 low-resolution pixels, which ones depending on the
parities (y & 1, x & 1): per parity class the layer is a 2 x 2-tap convolution on `lo` with PRE-SUMMED weights

    rows  py = 0: {ky = 0} -> Y - 1, {ky = 1, 2} -> Y          py = 1: {ky = 0, 1} -> Y, {ky = 2} -> Y + 1        (columns alike)

i.e. C0 x 4 instead of C0 x 9 multiply-adds per output pixel for that operand.  This script checks (a) the identity in fp64 and (b) what the
one extra rounding costs: the summed weights must be bf16 for the MFMA, so W_eff = bf16(W_a + W_b [+ W_c + W_d]) is not the sum of the bf16
weights the 9-tap form multiplies.  Reported per layer shape: the difference between the two forms' bf16 OUTPUTS in units of a bf16 ulp of the
output, against the difference either form has to the fp32-weights result anyway.
usage: python tools/parity_class_study.py            (CPU, ~1 min)"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(8)


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float64)


SETS = {0: {-1: (0,), 0: (1, 2)}, 1: {0: (0, 1), 1: (2,)}}      # parity -> low-res offset -> taps that land there


def eff_weights(w, round_bf16):
    """w (Cout, C0, 3, 3) -> {(py, px): {(dy, dx): (Cout, C0)}}"""
    out = {}
    for py in (0, 1):
        for px in (0, 1):
            d = {}
            for dy, kys in SETS[py].items():
                for dx, kxs in SETS[px].items():
                    s = sum(w[:, :, ky, kx] for ky in kys for kx in kxs)
                    d[(dy, dx)] = bf16r(s) if round_bf16 else s
            out[(py, px)] = d
    return out


def conv_parity(lo, weff, H, W):
    """lo (N, C0, H/2, W/2) -> (N, Cout, H, W): the upsampled operand's part of the layer, per parity class at half resolution."""
    N, C0, h, w_ = lo.shape
    lop = F.pad(lo, (1, 1, 1, 1))
    Cout = next(iter(weff[(0, 0)].values())).shape[0]
    out = torch.zeros((N, Cout, H, W), dtype=lo.dtype)
    for (py, px), taps in weff.items():
        acc = torch.zeros((N, Cout, h, w_), dtype=lo.dtype)
        for (dy, dx), wm in taps.items():
            src = lop[:, :, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w_]
            acc += torch.einsum("oc,nchw->nohw", wm, src)
        out[:, :, py::2, px::2] = acc
    return out


print("layer      C0+C1->Cout @HxW | identity (fp64, max |d| / max |y|) | 9-tap bf16 vs fp32-weights | parity bf16 vs fp32-weights | parity vs 9-tap   (bf16 ulps of the output: mean, p99, max; rms error / rms output)")
for name, C0, C1, Cout, H in (("conv8_1", 64, 32, 32, 64), ("conv7_1", 128, 64, 64, 32), ("conv6_1", 256, 128, 128, 16), ("conv5_1", 512, 256, 256, 8)):
    N = 2
    lo = bf16r(torch.relu(torch.randn(N, C0, H // 2, H // 2)))
    skip = bf16r(torch.relu(torch.randn(N, C1, H, H)))
    w = torch.randn(Cout, C0 + C1, 3, 3, dtype=torch.float64) * (2.0 / ((C0 + C1) * 9)) ** 0.5      # He init, fp32 master weights
    up = lo.repeat_interleave(2, 2).repeat_interleave(2, 3)
    x = torch.cat((up, skip), 1)
    y_fp = F.conv2d(x, w, None, 1, 1)                                   # fp32-weights result (what the bf16 network approximates)
    w_b = bf16r(w)
    y9 = F.conv2d(x, w_b, None, 1, 1)                                   # today's form: bf16 weights, 9 taps
    # (a) identity with unrounded sums of the bf16 weights
    y_id = conv_parity(lo, eff_weights(w_b[:, :C0], False), H, H) + F.conv2d(skip, w_b[:, C0:], None, 1, 1)
    ident = float((y_id - y9).abs().max() / y9.abs().max())
    # (b) the MFMA form: effective weights rounded to bf16 (from the fp32 master weights: one rounding)
    yp = conv_parity(lo, eff_weights(w[:, :C0], True), H, H) + F.conv2d(skip, w_b[:, C0:], None, 1, 1)
    o9, op = bf16r(torch.relu(y9)), bf16r(torch.relu(yp))
    ref = torch.relu(y_fp)
    ulp = torch.clamp(y_fp.abs(), min=2.0 ** -6) * 2.0 ** -8             # one bf16 ulp at the output's magnitude (floor: tiny outputs)

    def stats(a, b):
        d = ((a - b).abs() / ulp).flatten()
        rms = float(((a - b) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        return "%.2f %.2f %.2f rms %.1e" % (float(d.mean()), float(d.quantile(0.99)), float(d.max()), rms)
    print("%-9s %4d+%-4d->%-4d @%-3d | %.1e | %s | %s | %s" % (name, C0, C1, Cout, H, ident, stats(o9, ref), stats(op, ref), stats(op, o9)))
print("MACs per output pixel, upsampled operand: 9 C0 -> 4 C0; whole layer: conv8_1 864 -> 544, conv7_1 1728 -> 1088, conv6_1 3456 -> 2176, conv5_1 6912 -> 4352 (-37 %)")
