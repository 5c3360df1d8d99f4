"""3x3 stride-1 convolution for TRAINING on the hand-written HIP kernels.

Upstream trains through torch.autograd over nn.Conv2d (cuDNN; MIOpen on ROCm).  Here forward, data gradient and weight
gradient of the eligible layers run on libv2x_amd.so:

    forward          y  = conv3x3(x, W) + b                      v2x_conv2d (the inference kernels, plain epilogue)
    data gradient    dx = conv3x3(dy, W'),  W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]      v2x_conv2d on the flipped weights
    weight gradient  dW = sum_pixels dy (x) x_shifted            v2x_conv3x3_wgrad (conv_wgrad.hip, MFMA over pixels)
    bias gradient    db = sum_pixels dy                          (a torch reduction: plumbing)

Mixed precision as on the inference path: activations, gradients and weights are rounded to bf16 where they enter an MFMA,
every sum is fp32, the master weights stay fp32.  Eligible: 3x3, stride 1, pad 1, Cin % 32 == 0, Cout % 64 == 0, H % 8 == 0,
W % 32 == 0 (conv1_2 ... conv7_2 of the backbone at the training resolutions).  `train/graph.py` routes those layers here when
V2X_TRAIN_HIP_CONV=1; everything else (BatchNorm with batch statistics, ReLU, the stride-2 / 13- and 32-channel layers, the
heads) stays on PyTorch-ROCm ops.  There is no CPU path.
"""
import weakref

import torch

from .. import ops, packing
from .._lib import V2X_EPI_BF16


def eligible(weight, stride, padding, H, W):
    cout, cin, kh, kw = weight.shape
    return (kh == 3 and kw == 3 and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and cin % 32 == 0 and cout % 64 == 0
            and cin >= 32 and H % 8 == 0 and W % 32 == 0)


def _pack_plain(name, w, bias, device):
    """conv3x3 stride 1 with bias, no BN, no ReLU, bf16 out -> the kernel the extent allows (streamed / gather)."""
    cout, cin = w.shape[0], w.shape[1]
    scale = torch.ones(cout)
    shift = bias.detach().float().cpu() if bias is not None else torch.zeros(cout)
    fb = packing.pack_conv(name, w, scale, shift, stride=1, pad=1, relu=False, epilogue=V2X_EPI_BF16, device=device)
    halo = None
    if cin >= 64 and cin % 32 == 0 and cout % 64 == 0:
        halo = packing.pack_conv_stream(name, w, scale, shift, C0=cin, relu=False, device=device)
    return ops.Layer([fb], halo, name=name)


_PACK_CACHE = {}


def _packed(kind, weight, bias, device):
    """Packed forward / dgrad weights of a parameter, re-packed only when the parameter changed (optimizer step)."""
    key = (kind, id(weight), str(device))   # the parameter OBJECT (weak reference checked on a hit): a data_ptr can be reused
    ver = (packing.param_version(weight), None if bias is None else packing.param_version(bias))
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is weight:
        return hit[1]
    if kind == "fwd":
        layer = _pack_plain("train.fwd", weight, bias, device)
    else:
        layer = _pack_plain("train.dgrad", weight.detach().flip(2, 3).transpose(0, 1).contiguous(), None, device)   # W'[ci][co][ky][kx]
    if len(_PACK_CACHE) > 256:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (ver, layer, weakref.ref(weight))
    return layer


class _HipConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        """x (N, H, W, Cin) bf16 NHWC on the MI355X; weight (Cout, Cin, 3, 3) fp32; bias (Cout,) or None -> (N, H, W, Cout) bf16."""
        dev = x.device
        y = ops.run_layer(_packed("fwd", weight, bias, dev), x)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.run_layer(_packed("dgrad", weight, None, dy.device), dy)
        if ctx.needs_input_grad[1]:
            dw = ops.conv3x3_wgrad(x, dy)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum((0, 1, 2))
        return dx, dw, db


def conv3x3_nhwc(x, weight, bias=None):
    """Differentiable 3x3 / stride 1 / pad 1 convolution on the HIP kernels.  x bf16 NHWC (N, H, W, Cin)."""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        raise RuntimeError("hip_conv.conv3x3_nhwc takes a bf16 NHWC tensor on the MI355X (no CPU path)")
    return _HipConv3x3.apply(x.contiguous(), weight, bias)


def conv2d_nchw(x, conv):
    """Drop-in for `conv(x)` inside the fp32 NCHW training graph: layout / precision conversions around conv3x3_nhwc
    (torch permutes + casts: plumbing; a graph that keeps NHWC bf16 between layers would shed them)."""
    y = conv3x3_nhwc(x.permute(0, 2, 3, 1).to(torch.bfloat16), conv.weight, conv.bias)
    return y.permute(0, 3, 1, 2).to(x.dtype)
