"""BEV semantic segmentation variants (BASELINE.json config 5).

Upstream coperception/models/seg/* is not in /root/reference and its backbone widths are
unverified, so the build-owned spec reuses the detection backbone (rows a2/a6)
and replaces the det heads by a 1x1 conv 32 -> n_classes producing NHWC fp32 logits
(DESIGN.md section 3.6).  n_classes = 8 as recollected for V2X-Sim.
"""
import torch.nn as nn

from ... import ops, packing
from ..._lib import V2X_EPI_F32
from ..det.base import LidarDecoder, LidarEncoder, _ParamsOnly
from ..det.FaFNet import FaFNet
from ..det.V2VNet import V2VNet


class OutConv(_ParamsOnly):
    def __init__(self, cin, n_classes):
        super().__init__()
        self.conv = nn.Conv2d(cin, n_classes, kernel_size=1)


def _pack_seg_head(outc, device):
    scale, shift = packing.fold_bn(outc.conv.bias, None, outc.conv.out_channels)
    return packing.pack_conv("outc", outc.conv.weight, scale, shift, stride=1, pad=0, relu=False,
                             epilogue=V2X_EPI_F32, device=device)


class V2VNetSeg(V2VNet):
    """forward -> logits (A*B, 256, 256, n_classes) fp32 NHWC (upstream returns NCHW; argmax / CE are
    layout-agnostic, and ops.seg_argmax_confusion consumes NHWC directly)."""

    def __init__(self, config, n_classes=8, **kw):
        super().__init__(config, **kw)
        self.n_classes = n_classes
        self.outc = OutConv(32, n_classes)

    def _pack(self, device):
        pk = super()._pack(device)
        pk["seg"] = _pack_seg_head(self.outc, device)
        return pk

    @ops.latency_entry
    def forward_nhwc(self, x0, trans_matrices, num_agent_tensor, batch_size=1, plan=None):
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0)
        if plan is None:
            plan = self.make_plan(num_agent_tensor, batch_size, x0.device)
        feats[self.layer] = self.fuse(feats[self.layer], trans_matrices, plan, batch_size, pk)
        x = LidarDecoder.run(pk["dec"], *feats)
        return ops.conv2d(pk["seg"], x)


class FaFNetSeg(FaFNet):
    def __init__(self, config, n_classes=8, **kw):
        super().__init__(config, **kw)
        self.n_classes = n_classes
        self.outc = OutConv(32, n_classes)

    def _pack(self, device):
        pk = super()._pack(device)
        pk["seg"] = _pack_seg_head(self.outc, device)
        return pk

    @ops.latency_entry
    def forward_nhwc(self, x0):
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0)
        return ops.conv2d(pk["seg"], LidarDecoder.run(pk["dec"], *feats))
