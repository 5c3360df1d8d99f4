"""BEV semantic segmentation variants (BASELINE.json config 5).

Upstream coperception/models/seg/* is not in /root/reference and its backbone widths are
unverified, so the build-owned spec reuses the detection backbone (rows a2/a6)
and replaces the det heads by a 1x1 conv 32 -> n_classes producing NHWC fp32 logits
(DESIGN.md section 3.6).  n_classes = 8 as recollected for V2X-Sim.
"""
import torch.nn as nn

from ... import ops, packing, tuning
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


def _pack_seg_fused(decoder, outc, device):
    """conv8_2 (3x3 32 -> 32, BN, ReLU) chained with the 1x1 class head in ONE halo launch (conv_halo.hip: <0, 32, 32, 16, 2>): the 32-channel
    map is rounded to bf16 exactly as the stand-alone layer stores it but never reaches HBM (-1.3 GB written and read back per 320 maps, and the
    gather kernel's two launches for the head).  None when the head has more than 16 classes (one padded channel tile) or a class count that
    is not a multiple of 4 (the chained fp32 epilogue stores whole float4s per k-slot quarter; v2x_conv2d(halo) requires Cout2 % 4 == 0): such
    heads keep the two-layer path."""
    import torch
    n_cls = outc.conv.out_channels
    if n_cls > 16 or n_cls % 4 != 0 or decoder.conv8_2.out_channels != 32 or decoder.conv8_2.in_channels != 32:
        return None
    s1, t1 = packing.fold_bn(decoder.conv8_2.bias, decoder.bn8_2, 32)
    w2 = outc.conv.weight.detach().float().cpu()
    b2 = outc.conv.bias.detach().float().cpu() if outc.conv.bias is not None else torch.zeros(n_cls)
    return packing.pack_conv_halo("decoder.conv8_2+outc", decoder.conv8_2.weight, s1, t1, relu=True, chain=(w2, torch.ones(n_cls), b2, False),
                                  epilogue=V2X_EPI_F32, device=device)


def _seg_tail(pk, feats):
    """Decoder + class head -> fp32 NHWC logits; conv8_2 and the head as one launch where the map tiles (tuning switch SEG_FUSE)."""
    fused = pk.get("seg_fused")
    x0 = feats[0]
    if fused is not None and tuning.get("SEG_FUSE") != 0 and ops.halo_eligible(x0.shape[1], x0.shape[2], 1, 32):
        return ops.conv2d(fused, LidarDecoder.run(pk["dec"], *feats, last=False))
    return ops.conv2d(pk["seg"], LidarDecoder.run(pk["dec"], *feats))


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
        pk["seg_fused"] = _pack_seg_fused(self.decoder, self.outc, device)
        return pk

    @ops.latency_entry
    def forward_nhwc(self, x0, trans_matrices, num_agent_tensor, batch_size=1, plan=None, zbits=0):
        """x0: (A*B, X, Y, 32) bf16 NHWC, or -- zbits > 0 -- the voxeliser's int32 bit grid (A*B, X, Y) with zbits height bins."""
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0, zbits=zbits)
        if plan is None:
            plan = self.make_plan(num_agent_tensor, batch_size, x0.device)
        feats[self.layer] = self.fuse(feats[self.layer], trans_matrices, plan, batch_size, pk)
        return _seg_tail(pk, feats)


class FaFNetSeg(FaFNet):
    def __init__(self, config, n_classes=8, **kw):
        super().__init__(config, **kw)
        self.n_classes = n_classes
        self.outc = OutConv(32, n_classes)

    def _pack(self, device):
        pk = super()._pack(device)
        pk["seg"] = _pack_seg_head(self.outc, device)
        pk["seg_fused"] = _pack_seg_fused(self.stpn.decoder, self.outc, device)
        return pk

    @ops.latency_entry
    def forward_nhwc(self, x0, zbits=0):
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0, zbits=zbits)
        return _seg_tail(pk, feats)
