"""Simple intermediate-fusion baselines -- mirrors of upstream coperception/models/det/
{SumFusion,MeanFusion,MaxFusion,CatFusion}.py on base/FusionBase.py (absent from /root/reference; README.md:101 lists
the benchmark family).  Per ego agent: the ego map and every neighbour's map warped into the ego frame are reduced by
sum / mean / max; CatFusion concatenates the ego map with that mean and applies a 1x1 conv + BN + ReLU
(ModulationLayer3).  From recollection upstream's list STARTS with the ego map; frozen here.

MI355X mapping: the whole reduction is ONE warp_fuse launch (modes WSUM with unit coefficients / MEAN / MAX; the ego
term is read unwarped), and CatFusion's concat is the two-source loader of the 1x1 implicit-GEMM layer.
"""
import torch
import torch.nn as nn

from ... import ops, packing
from ..._lib import V2X_FUSE_MAX, V2X_FUSE_MEAN, V2X_FUSE_WSUM
from .base import IntermediateModelBase, LidarDecoder, LidarEncoder, _ParamsOnly


class FusionBase(IntermediateModelBase):
    FUSE_MODE = None

    def __init__(self, config, layer=3, in_channels=13, kd_flag=0, num_agent=5, compress_level=0, only_v2i=False):
        super().__init__(config, layer, in_channels, kd_flag=kd_flag, num_agent=num_agent,
                         compress_level=compress_level, only_v2i=only_v2i)

    def _pack(self, device):
        return {"enc": self.u_encoder.pack("u_encoder.", device), "dec": self.decoder.pack("decoder.", device),
                "heads": self._pack_heads(device)}

    def make_plan(self, num_agent_tensor, batch_size, device):
        A = self.agent_num
        counts, items, rows = self.frame_plan(num_agent_tensor, batch_size, A)
        coef = torch.zeros((len(items), A), dtype=torch.float32)
        for m, (a, f) in enumerate(items):
            coef[m, :counts[f]] = 1.0        # ego included
        full = len(items) == A * batch_size
        return {"items": torch.tensor(items, dtype=torch.int32, device=device), "coef": coef.to(device),
                "rows": None if full else torch.tensor(rows, device=device)}

    def post_fusion(self, ego, fused, pk):
        return fused

    def fuse(self, feat, trans_matrices, plan, batch_size, pk):
        fused = ops.warp_fuse(feat, self.agent_num, batch_size, trans_matrices.to(torch.float32).contiguous(),
                              plan["items"], plan["coef"], self.FUSE_MODE)
        rows = plan["rows"]
        ego = feat if rows is None else feat.index_select(0, rows)
        out = self.post_fusion(ego, fused, pk)
        if rows is None:
            return out
        cur = feat.clone()
        cur.index_copy_(0, rows, out)
        return cur

    def forward_nhwc(self, x0, trans_matrices, num_agent_tensor, batch_size=1, plan=None):
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0)
        if plan is None:
            plan = self.make_plan(num_agent_tensor, batch_size, x0.device)
        feats[self.layer] = self.fuse(feats[self.layer], trans_matrices, plan, batch_size, pk)
        return self.get_cls_loc_result(LidarDecoder.run(pk["dec"], *feats), pk["heads"])

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        return self.forward_nhwc(self._input_nhwc(bevs), trans_matrices, num_agent_tensor, batch_size)


class SumFusion(FusionBase):
    FUSE_MODE = V2X_FUSE_WSUM


class MeanFusion(FusionBase):
    FUSE_MODE = V2X_FUSE_MEAN


class MaxFusion(FusionBase):
    FUSE_MODE = V2X_FUSE_MAX


class ModulationLayer3(_ParamsOnly):
    def __init__(self, channel=256):
        super().__init__()
        self.conv1_1 = nn.Conv2d(2 * channel, channel, kernel_size=1, stride=1, padding=0)
        self.bn1_1 = nn.BatchNorm2d(channel)


class CatFusion(FusionBase):
    FUSE_MODE = V2X_FUSE_MEAN

    def __init__(self, config, layer=3, in_channels=13, kd_flag=0, num_agent=5, compress_level=0, only_v2i=False):
        super().__init__(config, layer, in_channels, kd_flag, num_agent, compress_level, only_v2i)
        self.modulation_layer_3 = ModulationLayer3(self.fusion_shape()[0])

    def _pack(self, device):
        pk = super()._pack(device)
        c = self.fusion_shape()[0]
        m = self.modulation_layer_3
        pk["mod"] = packing.pack_conv_bn("modulation_layer_3.conv1_1", m.conv1_1, m.bn1_1, C0=c, C1=c, device=device)
        return pk

    def post_fusion(self, ego, fused, pk):
        return ops.conv2d(pk["mod"], ego, fused)     # cat(ego, mean) never exists: two-source 1x1 conv
