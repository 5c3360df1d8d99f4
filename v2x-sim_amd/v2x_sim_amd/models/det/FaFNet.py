"""lowerbound / upperbound detector -- mirror of upstream coperception/models/det/FaFNet.py
(absent from /root/reference; README.md:101 "lowerbound and upperbound").  The two baselines
share this network and differ only in the input cloud (ego-only vs. early-fused)."""
import torch.nn as nn

from ... import ops
from .base import LidarDecoder, LidarEncoder, NonIntermediateModelBase


class STPN_KD(nn.Module):
    """Encoder + decoder parameter container (upstream keeps them in one module for FaFNet)."""

    def __init__(self, height_feat_size=13):
        super().__init__()
        self.encoder = LidarEncoder(height_feat_size)
        self.decoder = LidarDecoder(height_feat_size)


class FaFNet(NonIntermediateModelBase):
    def __init__(self, config, layer=3, in_channels=13, kd_flag=0, num_agent=5, compress_level=0,
                 train_completion=False):
        super().__init__(config, layer, in_channels, kd_flag, num_agent=num_agent)
        self.stpn = STPN_KD(config.map_dims[2])

    def _pack(self, device):
        return {"enc": self.stpn.encoder.pack("stpn.encoder.", device),
                "dec": self.stpn.decoder.pack("stpn.decoder.", device),
                "heads": self._pack_heads(device)}

    @ops.latency_entry     # the plain single-GPU entry: small batches take the latency forms (ops.latency_launches)
    def forward_nhwc(self, x0):
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0)
        return self.decode_heads(pk, feats)

    def forward(self, bevs, maps=None, vis=None, batch_size=None):
        """bevs: (batch*agents, 1, 256, 256, 13) dense occupancy, as the reference Dataset yields."""
        return self.forward_nhwc(self._input_nhwc(bevs))
