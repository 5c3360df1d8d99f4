"""Host-side mirror of upstream coperception/models/det/backbone/Backbone.py and
coperception/models/det/base/{DetModelBase,IntermediateModelBase,NonIntermediateModelBase}.py
(not present in /root/reference; README.md:101 names the benchmarks they implement).

The nn.Module tree below exists to hold parameters under the upstream attribute names
(`u_encoder.conv1_1.weight`, `decoder.bn5_1.running_var`, `classification.conv1.weight`,
`regression.box_prediction.0.weight`, ...) so upstream-style checkpoints / optimizers see the
same state_dict.  None of these torch modules is ever *called*: forward() packs the
parameters once (packing.py) and runs the hand-written HIP kernels through ops.py.  On a
machine without the MI355X library the forward raises -- there is no eager fallback.
"""
import torch
import torch.nn as nn

from ... import ops, packing
from ...ops import V2X_EPI_BF16, V2X_EPI_F32  # noqa: F401

INPUT_C_PAD = 32  # BEV height bins (13) zero-padded to one MFMA k-step of channels
LAYER_SHAPES = {0: (32, 256, 256), 1: (64, 128, 128), 2: (128, 64, 64), 3: (256, 32, 32), 4: (512, 16, 16)}


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("%s is a parameter container; the compute path is the HIP engine of the owning model"
                           % type(self).__name__)


class Conv3D(_ParamsOnly):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv3d = nn.Conv3d(cin, cout, kernel_size=(1, 1, 1), stride=1, padding=(0, 0, 0))
        self.bn3d = nn.BatchNorm3d(cout)


class LidarEncoder(_ParamsOnly):
    """13 -> 32,32 @256^2 -> 64,64 (+1x1) @128^2 -> 128,128 (+1x1) @64^2 -> 256,256 @32^2 -> 512,512 @16^2."""

    def __init__(self, height_feat_size=13):
        super().__init__()
        self.height_feat_size = height_feat_size
        self.conv_pre_1 = nn.Conv2d(height_feat_size, 32, 3, 1, 1)
        self.conv_pre_2 = nn.Conv2d(32, 32, 3, 1, 1)
        self.bn_pre_1 = nn.BatchNorm2d(32)
        self.bn_pre_2 = nn.BatchNorm2d(32)
        self.conv3d_1 = Conv3D(64, 64)
        self.conv3d_2 = Conv3D(128, 128)
        self.conv1_1 = nn.Conv2d(32, 64, 3, 2, 1)
        self.conv1_2 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv2_1 = nn.Conv2d(64, 128, 3, 2, 1)
        self.conv2_2 = nn.Conv2d(128, 128, 3, 1, 1)
        self.conv3_1 = nn.Conv2d(128, 256, 3, 2, 1)
        self.conv3_2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.conv4_1 = nn.Conv2d(256, 512, 3, 2, 1)
        self.conv4_2 = nn.Conv2d(512, 512, 3, 1, 1)
        for n, c in (("1_1", 64), ("1_2", 64), ("2_1", 128), ("2_2", 128),
                     ("3_1", 256), ("3_2", 256), ("4_1", 512), ("4_2", 512)):
            setattr(self, "bn" + n, nn.BatchNorm2d(c))

    def pack(self, prefix, device):
        """-> list of ops.Layer.  Full-resolution layers carry a halo-kernel packing next to the
        gather-kernel one; conv1_2 + the 1x1x1 conv3d_1 are one fused layer when the halo kernel runs."""
        L = packing.layer_conv_bn
        cin_pad = INPUT_C_PAD
        plan = [L(prefix + "conv_pre_1", self.conv_pre_1, self.bn_pre_1, cin_pad=cin_pad, device=device),
                L(prefix + "conv_pre_2", self.conv_pre_2, self.bn_pre_2, device=device)]
        levels = [plan]
        for lvl in ("1", "2", "3", "4"):
            c1, b1 = getattr(self, "conv%s_1" % lvl), getattr(self, "bn%s_1" % lvl)
            c2, b2 = getattr(self, "conv%s_2" % lvl), getattr(self, "bn%s_2" % lvl)
            stage = [L(prefix + "conv%s_1" % lvl, c1, b1, device=device)]
            if lvl == "1":
                # 64 -> 64 3x3, then the 1x1x1 "Conv3D" 64 -> 64: chained in the streamed kernel's epilogue
                c3 = self.conv3d_1
                fb = [packing.pack_conv_bn(prefix + "conv1_2", c2, b2, device=device),
                      packing.pack_conv_bn(prefix + "conv3d_1", c3.conv3d, c3.bn3d, device=device)]
                s1, t1 = packing.fold_bn(c2.bias, b2, c2.out_channels)
                s2, t2 = packing.fold_bn(c3.conv3d.bias, c3.bn3d, c3.conv3d.out_channels)
                # resident-weights 8-wave ping-pong halo kernel, 1x1 chained in its store phase
                halo = packing.pack_conv_halo(prefix + "conv1_2+conv3d_1", c2.weight, s1, t1, relu=True,
                                              chain=(c3.conv3d.weight[:, :, 0], s2, t2, True), device=device)
                stage.append(ops.Layer(fb, halo, name=prefix + "conv1_2+conv3d_1"))
            elif lvl == "2" and packing.CHAIN_STREAM and packing.STREAM_KERNEL:
                # 128 -> 128 3x3, then the 1x1x1 "Conv3D" 128 -> 128: chained in the streamed kernel's epilogue (the
                # intermediate map never reaches HBM: -2 x 1 MB per map and one launch less)
                c3 = self.conv3d_2
                fb = [packing.pack_conv_bn(prefix + "conv2_2", c2, b2, device=device),
                      packing.pack_conv_bn(prefix + "conv3d_2", c3.conv3d, c3.bn3d, device=device)]
                s1, t1 = packing.fold_bn(c2.bias, b2, c2.out_channels)
                s2, t2 = packing.fold_bn(c3.conv3d.bias, c3.bn3d, c3.conv3d.out_channels)
                halo = packing.pack_conv_stream(prefix + "conv2_2+conv3d_2", c2.weight, s1, t1, relu=True,
                                                chain=(c3.conv3d.weight[:, :, 0], s2, t2, True), device=device)
                stage.append(ops.Layer(fb, halo, name=prefix + "conv2_2+conv3d_2"))
            else:
                stage.append(L(prefix + "conv%s_2" % lvl, c2, b2, device=device))
                if lvl == "2":
                    c3 = self.conv3d_2
                    stage.append(ops.Layer([packing.pack_conv_bn(prefix + "conv3d_2", c3.conv3d, c3.bn3d, device=device)]))
            levels.append(stage)
        return levels

    @staticmethod
    def run(levels, x, zbits=0):
        """x: (N, X, Y, INPUT_C_PAD) bf16 NHWC, or the voxelizer's int32 bit grid (N, X, Y) with zbits height bins
        (conv_pre_1 then expands the bits while filling its LDS patch) -> [x, x_1, x_2, x_3, x_4]."""
        feats = []
        for stage in levels:
            k = 0
            if stage is levels[0] and len(stage) == 2 and ops.pair_eligible(stage[0].halo, stage[1].halo, x, zbits):
                # conv_pre_1 -> conv_pre_2 in one launch, the 32-channel intermediate stays in LDS (conv_halo_pair.hip)
                x = ops.conv2d_pair(stage[0].halo, stage[1].halo, x, zbits)
                k = 2
            for layer in stage[k:]:
                x = ops.run_layer(layer, x, zbits=zbits if x.dtype == torch.int32 else 0)
            feats.append(x)
        return feats


class LidarDecoder(_ParamsOnly):
    """4 x [nearest x2 upsample, concat skip, conv+BN+ReLU x2]; the upsample+concat never exists
    in memory -- the first conv of each level reads both sources directly."""

    def __init__(self, height_feat_size=13):
        super().__init__()
        self.conv5_1 = nn.Conv2d(512 + 256, 256, 3, 1, 1)
        self.conv5_2 = nn.Conv2d(256, 256, 3, 1, 1)
        self.conv6_1 = nn.Conv2d(256 + 128, 128, 3, 1, 1)
        self.conv6_2 = nn.Conv2d(128, 128, 3, 1, 1)
        self.conv7_1 = nn.Conv2d(128 + 64, 64, 3, 1, 1)
        self.conv7_2 = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv8_1 = nn.Conv2d(64 + 32, 32, 3, 1, 1)
        self.conv8_2 = nn.Conv2d(32, 32, 3, 1, 1)
        for n, c in (("5_1", 256), ("5_2", 256), ("6_1", 128), ("6_2", 128),
                     ("7_1", 64), ("7_2", 64), ("8_1", 32), ("8_2", 32)):
            setattr(self, "bn" + n, nn.BatchNorm2d(c))

    def pack(self, prefix, device):
        L = packing.layer_conv_bn
        plan = []
        for lvl, (cup, cskip) in (("5", (512, 256)), ("6", (256, 128)), ("7", (128, 64)), ("8", (64, 32))):
            plan.append(L(prefix + "conv%s_1" % lvl, getattr(self, "conv%s_1" % lvl), getattr(self, "bn%s_1" % lvl),
                          C0=cup, C1=cskip, up0=1, device=device))
            plan.append(L(prefix + "conv%s_2" % lvl, getattr(self, "conv%s_2" % lvl), getattr(self, "bn%s_2" % lvl),
                          device=device))
        return plan

    @staticmethod
    def run(plan, x, x_1, x_2, x_3, x_4, last=True):
        """last=False stops before conv8_2 (DetModelBase.decode_heads then runs conv8_2 + the heads as one launch)."""
        it = iter(plan)
        y = x_4
        for skip in (x_3, x_2, x_1, x):
            y = ops.run_layer(next(it), y, skip)
            if skip is x and not last:
                return y
            y = ops.run_layer(next(it), y)
        return y


class ClassificationHead(_ParamsOnly):
    def __init__(self, config):
        super().__init__()
        channel = 32
        self.conv1 = nn.Conv2d(channel, channel, kernel_size=3, stride=1, padding=1)
        self.conv2 = nn.Conv2d(channel, config.category_num * len(config.anchor_size), kernel_size=1, stride=1,
                               padding=0)
        self.bn1 = nn.BatchNorm2d(channel)


class SingleRegressionHead(_ParamsOnly):
    def __init__(self, config):
        super().__init__()
        channel = 32
        out_seq_len = 1 if config.only_det else config.pred_len
        self.box_prediction = nn.Sequential(
            nn.Conv2d(channel, channel, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(channel), nn.ReLU(),
            nn.Conv2d(channel, len(config.anchor_size) * config.box_code_size * out_seq_len, kernel_size=1,
                      stride=1, padding=0))


class DetModelBase(nn.Module):
    """Abstract detection model (upstream DetModelBase): heads + agent<->batch helpers + the
    packed-parameter cache shared by every concrete model."""

    def __init__(self, config, layer=3, in_channels=13, kd_flag=True, p_com_outage=0.0, num_agent=5,
                 only_v2i=False):
        super().__init__()
        if kd_flag not in (0, False):
            raise NotImplementedError("knowledge distillation (DiscoNet teacher) is out of scope (DESIGN.md s8)")
        self.motion_state = config.motion_state
        self.out_seq_len = 1 if config.only_det else config.pred_len
        self.box_code_size = config.box_code_size
        self.category_num = config.category_num
        self.use_map = config.use_map
        self.anchor_num_per_loc = len(config.anchor_size)
        self.classification = ClassificationHead(config)
        self.regression = SingleRegressionHead(config)
        self.agent_num = num_agent
        self.kd_flag = kd_flag
        self.layer = layer
        self.in_channels = in_channels
        self._packed = None
        self._packed_key = None

    # ---- packed-parameter cache -----------------------------------------------------------
    def _pack(self, device):  # pragma: no cover - overridden
        raise NotImplementedError

    def packed(self, device=None):
        if device is None:
            device = next(self.parameters()).device
        if torch.device(device).type != "cuda":
            raise RuntimeError("v2x_sim_amd models run on the MI355X only: move the model to 'cuda' "
                               "(no CPU fallback exists in the product path)")
        key = (str(device), sum(t._version + getattr(t, "_v2x_epoch", 0) for t in self.parameters())
               + sum(t._version + getattr(t, "_v2x_epoch", 0) for t in self.buffers()))   # _v2x_epoch: packing.watch_optimizer (fused optimizers do not bump _version)
        if self._packed is None or self._packed_key != key:
            with torch.no_grad():
                self._packed = self._pack(device)
            self._packed_key = key
        return self._packed

    def repack(self):
        self._packed = None

    # ---- shared pieces --------------------------------------------------------------------
    def _pack_heads(self, device):
        """cls | reg hidden 3x3 (32 -> 64) chained with the block-diagonal 1x1 (64 -> 12 + 36): one halo
        launch writes both fp32 logit tensors; the gather-kernel pair is the fallback for odd extents."""
        bp = self.regression.box_prediction
        c = self.classification
        hidden, final, ncls = packing.pack_heads("heads", c.conv1, c.bn1, c.conv2, bp[0], bp[1], bp[3], device=device)
        s1, t1 = packing.fold_bn(c.conv1.bias, c.bn1, c.conv1.out_channels)
        s2, t2 = packing.fold_bn(bp[0].bias, bp[1], bp[0].out_channels)
        w1 = torch.cat([c.conv1.weight.detach().float().cpu(), bp[0].weight.detach().float().cpu()], 0)
        hc, hr = c.conv1.out_channels, bp[0].out_channels
        nreg = bp[3].out_channels
        w2 = torch.zeros((ncls + nreg, hc + hr, 1, 1), dtype=torch.float32)
        w2[:ncls, :hc] = c.conv2.weight.detach().float().cpu()
        w2[ncls:, hc:] = bp[3].weight.detach().float().cpu()
        b2 = torch.cat([c.conv2.bias.detach().float().cpu(), bp[3].bias.detach().float().cpu()])
        halo = None
        if (hc + hr, ncls + nreg) == (64, 48):
            halo = packing.pack_conv_halo("heads.fused", w1, torch.cat([s1, s2]), torch.cat([t1, t2]), relu=True,
                                          chain=(w2, torch.ones(ncls + nreg), b2, False), epilogue=V2X_EPI_F32,
                                          device=device)
        layer = ops.Layer([hidden, final], halo, split=ncls, name="heads")
        # the same heads with the score threshold in the epilogue (detections without the logits round trip): DetModelBase.detections()
        layer.det = packing.pack_heads_det("heads.det", c.conv1, c.bn1, c.conv2, bp[0], bp[1], bp[3], device=device) if halo is not None else None
        return layer

    def _input_nhwc(self, bevs):
        """(N, 1, X, Y, Z) fp32 dense BEV (the reference Dataset format) -> (N, X, Y, c_pad) bf16."""
        if bevs.dim() != 5 or bevs.shape[1] != 1:
            raise ValueError("bevs must be (batch*agents, 1, X, Y, Z); got %s" % (tuple(bevs.shape),))
        return ops.dense_to_nhwc(bevs[:, 0].to(torch.float32).contiguous(), INPUT_C_PAD)

    def decode_heads(self, pk, feats):
        """Decoder + heads on the (possibly fused) pyramid `feats` -> {'loc', 'cls'}; inside `with model.detections(thr, cap)` and when
        the extent allows -> {'det': (keys, codes, counts)}: the candidates of apply_nms_det's threshold step straight from the heads'
        epilogue (ops.conv2d_det), the logits never written."""
        req = getattr(self, "_det_request", None)
        heads, last = pk["heads"], pk["dec"][-1]
        if req is None and heads.halo is not None and last.halo is not None:
            # logits: conv8_2 and the heads as ONE launch when the extent allows (conv_tail.hip; bit-identical to the two launches)
            x = LidarDecoder.run(pk["dec"], *feats, last=False)
            if ops.tail_eligible(last.halo, heads.halo, x, heads.split):
                cls, loc = ops.conv2d_tail(last.halo, heads.halo, x, heads.split)
                return self._shape_cls_loc(cls, loc)
            x = ops.run_layer(last, x)
        else:
            x = LidarDecoder.run(pk["dec"], *feats)
        det = getattr(pk["heads"], "det", None)
        if (req is not None and det is not None and x.shape[1] % 8 == 0 and x.shape[2] % 32 == 0 and x.shape[1] * x.shape[2] * 6 < (1 << 20)
                and (self.anchor_num_per_loc, self.category_num, self.box_code_size, self.out_seq_len) == (6, 2, 6, 1)):
            return {"det": ops.conv2d_det(det, x, req[0], req[1])}
        return self.get_cls_loc_result(x, pk["heads"])

    def detections(self, score_thr, cap=4096):
        """Context manager: forward() calls inside return candidates instead of logits (see decode_heads)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            prev = getattr(self, "_det_request", None)
            self._det_request = (float(score_thr), int(cap))
            try:
                yield self
            finally:
                self._det_request = prev
        return _cm()

    def _shape_cls_loc(self, cls, loc):
        n = cls.shape[0]
        return {"loc": loc.view(-1, loc.size(1), loc.size(2), self.anchor_num_per_loc, self.out_seq_len, self.box_code_size),
                "cls": cls.view(n, -1, self.category_num)}

    def get_cls_loc_result(self, x, heads):
        cls, loc = ops.run_layer(heads, x)  # fp32 NHWC == upstream's permute(0, 2, 3, 1)
        n = cls.shape[0]
        cls_preds = cls.view(n, -1, self.category_num)
        loc_preds = loc.view(-1, loc.size(1), loc.size(2), self.anchor_num_per_loc, self.out_seq_len,
                             self.box_code_size)
        return {"loc": loc_preds, "cls": cls_preds}

    @staticmethod
    def agents_to_batch(feats):
        return torch.cat([feats[:, i] for i in range(feats.shape[1])], 0)


class NonIntermediateModelBase(DetModelBase):
    pass


class IntermediateModelBase(DetModelBase):
    """Models that exchange an intermediate feature map (upstream IntermediateModelBase/FusionBase)."""

    def __init__(self, config, layer=3, in_channels=13, kd_flag=True, p_com_outage=0.0, num_agent=5,
                 compress_level=0, only_v2i=False):
        super().__init__(config, layer, in_channels, kd_flag, p_com_outage, num_agent, only_v2i)
        if compress_level != 0 or only_v2i or p_com_outage != 0.0:
            raise NotImplementedError("compress_level / only_v2i / p_com_outage are out of scope (DESIGN.md s8)")
        self.u_encoder = LidarEncoder(in_channels)
        self.decoder = LidarDecoder(in_channels)

    def fusion_shape(self):
        return LAYER_SHAPES[self.layer]

    @staticmethod
    def frame_plan(num_agent_tensor, batch_size, agent_num):
        """Host-side bookkeeping shared by the fusion models.

        num_agent_tensor (B, A) -- [b, 0] is the number of real agents of frame b (upstream
        convention).  Returns python lists: items [(agent, frame)] in agent-major order for every
        real agent, and the flat row of each item in the (A*B) agent-major batch."""
        nat = num_agent_tensor.detach().to("cpu") if isinstance(num_agent_tensor, torch.Tensor) else num_agent_tensor
        counts = [int(nat[b][0]) for b in range(batch_size)]
        for c in counts:
            if c < 1 or c > agent_num:
                raise ValueError("num_agent_tensor[b, 0]=%d outside [1, %d]" % (c, agent_num))
        items = [(a, f) for a in range(agent_num) for f in range(batch_size) if a < counts[f]]
        rows = [a * batch_size + f for a, f in items]
        return counts, items, rows
