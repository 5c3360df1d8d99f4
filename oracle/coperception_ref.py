"""ORACLE (test infrastructure, NOT product code) -- PyTorch-CPU fp32 restatement
of the coperception detection / segmentation baselines that V2X-Sim ships as
its benchmark.

PARITY UNPINNED.  /root/reference contains only README.md and .gitmodules; every
model named at /root/reference/README.md:101 ("when2com, who2com, V2VNet,
lowerbound and upperbound ... coperception/tools/det, tools/seg") lives in the
un-vendored, un-pinned submodule declared at /root/reference/.gitmodules:1-3
(https://github.com/coperception/coperception.git, commit unknown) and in its
dependency `convolutional_rnn` (Conv2dGRU).  Nothing can be imported, compiled
or diffed here, and the reference ships no tests or golden vectors.  The classes
below restate the *published* architecture of those files from recollection:

  upstream path (no line numbers available)            restated here as
  ---------------------------------------------------  -------------------------
  coperception/models/det/backbone/Backbone.py         LidarEncoder, LidarDecoder
  coperception/models/det/base/DetModelBase.py         DetModelBase (+ heads)
     ::feature_transformation                          feature_transformation
  coperception/models/det/FaFNet.py                    FaFNet
  coperception/models/det/V2VNet.py                    V2VNet
  convolutional_rnn (Conv2dGRU, GRUCell)               Conv2dGRUCell
  coperception/models/det/When2com.py                  When2com, PolicyNet4,
                                                       KmGenerator, MIMOGeneral...
  coperception/models/seg/*                            V2VNetSeg / FaFNetSeg (build-
                                                       owned: det backbone + 1x1 head)

The framework's op semantics follow /root/reference/README.md:88-95 (PyTorch
1.8): F.interpolate default 'nearest', grid_sample/affine_grid with
align_corners=False, bilinear, zero padding.

`emulate_bf16=True` re-runs the same graph with the storage roundings of the
HIP path (bf16 weights and activations, fp32 accumulation, BN folded to an fp32
scale/shift applied after the accumulation); it is the tight comparator for the
kernels, while the plain fp32 graph is the spec the tolerance is stated against.

Only tests/, __graft_entry__.smoke() and bench.py's baseline legs (`cpu_baseline`,
and since round 6 `gpu_stock_baseline`: this same graph moved to cuda:0 as the
same-node stock-PyTorch comparator, outside the timed region) may import this
module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# frozen config (upstream coperception/configs/Config.py)
# ----------------------------------------------------------------------------
MAP_DIMS = (256, 256, 13)
CATEGORY_NUM = 2
BOX_CODE_SIZE = 6
ANCHOR_SIZE = (
    (2.0, 4.0, 0.0), (2.0, 4.0, math.pi / 2.0), (2.0, 4.0, -math.pi / 4.0),
    (3.0, 12.0, 0.0), (3.0, 12.0, math.pi / 2.0), (3.0, 12.0, -math.pi / 4.0),
)
SEG_CLASSES = 8
LAYER_SHAPES = {0: (32, 256, 256), 1: (64, 128, 128), 2: (128, 64, 64), 3: (256, 32, 32), 4: (512, 16, 16)}


def _q(x, emulate):
    """bf16 storage rounding (round-to-nearest-even) when emulating the HIP path."""
    return x.to(torch.bfloat16).to(torch.float32) if emulate else x


def _bn_fold(conv_bias, bn):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    b = conv_bias if conv_bias is not None else torch.zeros_like(bn.running_mean)
    t = bn.bias + s * (b - bn.running_mean)
    return s, t


def cbr(x, conv, bn, emulate=False, relu=True):
    """conv -> eval-mode batch-norm -> ReLU.  x is NCHW fp32."""
    if not emulate:
        y = bn(conv(x))
        return F.relu(y) if relu else y
    s, t = _bn_fold(conv.bias, bn)
    y = F.conv2d(_q(x, True), _q(conv.weight, True), None, conv.stride, conv.padding)
    y = y * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    if relu:
        y = F.relu(y)
    return _q(y, True)


def conv_linear(x, conv, emulate=False):
    """plain conv + bias, fp32 result (head logits are never rounded)."""
    if not emulate:
        return conv(x)
    y = F.conv2d(_q(x, True), _q(conv.weight, True), None, conv.stride, conv.padding)
    return y + conv.bias.view(1, -1, 1, 1)


# ----------------------------------------------------------------------------
# backbone  (upstream Backbone.py: LidarEncoder / LidarDecoder / Conv3D)
# ----------------------------------------------------------------------------
class Conv3D(nn.Module):
    """1x1x1 temporal conv + BN3d + ReLU over a length-1 sequence."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv3d = nn.Conv3d(cin, cout, kernel_size=(1, 1, 1), stride=1, padding=(0, 0, 0))
        self.bn3d = nn.BatchNorm3d(cout)

    def forward(self, x, emulate=False):
        # x: (batch*seq, c, h, w) with seq == 1
        if not emulate:
            y = x.unsqueeze(2)  # (b, c, 1, h, w)
            y = F.relu(self.bn3d(self.conv3d(y)))
            return y.squeeze(2)
        w2 = self.conv3d.weight[:, :, 0]  # (co, ci, 1, 1)
        bn = self.bn3d
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        t = bn.bias + s * (self.conv3d.bias - bn.running_mean)
        y = F.conv2d(_q(x, True), _q(w2, True)) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
        return _q(F.relu(y), True)


class LidarEncoder(nn.Module):
    def __init__(self, height_feat_size=13):
        super().__init__()
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

    def forward(self, x, emulate=False):
        # x: (batch, seq=1, z, h, w)
        x = x.reshape(-1, x.size(-3), x.size(-2), x.size(-1)).to(torch.float)
        e = emulate
        x = cbr(x, self.conv_pre_1, self.bn_pre_1, e)
        x = cbr(x, self.conv_pre_2, self.bn_pre_2, e)
        x_1 = cbr(x, self.conv1_1, self.bn1_1, e)
        x_1 = cbr(x_1, self.conv1_2, self.bn1_2, e)
        x_1 = self.conv3d_1(x_1, e)
        x_2 = cbr(x_1, self.conv2_1, self.bn2_1, e)
        x_2 = cbr(x_2, self.conv2_2, self.bn2_2, e)
        x_2 = self.conv3d_2(x_2, e)
        x_3 = cbr(x_2, self.conv3_1, self.bn3_1, e)
        x_3 = cbr(x_3, self.conv3_2, self.bn3_2, e)
        x_4 = cbr(x_3, self.conv4_1, self.bn4_1, e)
        x_4 = cbr(x_4, self.conv4_2, self.bn4_2, e)
        return [x, x_1, x_2, x_3, x_4]


class LidarDecoder(nn.Module):
    def __init__(self):
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

    def forward(self, x, x_1, x_2, x_3, x_4, emulate=False):
        e = emulate
        up = lambda t: F.interpolate(t, scale_factor=(2, 2))  # nearest
        x_5 = cbr(torch.cat((up(x_4), x_3), dim=1), self.conv5_1, self.bn5_1, e)
        x_5 = cbr(x_5, self.conv5_2, self.bn5_2, e)
        x_6 = cbr(torch.cat((up(x_5), x_2), dim=1), self.conv6_1, self.bn6_1, e)
        x_6 = cbr(x_6, self.conv6_2, self.bn6_2, e)
        x_7 = cbr(torch.cat((up(x_6), x_1), dim=1), self.conv7_1, self.bn7_1, e)
        x_7 = cbr(x_7, self.conv7_2, self.bn7_2, e)
        x_8 = cbr(torch.cat((up(x_7), x), dim=1), self.conv8_1, self.bn8_1, e)
        x_8 = cbr(x_8, self.conv8_2, self.bn8_2, e)
        return x_8


class STPN(nn.Module):
    """Encoder+decoder in one module (upstream STPN_KD used by FaFNet)."""

    def __init__(self, height_feat_size=13):
        super().__init__()
        self.encoder = LidarEncoder(height_feat_size)
        self.decoder = LidarDecoder()

    def forward(self, x, emulate=False):
        return self.decoder(*self.encoder(x, emulate), emulate=emulate)


# ----------------------------------------------------------------------------
# heads  (upstream DetModelBase.py: ClassificationHead, SingleRegressionHead)
# ----------------------------------------------------------------------------
class ClassificationHead(nn.Module):
    def __init__(self, channel=32, category_num=CATEGORY_NUM, anchors=len(ANCHOR_SIZE)):
        super().__init__()
        self.conv1 = nn.Conv2d(channel, channel, 3, 1, 1)
        self.conv2 = nn.Conv2d(channel, category_num * anchors, 1, 1, 0)
        self.bn1 = nn.BatchNorm2d(channel)

    def forward(self, x, emulate=False):
        return conv_linear(cbr(x, self.conv1, self.bn1, emulate), self.conv2, emulate)


class SingleRegressionHead(nn.Module):
    def __init__(self, channel=32, anchors=len(ANCHOR_SIZE), box_code_size=BOX_CODE_SIZE, out_seq_len=1):
        super().__init__()
        self.box_prediction = nn.Sequential(
            nn.Conv2d(channel, channel, 3, 1, 1), nn.BatchNorm2d(channel), nn.ReLU(),
            nn.Conv2d(channel, anchors * box_code_size * out_seq_len, 1, 1, 0))

    def forward(self, x, emulate=False):
        bp = self.box_prediction
        return conv_linear(cbr(x, bp[0], bp[1], emulate), bp[3], emulate)


# ----------------------------------------------------------------------------
# spatial warp  (upstream DetModelBase.feature_transformation)
# ----------------------------------------------------------------------------
def feature_transformation(feat, nb_warp, size):
    """Warp neighbour feature `feat` (C,H,W) into the ego frame.

    nb_warp: 4x4 pose of the neighbour w.r.t. the ego (trans_matrices[b, ego, nb]).
    Two-step resample exactly as upstream: rotate about the map centre, then
    translate by (4*T[0,3]/128, -4*T[1,3]/128) in normalised coordinates.
    """
    nb = feat.unsqueeze(0)
    x_trans = (4 * nb_warp[0, 3]) / 128
    y_trans = -(4 * nb_warp[1, 3]) / 128
    theta_rot = torch.tensor([[nb_warp[0, 0], nb_warp[0, 1], 0.0],
                              [nb_warp[1, 0], nb_warp[1, 1], 0.0]]).type(dtype=torch.float).unsqueeze(0)
    # (.to(feat.device): a no-op on the CPU, where this oracle is the checker; bench.py's `gpu_stock_baseline` times the same graph on cuda:0, with
    # the pose matrices kept on the host exactly so that these scalar reads do not synchronise the device)
    grid_rot = F.affine_grid(theta_rot.to(feat.device), size=torch.Size(size), align_corners=False)
    theta_trans = torch.tensor([[1.0, 0.0, x_trans], [0.0, 1.0, y_trans]]).type(dtype=torch.float).unsqueeze(0)
    grid_trans = F.affine_grid(theta_trans.to(feat.device), size=torch.Size(size), align_corners=False)
    warp_rot = F.grid_sample(nb, grid_rot, mode="bilinear", padding_mode="zeros", align_corners=False)
    warp_trans = F.grid_sample(warp_rot, grid_trans, mode="bilinear", padding_mode="zeros", align_corners=False)
    return warp_trans.squeeze(0)


# ----------------------------------------------------------------------------
# model bases
# ----------------------------------------------------------------------------
class DetModelBase(nn.Module):
    def __init__(self, layer=3, in_channels=13, num_agent=5):
        super().__init__()
        self.category_num = CATEGORY_NUM
        self.box_code_size = BOX_CODE_SIZE
        self.anchor_num_per_loc = len(ANCHOR_SIZE)
        self.out_seq_len = 1
        self.classification = ClassificationHead()
        self.regression = SingleRegressionHead()
        self.agent_num = num_agent
        self.layer = layer
        self.emulate_bf16 = False

    def get_cls_loc_result(self, x):
        e = self.emulate_bf16
        cls_preds = self.classification(x, e).permute(0, 2, 3, 1).contiguous()
        cls_preds = cls_preds.view(cls_preds.shape[0], -1, self.category_num)
        loc_preds = self.regression(x, e).permute(0, 2, 3, 1).contiguous()
        loc_preds = loc_preds.view(-1, loc_preds.size(1), loc_preds.size(2), self.anchor_num_per_loc,
                                   self.out_seq_len, self.box_code_size)
        return {"loc": loc_preds, "cls": cls_preds}

    def local_com_mat(self, feat_maps, batch_size):
        """(A*B, C, H, W) agent-major  ->  (B, A, C, H, W)."""
        return torch.stack([feat_maps[batch_size * i: batch_size * (i + 1)] for i in range(self.agent_num)], 1)

    @staticmethod
    def agents_to_batch(feats):
        return torch.cat([feats[:, i] for i in range(feats.shape[1])], 0)


class FaFNet(DetModelBase):
    """lowerbound / upperbound: no fusion; they differ only in the input cloud."""

    def __init__(self, layer=3, in_channels=13, num_agent=5):
        super().__init__(layer, in_channels, num_agent)
        self.stpn = STPN(in_channels)

    def forward(self, bevs, maps=None, vis=None, batch_size=None):
        bevs = bevs.permute(0, 1, 4, 2, 3)
        return self.get_cls_loc_result(self.stpn(bevs, self.emulate_bf16))


class IntermediateModelBase(DetModelBase):
    def __init__(self, layer=3, in_channels=13, num_agent=5):
        super().__init__(layer, in_channels, num_agent)
        self.u_encoder = LidarEncoder(in_channels)
        self.decoder = LidarDecoder()

    def size(self):
        c, h, w = LAYER_SHAPES[self.layer]
        return (1, c, h, w)

    def decode_heads(self, encoded_layers, feat_fuse_mat):
        encoded_layers = list(encoded_layers)
        encoded_layers[self.layer] = feat_fuse_mat
        x = self.decoder(*encoded_layers, emulate=self.emulate_bf16)
        return x


# ----------------------------------------------------------------------------
# V2VNet  (upstream V2VNet.py + convolutional_rnn.Conv2dGRU, one layer, one step)
# ----------------------------------------------------------------------------
class Conv2dGRUCell(nn.Module):
    """Single-layer ConvGRU cell with PyTorch GRU gate order (r, z, n)."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        k = kernel_size
        self.pad = (k - 1) // 2
        self.hidden = out_channels
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * out_channels, in_channels, k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * out_channels, out_channels, k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * out_channels))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * out_channels))
        stdv = 1.0 / math.sqrt(out_channels)
        for p in self.parameters():
            nn.init.uniform_(p, -stdv, stdv)

    def forward(self, x, hx=None, emulate=False):
        # x: (N, Cin, H, W); hx: (N, hidden, H, W) or None (-> zeros)
        if hx is None:
            hx = torch.zeros(x.shape[0], self.hidden, x.shape[2], x.shape[3], dtype=x.dtype, device=x.device)
        if emulate:
            gi = F.conv2d(_q(x, True), _q(self.weight_ih_l0, True), None, 1, self.pad) + self.bias_ih_l0.view(1, -1, 1, 1)
            gh = F.conv2d(_q(hx, True), _q(self.weight_hh_l0, True), None, 1, self.pad) + self.bias_hh_l0.view(1, -1, 1, 1)
        else:
            gi = F.conv2d(x, self.weight_ih_l0, self.bias_ih_l0, 1, self.pad)
            gh = F.conv2d(hx, self.weight_hh_l0, self.bias_hh_l0, 1, self.pad)
        i_r, i_i, i_n = gi.chunk(3, 1)
        h_r, h_i, h_n = gh.chunk(3, 1)
        resetgate = torch.sigmoid(i_r + h_r)
        inputgate = torch.sigmoid(i_i + h_i)
        newgate = torch.tanh(i_n + resetgate * h_n)
        hy = newgate + inputgate * (hx - newgate)
        return _q(hy, emulate)


class V2VNet(IntermediateModelBase):
    def __init__(self, gnn_iter_times=1, layer=3, layer_channel=256, in_channels=13, num_agent=5,
                 neighbor_source="initial"):
        super().__init__(layer, in_channels, num_agent)
        self.layer_channel = layer_channel
        self.gnn_iter_num = gnn_iter_times
        # "initial": every iteration warps the *encoder* features of the neighbours
        # (recollected upstream behaviour); "updated": warps the previous iteration's;
        # "frozen": the third reading (ASSUMPTIONS.md row 25) -- the inner loop takes the EGO map from
        # local_com_mat[b, i] too, so every round recomputes the same update (rounds > 1 are idempotent).
        if neighbor_source not in ("initial", "updated", "frozen"):
            raise ValueError("neighbor_source must be 'initial', 'updated' or 'frozen'")
        self.neighbor_source = neighbor_source
        self.convgru = Conv2dGRUCell(layer_channel * 2, layer_channel, 3)

    def fuse(self, local_com_mat, trans_matrices, num_agent_tensor, batch_size):
        e = self.emulate_bf16
        size = (1,) + tuple(local_com_mat.shape[2:])  # == self.size() on the 256x256 grid
        update = local_com_mat.clone()
        for b in range(batch_size):
            n = int(num_agent_tensor[b, 0])
            feats = [local_com_mat[b, k] for k in range(self.agent_num)]
            for _ in range(self.gnn_iter_num):
                updated = []
                for i in range(n):
                    all_warp = trans_matrices[b, i]
                    nb_list = []
                    for j in range(n):
                        if j != i:
                            src = feats[j] if self.neighbor_source == "updated" else local_com_mat[b, j]
                            nb_list.append(feature_transformation(src, all_warp[j], size))
                    mean_feat = _q(torch.mean(torch.stack(nb_list), dim=0), e)
                    ego = local_com_mat[b, i] if self.neighbor_source == "frozen" else feats[i]
                    cat_feat = torch.cat([ego, mean_feat], dim=0).unsqueeze(0)
                    updated.append(self.convgru(cat_feat, None, e).squeeze(0))
                feats = updated + feats[n:]
            for k in range(n):
                update[b, k] = feats[k]
        return update

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        bevs = bevs.permute(0, 1, 4, 2, 3)
        enc = self.u_encoder(bevs, self.emulate_bf16)
        lcm = self.local_com_mat(enc[self.layer], batch_size)
        upd = self.fuse(lcm, trans_matrices, num_agent_tensor, batch_size)
        x = self.decode_heads(enc, self.agents_to_batch(upd))
        return self.get_cls_loc_result(x)


# ----------------------------------------------------------------------------
# simple fusion baselines  (upstream SumFusion.py / MeanFusion.py / MaxFusion.py / CatFusion.py on FusionBase.py)
# ----------------------------------------------------------------------------
class FusionBase(IntermediateModelBase):
    """Per ego: [ego map, every neighbour's map warped into the ego frame] -> self.fusion(list).  The list starts with
    the ego (upstream: neighbor_feat_list.append(tg_agent) before the neighbour loop) -- recollected, frozen here."""

    def fusion(self, feats):  # pragma: no cover - abstract
        raise NotImplementedError

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        e = self.emulate_bf16
        bevs = bevs.permute(0, 1, 4, 2, 3)
        enc = self.u_encoder(bevs, e)
        lcm = self.local_com_mat(enc[self.layer], batch_size)
        size = (1,) + tuple(lcm.shape[2:])
        update = lcm.clone()
        for b in range(batch_size):
            n = int(num_agent_tensor[b, 0])
            for i in range(n):
                feats = [lcm[b, i]]
                for j in range(n):
                    if j != i:
                        feats.append(feature_transformation(lcm[b, j], trans_matrices[b, i][j], size))
                update[b, i] = self.fusion(feats)
        x = self.decode_heads(enc, self.agents_to_batch(update))
        return self.get_cls_loc_result(x)


class SumFusion(FusionBase):
    def fusion(self, feats):
        return _q(torch.sum(torch.stack(feats), dim=0), self.emulate_bf16)


class MeanFusion(FusionBase):
    def fusion(self, feats):
        return _q(torch.mean(torch.stack(feats), dim=0), self.emulate_bf16)


class MaxFusion(FusionBase):
    def fusion(self, feats):
        return _q(torch.max(torch.stack(feats), dim=0).values, self.emulate_bf16)


class ModulationLayer3(nn.Module):
    def __init__(self, channel=256):
        super().__init__()
        self.conv1_1 = nn.Conv2d(2 * channel, channel, kernel_size=1, stride=1, padding=0)
        self.bn1_1 = nn.BatchNorm2d(channel)

    def forward(self, x, emulate=False):
        return cbr(x, self.conv1_1, self.bn1_1, emulate)


class CatFusion(FusionBase):
    def __init__(self, layer=3, in_channels=13, num_agent=5):
        super().__init__(layer, in_channels, num_agent)
        self.modulation_layer_3 = ModulationLayer3(LAYER_SHAPES[layer][0])

    def fusion(self, feats):
        e = self.emulate_bf16
        mean_feat = _q(torch.mean(torch.stack(feats), dim=0), e)
        cat_feat = torch.cat([feats[0], mean_feat], dim=0).unsqueeze(0)
        return self.modulation_layer_3(cat_feat, e).squeeze(0)


class PixelWeightedFusionSoftmax(nn.Module):
    """upstream DiscoNet.py: 1x1 convs 2C -> 128 -> 32 -> 8 (+BN+ReLU) -> 1 (+ReLU) on cat(ego, neighbour)."""

    def __init__(self, channel=256):
        super().__init__()
        self.conv1_1 = nn.Conv2d(channel * 2, 128, 1)
        self.bn1_1 = nn.BatchNorm2d(128)
        self.conv1_2 = nn.Conv2d(128, 32, 1)
        self.bn1_2 = nn.BatchNorm2d(32)
        self.conv1_3 = nn.Conv2d(32, 8, 1)
        self.bn1_3 = nn.BatchNorm2d(8)
        self.conv1_4 = nn.Conv2d(8, 1, 1)

    def forward(self, x, emulate=False):
        x = cbr(x, self.conv1_1, self.bn1_1, emulate)
        x = cbr(x, self.conv1_2, self.bn1_2, emulate)
        x = cbr(x, self.conv1_3, self.bn1_3, emulate)
        return F.relu(conv_linear(x, self.conv1_4, emulate))


class DiscoNet(FusionBase):
    """DiscoNet's pixel-wise weighted fusion WITHOUT the knowledge-distillation teacher (kd_flag = 0): every source map
    (ego first, then the warped neighbours) gets a per-pixel scalar from PixelWeightedFusionSoftmax(cat(ego, source));
    the fused map is the softmax-over-sources weighted sum.  exp() without max-subtraction, as upstream."""

    def __init__(self, layer=3, in_channels=13, num_agent=5):
        super().__init__(layer, in_channels, num_agent)
        self.pixel_weighted_fusion = PixelWeightedFusionSoftmax(LAYER_SHAPES[layer][0])

    def fusion(self, feats):
        e = self.emulate_bf16
        feats = [feats[0]] + [_q(f, e) for f in feats[1:]]          # the HIP path stores the warped maps in bf16
        ws = [torch.exp(self.pixel_weighted_fusion(torch.cat([feats[0], f], 0).unsqueeze(0), e)[0, 0]) for f in feats]
        total = sum(ws)
        return _q(sum((w / total).unsqueeze(0) * f for w, f in zip(ws, feats)), e)


# ----------------------------------------------------------------------------
# when2com / who2com  (upstream When2com.py)
# ----------------------------------------------------------------------------
class Conv2DBatchNormRelu(nn.Module):
    def __init__(self, cin, cout, k_size, stride, padding):
        super().__init__()
        self.cbr_unit = nn.Sequential(nn.Conv2d(cin, cout, k_size, stride, padding, bias=True),
                                      nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x, emulate=False):
        return cbr(x, self.cbr_unit[0], self.cbr_unit[1], emulate)


class PolicyNet4(nn.Module):
    def __init__(self, in_channels=13):
        super().__init__()
        self.lidar_encoder = LidarEncoder(in_channels)
        self.conv1 = Conv2DBatchNormRelu(512, 512, 3, 1, 1)
        self.conv2 = Conv2DBatchNormRelu(512, 256, 3, 1, 1)
        self.conv3 = Conv2DBatchNormRelu(256, 256, 3, 2, 1)
        self.conv4 = Conv2DBatchNormRelu(256, 256, 3, 1, 1)
        self.conv5 = Conv2DBatchNormRelu(256, 256, 3, 2, 1)

    def forward(self, bevs, emulate=False):
        x = self.lidar_encoder(bevs, emulate)[4]
        for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            x = m(x, emulate)
        return x  # (N, 256, 4, 4)


class KmGenerator(nn.Module):
    def __init__(self, out_size=128, input_feat_sz=16.0):
        super().__init__()
        feat_map_sz = int(input_feat_sz // 4)
        self.n_feat = int(256 * feat_map_sz * feat_map_sz)
        self.fc = nn.Sequential(nn.Linear(self.n_feat, 256), nn.ReLU(inplace=True),
                                nn.Linear(256, 128), nn.ReLU(inplace=True),
                                nn.Linear(128, out_size))

    def forward(self, features_map, emulate=False):
        x = features_map.reshape(-1, self.n_feat)  # NCHW flatten: index = c*16 + h*4 + w
        if not emulate:
            return self.fc(x)
        for idx in (0, 2):
            lin = self.fc[idx]
            x = _q(F.relu(F.linear(_q(x, True), _q(lin.weight, True)) + lin.bias), True)
        lin = self.fc[4]
        return F.linear(_q(x, True), _q(lin.weight, True)) + lin.bias  # fp32, not rounded


class MIMOGeneralDotProductAttention(nn.Module):
    def __init__(self, query_size, key_size):
        super().__init__()
        self.softmax = nn.Softmax(dim=1)
        self.linear = nn.Linear(query_size, key_size)

    def scores(self, qu, k):
        query = self.linear(qu)                         # (b, q_agents, key_size)
        attn_orig = torch.bmm(k, query.transpose(2, 1))  # (b, k_agents, q_agents)
        return self.softmax(attn_orig)                  # softmax over the keys, per query


class When2com(IntermediateModelBase):
    def __init__(self, layer=3, in_channels=13, num_agent=5, key_size=1024, query_size=32,
                 image_size=512, warp_flag=1, sparse=False, attn_index="kq", renormalize=False):
        super().__init__(layer, in_channels, num_agent)
        if sparse:
            raise NotImplementedError("sparsemax attention is out of scope (DESIGN.md section 8)")
        # the two open readings of ASSUMPTIONS.md as switches (defaults = the first reading; the diff against upstream decides):
        #   row 30  attn_index "kq": fused[q] = sum_k attn[b, k, q] val[k -> q];  "qk": the transposed read attn[b, q, k]
        #   row 31  renormalize: 'activated' coefficients divided by their sum over the keys (a query whose keys all fell below the
        #           threshold keeps zeros)
        if attn_index not in ("kq", "qk"):
            raise ValueError("attn_index must be 'kq' or 'qk'")
        self.attn_index, self.renormalize = attn_index, bool(renormalize)
        self.warp_flag = warp_flag
        self.key_size, self.query_size = key_size, query_size
        self.query_key_net = PolicyNet4(in_channels)
        self.key_net = KmGenerator(key_size, image_size / 32)
        self.query_net = KmGenerator(query_size, image_size / 32)
        self.attention_net = MIMOGeneralDotProductAttention(query_size, key_size)

    def coefficients(self, prob_action, inference, training, thres=0.2):
        """(b, k, q) softmax scores -> fusion coefficients + communication rate."""
        if training or inference == "softmax":
            coef = prob_action
        elif inference == "activated":      # when2com
            coef = prob_action * (prob_action > thres).float()
            if getattr(self, "renormalize", False):
                tot = coef.sum(dim=1, keepdim=True)
                coef = torch.where(tot > 0, coef / torch.where(tot > 0, tot, torch.ones_like(tot)), coef)
        elif inference == "argmax_test":    # who2com
            coef = F.one_hot(prob_action.max(dim=1)[1], num_classes=prob_action.shape[1]).float().transpose(1, 2)
        else:
            raise ValueError("Incorrect inference mode")
        count = coef.clone()
        idx = torch.arange(self.agent_num)
        count[:, idx, idx] = 0
        num_connect = torch.nonzero(count).shape[0] / (self.agent_num * count.shape[0])
        return coef, num_connect

    def forward(self, bevs, trans_matrices, num_agent_tensor, maps=None, vis=None, training=True,
                MO_flag=True, inference="activated", batch_size=1):
        e = self.emulate_bf16
        bevs = bevs.permute(0, 1, 4, 2, 3)
        enc = self.u_encoder(bevs, e)
        lcm = self.local_com_mat(enc[self.layer], batch_size)  # (B, A, C, H, W)
        size = (1,) + tuple(lcm.shape[2:])  # == self.size() on the 256x256 grid
        qk_maps = self.query_key_net(bevs, e)
        keys = self.key_net(qk_maps, e)
        querys = self.query_net(qk_maps, e)
        A = self.agent_num
        key_mat = torch.stack([keys[batch_size * i: batch_size * (i + 1)] for i in range(A)], 1)
        query_mat = torch.stack([querys[batch_size * i: batch_size * (i + 1)] for i in range(A)], 1)
        prob_action = self.attention_net.scores(query_mat, key_mat)  # (B, k, q)
        coef, num_connect = self.coefficients(prob_action, inference, training)
        # val[b, k, q] = feature of source k in the frame of target q (zeros for padding agents)
        fused = torch.zeros_like(lcm)
        for b in range(batch_size):
            n = int(num_agent_tensor[b, 0])
            for q in range(n):
                acc = torch.zeros_like(lcm[b, q])
                for k in range(n):
                    if k == q or self.warp_flag != 1:
                        v = lcm[b, k]
                    else:
                        v = feature_transformation(lcm[b, k], trans_matrices[b, q][k], size)
                    acc = acc + (coef[b, k, q] if self.attn_index == "kq" else coef[b, q, k]) * v
                fused[b, q] = _q(acc, e)
        x = self.decode_heads(enc, self.agents_to_batch(fused))
        res = self.get_cls_loc_result(x)
        res["prob_action"] = prob_action
        res["coef"] = coef
        res["num_connect"] = num_connect
        return res


# ----------------------------------------------------------------------------
# segmentation (config 5): build-owned spec -- det backbone + 1x1 head, 8 classes
# ----------------------------------------------------------------------------
class OutConv(nn.Module):
    def __init__(self, channel=32, n_classes=SEG_CLASSES):
        super().__init__()
        self.conv = nn.Conv2d(channel, n_classes, 1)

    def forward(self, x, emulate=False):
        return conv_linear(x, self.conv, emulate)


class V2VNetSeg(V2VNet):
    def __init__(self, n_classes=SEG_CLASSES, **kw):
        super().__init__(**kw)
        self.outc = OutConv(32, n_classes)

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        bevs = bevs.permute(0, 1, 4, 2, 3)
        enc = self.u_encoder(bevs, self.emulate_bf16)
        lcm = self.local_com_mat(enc[self.layer], batch_size)
        upd = self.fuse(lcm, trans_matrices, num_agent_tensor, batch_size)
        x = self.decode_heads(enc, self.agents_to_batch(upd))
        return self.outc(x, self.emulate_bf16)  # (A*B, n_classes, H, W) logits


def confusion_matrix(pred, label, n_classes=SEG_CLASSES):
    """Integer confusion matrix (exact target): rows = label, cols = prediction."""
    idx = label.reshape(-1).to(torch.int64) * n_classes + pred.reshape(-1).to(torch.int64)
    return torch.bincount(idx, minlength=n_classes * n_classes).view(n_classes, n_classes)
