"""when2com / who2com -- mirror of upstream coperception/models/det/When2com.py (absent from
/root/reference; README.md:101 names both benchmarks).  A policy tower (its own LidarEncoder
+ 5 convs) feeds two MLPs producing a 1024-d key and a 32-d query per agent; scores =
key . Linear(query), softmax over the keys; at inference 'activated' keeps weights > 0.2
(when2com), 'argmax_test' keeps the top-1 key (who2com); the fused map of agent q is
sum_k coef[k, q] * warp(F_k -> q).

MI355X mapping: tower convs + MLPs run on the implicit-GEMM kernel (the MLPs as 1x1 convs on
1x1 maps), the 5x5 handshake is one wavefront-shuffle kernel per frame, the weighted sum is the
same warp_fuse launch V2VNet uses.  Zero coefficients skip their warp entirely -- the
communication sparsity of when2com becomes skipped work.
"""
import torch
import torch.nn as nn

from ... import ops, packing
from ..._lib import V2X_EPI_BF16, V2X_EPI_F32, V2X_FUSE_WSUM
from .base import IntermediateModelBase, LidarDecoder, LidarEncoder, _ParamsOnly


class Conv2DBatchNormRelu(_ParamsOnly):
    def __init__(self, cin, cout, k_size, stride, padding):
        super().__init__()
        self.cbr_unit = nn.Sequential(nn.Conv2d(cin, cout, k_size, stride, padding, bias=True),
                                      nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class PolicyNet4(_ParamsOnly):
    def __init__(self, in_channels=13, input_feat_sz=32):
        super().__init__()
        self.lidar_encoder = LidarEncoder(in_channels)
        self.conv1 = Conv2DBatchNormRelu(512, 512, 3, 1, 1)
        self.conv2 = Conv2DBatchNormRelu(512, 256, 3, 1, 1)
        self.conv3 = Conv2DBatchNormRelu(256, 256, 3, 2, 1)
        self.conv4 = Conv2DBatchNormRelu(256, 256, 3, 1, 1)
        self.conv5 = Conv2DBatchNormRelu(256, 256, 3, 2, 1)


class KmGenerator(_ParamsOnly):
    def __init__(self, out_size=128, input_feat_sz=32.0):
        super().__init__()
        feat_map_sz = int(input_feat_sz // 4)
        self.feat_map_sz = feat_map_sz
        self.n_feat = int(256 * feat_map_sz * feat_map_sz)
        self.fc = nn.Sequential(nn.Linear(self.n_feat, 256), nn.ReLU(inplace=True),
                                nn.Linear(256, 128), nn.ReLU(inplace=True),
                                nn.Linear(128, out_size))

    def pack(self, prefix, device):
        # upstream flattens NCHW (c*S*S + h*S + w); our maps are NHWC ((h*S + w)*256 + c)
        S = self.feat_map_sz
        hw = torch.arange(S * S).view(S * S, 1)
        c = torch.arange(256).view(1, 256)
        perm = (c * (S * S) + hw).reshape(-1)  # nhwc position -> nchw column
        return [packing.pack_linear(prefix + "fc.0", self.fc[0], relu=True, col_perm=perm, device=device),
                packing.pack_linear(prefix + "fc.2", self.fc[2], relu=True, device=device),
                packing.pack_linear(prefix + "fc.4", self.fc[4], relu=False, epilogue=V2X_EPI_F32, device=device)]

    @staticmethod
    def pack_pair(key_net, query_net, prefix_k, prefix_q, device):
        """The key and the query MLP as ONE chain of three launches instead of two chains of three: both read the same flattened tower output,
        so the first layers' rows are stacked (4096 -> 256 | 256) and the later layers are block-diagonal (512 -> 128 | 128, 256 -> 1024 | 32).
        The zero blocks add exact zeros to the fp32 sums in whole 32-channel K chunks: the results are the separate MLPs' bit for bit.
        -> (plan of three packed layers, key_size)."""
        S = key_net.feat_map_sz
        hw = torch.arange(S * S).view(S * S, 1)
        c = torch.arange(256).view(1, 256)
        perm = (c * (S * S) + hw).reshape(-1)  # nhwc position -> nchw column
        k, q = key_net.fc, query_net.fc
        f32 = lambda t: t.detach().float().cpu()   # noqa: E731

        def blockdiag(a, b):
            w = torch.zeros((a.shape[0] + b.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32)
            w[:a.shape[0], :a.shape[1]] = a
            w[a.shape[0]:, a.shape[1]:] = b
            return w
        layers = [(torch.cat([f32(k[0].weight)[:, perm], f32(q[0].weight)[:, perm]], 0), torch.cat([f32(k[0].bias), f32(q[0].bias)]), True, V2X_EPI_BF16),
                  (blockdiag(f32(k[2].weight), f32(q[2].weight)), torch.cat([f32(k[2].bias), f32(q[2].bias)]), True, V2X_EPI_BF16),
                  (blockdiag(f32(k[4].weight), f32(q[4].weight)), torch.cat([f32(k[4].bias), f32(q[4].bias)]), False, V2X_EPI_F32)]
        plan = [packing.pack_conv("%s+%sfc.%d" % (prefix_k, prefix_q, 2 * i), w[:, :, None, None], torch.ones(w.shape[0]), b, stride=1, pad=0, relu=relu,
                                  epilogue=epi, device=device) for i, (w, b, relu, epi) in enumerate(layers)]
        return plan, k[4].out_features

    @staticmethod
    def run_pair(pair, x):
        """x: (N, S, S, 256) bf16 -> (keys (N, key_size), querys (N, query_size)) fp32, contiguous."""
        plan, key_size = pair
        y = KmGenerator.run(plan, x)
        return y[:, :key_size].contiguous(), y[:, key_size:].contiguous()

    @staticmethod
    def run(plan, x):
        """x: (N, S, S, 256) bf16 -> (N, out_size) fp32."""
        n = x.shape[0]
        y = x.reshape(n, 1, 1, -1)
        for pc in plan:
            y = ops.conv2d(pc, y)
        return y.view(n, -1)


class MIMOGeneralDotProductAttention(_ParamsOnly):
    def __init__(self, query_size, key_size, warp_flag, attn_dropout=0.1):
        super().__init__()
        self.linear = nn.Linear(query_size, key_size)
        self.warp_flag = warp_flag


class When2com(IntermediateModelBase):
    def __init__(self, config, n_classes=21, in_channels=13, feat_channel=512, feat_squeezer=-1,
                 attention="additive", has_query=True, sparse=False, layer=3, warp_flag=1, image_size=512,
                 shared_img_encoder="unified", key_size=1024, query_size=32, num_agent=5, compress_level=0,
                 only_v2i=False, attn_index="kq", renormalize=False):
        super().__init__(config, layer, in_channels, kd_flag=0, num_agent=num_agent,
                         compress_level=compress_level, only_v2i=only_v2i)
        if sparse:
            raise NotImplementedError("sparsemax attention is out of scope (DESIGN.md section 8)")
        if not has_query:
            raise NotImplementedError("has_query=False is out of scope")
        # the two open readings of oracle/ASSUMPTIONS.md as switches (defaults = the first reading): row 30 attn_index "kq" -- fused[q] =
        # sum_k attn[b, k, q] val[k -> q] -- or "qk", the transposed read; row 31 renormalize -- 'activated' coefficients divided by their sum
        # over the keys (a query whose keys all fell below the threshold keeps zeros)
        if attn_index not in ("kq", "qk"):
            raise ValueError("attn_index must be 'kq' or 'qk'")
        self.attn_index, self.renormalize = attn_index, bool(renormalize)
        self.sparse = sparse
        self.warp_flag = warp_flag
        self.key_size, self.query_size = key_size, query_size
        self.query_key_net = PolicyNet4(in_channels=in_channels)
        self.key_net = KmGenerator(out_size=key_size, input_feat_sz=image_size / 32)
        self.query_net = KmGenerator(out_size=query_size, input_feat_sz=image_size / 32)
        self.attention_net = MIMOGeneralDotProductAttention(query_size, key_size, warp_flag)

    def _pack(self, device):
        qk = self.query_key_net
        tower = {"enc": qk.lidar_encoder.pack("query_key_net.lidar_encoder.", device), "convs": []}
        for i in range(1, 6):
            u = getattr(qk, "conv%d" % i).cbr_unit
            tower["convs"].append(packing.layer_conv_bn("query_key_net.conv%d" % i, u[0], u[1], device=device))
        lin = self.attention_net.linear
        return {"enc": self.u_encoder.pack("u_encoder.", device),
                "dec": self.decoder.pack("decoder.", device),
                "heads": self._pack_heads(device),
                "tower": tower,
                "key": self.key_net.pack("key_net.", device),
                "query": self.query_net.pack("query_net.", device),
                "keyquery": KmGenerator.pack_pair(self.key_net, self.query_net, "key_net.", "query_net.", device),
                "w_lin": lin.weight.detach().float().to(device).contiguous(),
                "b_lin": lin.bias.detach().float().to(device).contiguous()}

    def make_plan(self, num_agent_tensor, batch_size, device):
        A = self.agent_num
        counts, items, rows = self.frame_plan(num_agent_tensor, batch_size, A)
        mask = torch.zeros((len(items), A), dtype=torch.float32)
        for m, (a, f) in enumerate(items):
            mask[m, :counts[f]] = 1.0
        full = len(items) == A * batch_size
        it = torch.tensor(items, dtype=torch.int64)
        return {"items": ops.items_tensor(items, A, batch_size, device), "mask": mask.to(device),
                "rows": None if full else torch.tensor(rows, device=device),
                "q_idx": it[:, 0].to(device), "f_idx": it[:, 1].to(device)}

    def handshake(self, x0, pk, batch_size, mode):
        """Policy tower + key/query MLPs + attention scores.  -> prob, coef (B, A_key, A_query)."""
        y = LidarEncoder.run(pk["tower"]["enc"], x0)[4]
        for layer in pk["tower"]["convs"]:
            y = ops.run_layer(layer, y)
        keys, querys = KmGenerator.run_pair(pk["keyquery"], y)     # (the separate MLPs pk["key"], pk["query"] give the same bits: tests)
        return ops.attn_handshake(keys, querys, pk["w_lin"], pk["b_lin"], self.agent_num, batch_size, mode)

    @ops.latency_entry
    def forward_nhwc(self, x0, trans_matrices, num_agent_tensor, training=True, inference="activated",
                     batch_size=1, plan=None):
        pk = self.packed(x0.device)
        A = self.agent_num
        feats = LidarEncoder.run(pk["enc"], x0)
        if plan is None:
            plan = self.make_plan(num_agent_tensor, batch_size, x0.device)
        if training or inference == "softmax":
            mode = "softmax"
        elif inference in ("activated", "argmax_test"):
            mode = inference
        else:
            raise ValueError("Incorrect inference mode")
        prob, coef = self.handshake(x0, pk, batch_size, mode)
        if self.renormalize and mode == "activated":
            coef = self.renormalized(coef)
        # per output item (q, f): coefficient of every source k, zeroed for padding agents
        coef_items = (coef[plan["f_idx"], :, plan["q_idx"]] if self.attn_index == "kq" else coef[plan["f_idx"], plan["q_idx"], :]).contiguous() * plan["mask"]
        if self.warp_flag != 1:
            raise NotImplementedError("warp_flag=0 (no spatial alignment) is out of scope")
        feat = feats[self.layer]
        fused_items = ops.warp_fuse(feat, A, batch_size, trans_matrices.to(torch.float32).contiguous(),
                                    plan["items"], coef_items, V2X_FUSE_WSUM)
        if plan["rows"] is None:
            fused = fused_items
        else:  # padding agents decode zeros, as upstream's zero-initialised val_mat
            fused = torch.zeros_like(feat)
            fused.index_copy_(0, plan["rows"], fused_items)
        feats[self.layer] = fused
        res = self.decode_heads(pk, feats)
        res["prob_action"] = prob
        res["coef"] = coef
        return res

    @staticmethod
    def renormalized(coef):
        """(B, A_key, A_query) thresholded coefficients -> divided by their sum over the keys where that is non-zero (row 31, second reading)."""
        tot = coef.sum(dim=1, keepdim=True)
        return torch.where(tot > 0, coef / torch.where(tot > 0, tot, torch.ones_like(tot)), coef)

    @staticmethod
    def num_connect(coef, agent_num):
        """Communication rate as upstream: off-diagonal non-zero links / (agents * frames)."""
        c = coef.clone()
        idx = torch.arange(agent_num, device=c.device)
        c[:, idx, idx] = 0
        return torch.nonzero(c).shape[0] / (agent_num * c.shape[0])

    def forward(self, bevs, trans_matrices, num_agent_tensor, maps=None, vis=None, training=True, MO_flag=True,
                inference="activated", batch_size=1):
        res = self.forward_nhwc(self._input_nhwc(bevs), trans_matrices, num_agent_tensor, training, inference,
                                batch_size)
        res["num_connect"] = self.num_connect(res["coef"], self.agent_num)
        return res
