"""V2VNet -- mirror of upstream coperception/models/det/V2VNet.py (absent from /root/reference;
README.md:101 names the V2VNet benchmark).  Per frame and per ego agent: warp every neighbour's
fusion-layer map into the ego frame, average, concatenate with the ego map and run one
ConvGRU cell step (convolutional_rnn.Conv2dGRU called with hidden=None, i.e. h0 = 0);
repeat gnn_iter_times; decode.

MI355X mapping: the O(B*A^2) python loop of affine_grid/grid_sample launches becomes ONE
warp_fuse launch per GNN round (all frames, all egos), the concat is the two-source loader of
the GRU implicit-GEMM kernel, and the gate math is that kernel's epilogue.
"""
import torch

from ... import ops, packing
from ..._lib import V2X_FUSE_MEAN
from .base import IntermediateModelBase, LidarDecoder, LidarEncoder


class Conv2dGRU(torch.nn.Module):
    """Parameter container with convolutional_rnn.Conv2dGRU's parameter names (one layer)."""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        import math
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.weight_ih_l0 = torch.nn.Parameter(torch.empty(3 * out_channels, in_channels, k, k))
        self.weight_hh_l0 = torch.nn.Parameter(torch.empty(3 * out_channels, out_channels, k, k))
        self.bias_ih_l0 = torch.nn.Parameter(torch.empty(3 * out_channels))
        self.bias_hh_l0 = torch.nn.Parameter(torch.empty(3 * out_channels))
        stdv = 1.0 / math.sqrt(out_channels)
        for p in self.parameters():
            torch.nn.init.uniform_(p, -stdv, stdv)

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("Conv2dGRU is a parameter container; V2VNet.forward runs the HIP GRU kernel")


class V2VNet(IntermediateModelBase):
    def __init__(self, config, gnn_iter_times=1, layer=3, layer_channel=256, in_channels=13, num_agent=5,
                 compress_level=0, only_v2i=False, neighbor_source="initial"):
        super().__init__(config, layer, in_channels, kd_flag=0, num_agent=num_agent,
                         compress_level=compress_level, only_v2i=only_v2i)
        # which maps a GNN round reads (oracle/ASSUMPTIONS.md row 25, three readings of upstream's inner loop): "initial" = neighbours from the
        # encoder maps, ego from the previous round; "updated" = both from the previous round (the paper); "frozen" = both from the encoder maps,
        # i.e. every round recomputes round 1 (idempotent) -- run once here, whatever gnn_iter_times says
        if neighbor_source not in ("initial", "updated", "frozen"):
            raise ValueError("neighbor_source must be 'initial', 'updated' or 'frozen'")
        self.layer_channel = layer_channel
        self.gnn_iter_num = gnn_iter_times
        self.neighbor_source = neighbor_source
        self.convgru = Conv2dGRU(in_channels=layer_channel * 2, out_channels=layer_channel, kernel_size=3)

    def _pack(self, device):
        g = self.convgru
        return {"enc": self.u_encoder.pack("u_encoder.", device),
                "dec": self.decoder.pack("decoder.", device),
                "heads": self._pack_heads(device),
                "gru": ops.Layer(
                    [packing.pack_gru("convgru", g.weight_ih_l0, g.bias_ih_l0, g.bias_hh_l0,
                                      C0=self.layer_channel, C1=self.layer_channel, device=device)],
                    packing.pack_gru_stream("convgru", g.weight_ih_l0, g.bias_ih_l0, g.bias_hh_l0,
                                            C0=self.layer_channel, C1=self.layer_channel, device=device)
                    if (self.layer_channel % 32 == 0 and packing.STREAM_KERNEL) else None, name="convgru")}

    def gnn_rounds(self):
        """Rounds that are actually computed: 'frozen' repeats round 1 bit for bit, so one."""
        return 1 if self.neighbor_source == "frozen" else self.gnn_iter_num

    # ---- fusion stage (rows a3 + a4) ------------------------------------------------------
    def make_plan(self, num_agent_tensor, batch_size, device):
        """Device-side index tensors of the fusion stage; reusable across calls with the same
        num_agent_tensor (bench.py builds it once)."""
        A = self.agent_num
        counts, items, rows = self.frame_plan(num_agent_tensor, batch_size, A)
        if min(counts) < 2:
            # upstream: torch.stack of an empty neighbour list raises for a 1-agent frame
            raise RuntimeError("V2VNet needs >= 2 agents in every frame (stack expects a non-empty TensorList)")
        coef = torch.zeros((len(items), A), dtype=torch.float32)
        for m, (a, f) in enumerate(items):
            for j in range(counts[f]):
                if j != a:
                    coef[m, j] = 1.0
        full = len(items) == A * batch_size
        return {"items": ops.items_tensor(items, A, batch_size, device),
                "coef": coef.to(device), "rows": None if full else torch.tensor(rows, device=device),
                "n_items": len(items)}

    def fuse(self, feat, trans_matrices, plan, batch_size, pk):
        """feat: (A*B, H, W, C) bf16 fusion-layer maps, agent-major -> updated maps, same shape."""
        A = self.agent_num
        trans = trans_matrices.to(torch.float32).contiguous()
        rows = plan["rows"]
        cur = feat
        for _ in range(self.gnn_rounds()):
            src = cur if self.neighbor_source == "updated" else feat
            mean = ops.warp_fuse(src, A, batch_size, trans, plan["items"], plan["coef"], V2X_FUSE_MEAN)
            ego = cur if rows is None else cur.index_select(0, rows)
            h = ops.run_layer(pk["gru"], ego, mean)
            if rows is None:
                cur = h
            else:
                cur = cur.clone()
                cur.index_copy_(0, rows, h)
        return cur

    @ops.latency_entry     # the plain single-GPU entry: small batches take the latency forms (ops.latency_launches)
    def forward_nhwc(self, x0, trans_matrices, num_agent_tensor, batch_size=1, plan=None, zbits=0):
        """x0: (A*B, X, Y, 32) bf16 NHWC, or -- zbits > 0 -- the voxeliser's int32 bit grid (A*B, X, Y) with zbits height bins."""
        pk = self.packed(x0.device)
        feats = LidarEncoder.run(pk["enc"], x0, zbits=zbits)
        if plan is None:
            plan = self.make_plan(num_agent_tensor, batch_size, x0.device)
        feats[self.layer] = self.fuse(feats[self.layer], trans_matrices, plan, batch_size, pk)
        return self.decode_heads(pk, feats)

    def forward_points(self, points, n_pts, trans_matrices, num_agent_tensor, batch_size=1, grid=None, plan=None):
        """points -> logits on ONE GPU, the plain entry for raw sweeps: points (A*B, max_pts, stride >= 3) fp32 agent-major, n_pts (A*B,) int32.
        Voxel scatter (a1) -> bit grid -> forward_nhwc.  Like forward(), a declared latency entry: small batches take the latency forms."""
        grid = grid or ops.VoxelGrid()
        with ops.latency_dispatch():
            bits = ops.voxelize_bits(points, n_pts, grid)
            return self.forward_nhwc(bits, trans_matrices, num_agent_tensor, batch_size, plan=plan, zbits=grid.dims[2])

    def forward(self, bevs, trans_matrices, num_agent_tensor, batch_size=1):
        """bevs (A*B, 1, 256, 256, 13) agent-major; trans_matrices (B, A, A, 4, 4); num_agent_tensor (B, A)."""
        return self.forward_nhwc(self._input_nhwc(bevs), trans_matrices, num_agent_tensor, batch_size)
