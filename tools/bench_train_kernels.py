#!/usr/bin/env python3
"""Row f-3 kernels next to MIOpen: weight gradient, data gradient and forward of the backbone's 3x3 stride-1 layers at the
training batch (2 frames x 5 agents = 10 maps), one FaFNet forward + backward with / without V2X_TRAIN_HIP_CONV=1, and a whole
training step on the fp32 MIOpen graph against the bf16 NHWC HIP graph (V2X_TRAIN_HIP=1, train/hip_graph.py).

    python tools/bench_train_kernels.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from v2x_sim_amd import tuning  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from v2x_sim_amd import ops
    from v2x_sim_amd.train import hip_conv
    dev = torch.device("cuda:0")
    N = 10
    print("%-22s %10s %10s %10s   (us; MIOpen = torch fp32 NCHW autograd pieces)" % ("layer", "wgrad HIP", "wgrad MIO", "dgrad HIP"))
    for name, cin, cout, hw in (("conv1_2 64->64 @128", 64, 64, 128), ("conv2_2 128->128 @64", 128, 128, 64), ("conv3_2 256->256 @32", 256, 256, 32),
                                ("conv6_1 384->128 @64", 384, 128, 64), ("conv7_2 64->64 @128", 64, 64, 128)):
        x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(N, hw, hw, cout, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        xf, dyf = x.float().permute(0, 3, 1, 2).contiguous(), dy.float().permute(0, 3, 1, 2).contiguous()
        t_h = timed(lambda: ops.conv3x3_wgrad(x, dy))
        t_m = timed(lambda: torch.nn.grad.conv2d_weight(xf, w.shape, dyf, stride=1, padding=1))
        layer = hip_conv._packed("dgrad", w, None, dev)
        t_d = timed(lambda: ops.run_layer(layer, dy))
        flops = 2.0 * N * hw * hw * cout * 9 * cin
        print("%-22s %10.1f %10.1f %10.1f   wgrad HIP %.0f TFLOP/s" % (name, t_h, t_m, t_d, flops / t_h / 1e6))
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import FaFNet
    from v2x_sim_amd.train import detection_loss, train_forward
    from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device
    cfg = Config("train")
    model = init_for_training(FaFNet(cfg, kd_flag=0, num_agent=5), seed=0).to(dev).train()
    data = synthetic_batch_on_device(cfg, 2, 5, seed=1, device=dev)

    def step():
        model.zero_grad()
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
        detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0].backward()
    for flag in ("0", "1"):
        tuning.set("TRAIN_HIP_CONV", int(flag))
        print("FaFNet forward + backward, 10 maps, V2X_TRAIN_HIP_CONV=%s: %.1f ms" % (flag, timed(step, 5) / 1e3))
    tuning.set("TRAIN_HIP_CONV", int("0"))
    # the bf16 NHWC graph on the HIP kernels (train/hip_graph.py) against the fp32 MIOpen graph, optimizer step included
    # (the HIP graph re-packs every layer's weights on the GPU after each step -- that cost is inside the number)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def full_step():
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for flag in ("0", "1"):
        tuning.set("TRAIN_HIP", int(flag))
        print("FaFNet training step (fwd + bwd + Adam), 10 maps, V2X_TRAIN_HIP=%s: %.1f ms" % (flag, timed(full_step, 5) / 1e3))
    # V2VNet (the fusion -- warp + ConvGRU -- between the HIP encoder and decoder)
    from v2x_sim_amd.models.det import V2VNet
    vmodel = init_for_training(V2VNet(cfg, num_agent=5), seed=0).to(dev).train()
    vopt = torch.optim.Adam(vmodel.parameters(), lr=1e-4)

    def v_step():
        res = train_forward(vmodel, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        vopt.zero_grad(set_to_none=True)
        loss.backward()
        vopt.step()
    for flag in ("0", "1"):
        tuning.set("TRAIN_HIP", int(flag))
        print("V2VNet training step (fwd + bwd + Adam), 10 maps, V2X_TRAIN_HIP=%s: %.1f ms" % (flag, timed(v_step, 5) / 1e3))
    tuning.set("TRAIN_HIP", int("1"))
    from v2x_sim_amd.train.graph_step import GraphedTrainStep
    opt_c = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True)
    gstep = GraphedTrainStep(model, opt_c, data, 2)
    print("FaFNet training step, 10 maps, HIP graph of train/hip_graph.py replayed as ONE hipGraph (train/graph_step.py): %.1f ms"
          % (timed(lambda: gstep(data), 10) / 1e3))
    ops.PROFILE = []
    full_step()
    torch.cuda.synchronize()
    agg = {}
    for name, flops, nbytes, e0, e1, layer in ops.PROFILE:
        t = e0.elapsed_time(e1) * 1e3
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += t
    ops.PROFILE = None
    print("HIP launches of one training step (us, by kernel family):")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-58s x%-3d %9.1f" % (k, n, t))
    print("  total %.1f ms in %d launches" % (sum(v[1] for v in agg.values()) / 1e3, sum(v[0] for v in agg.values())))


if __name__ == "__main__":
    main()
