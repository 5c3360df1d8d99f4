#!/usr/bin/env python3
"""Which launches of a training step are NOT this library's kernels, and which Python line issues each (row f-3).
One eager FaFNet / V2VNet step under torch.profiler (CPU + device activities, Python stacks); every device kernel / memcpy whose name is not one of
libv2x_amd.so's is attributed to the aten op and the innermost v2x_sim_amd / torch.optim source line above it.
    python tools/train_op_census.py [faf|v2v] [frames]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from v2x_sim_amd import packing, tuning  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import FaFNet, V2VNet  # noqa: E402
from v2x_sim_amd.train import detection_loss, train_forward  # noqa: E402
from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device  # noqa: E402

OURS = ("conv3x3_", "conv1x1_", "bn_", "wgrad_", "upcat_", "cast_pad_", "channel_sum", "det_loss", "pack_conv", "zero_insert", "dense_f32", "warp_affine", "gru_", "voxel",
        "conv_igemm", "conv_gather", "adam_", "splitk_reduce", "v2v_", "vt_sum", "warp_fuse", "attn_", "seg_")


def _ours(name):
    n = name
    for pre in ("void ", "(anonymous namespace)::"):
        if n.startswith(pre):
            n = n[len(pre):]
    if n.startswith("(anonymous namespace)::"):
        n = n[len("(anonymous namespace)::"):]
    return any(n.startswith(p) for p in OURS)


def main(family="faf", frames=2):
    dev = torch.device("cuda:0")
    cfg = Config("train")
    v2v = family == "v2v"
    model = init_for_training(V2VNet(cfg, num_agent=5) if v2v else FaFNet(cfg, kd_flag=0, num_agent=5), seed=0).to(dev).train()
    data = synthetic_batch_on_device(cfg, frames, 5, seed=1, device=dev)
    opt = packing.watch_optimizer(torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True, fused=True))
    tuning.set("TRAIN_HIP", 1)

    def step():
        res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], frames)
        loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    events = prof.events()
    rows = collections.defaultdict(lambda: [0, 0.0])
    ours_n, ours_t = 0, 0.0
    for ev in events:
        for k in ev.kernels:
            name = k.name
            if _ours(name):
                ours_n += 1
                ours_t += k.duration
                if os.environ.get("V2X_CENSUS_SHOW") and os.environ["V2X_CENSUS_SHOW"] in name:
                    print("# %-60s %8.1f us" % (name[:60], k.duration))
                continue
            where = "?"
            for fr in (ev.stack or []):
                if "v2x_sim_amd" in fr or "torch/optim" in fr or "tools/" in fr:
                    where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-90:]
                    break
            r = rows[(ev.name, name[:70], where)]
            r[0] += 1
            r[1] += k.duration
    tot_n = sum(r[0] for r in rows.values())
    tot_t = sum(r[1] for r in rows.values())
    print("# %s, %d maps: one eager step -- %d launches of libv2x_amd.so kernels (%.0f us), %d launches of PyTorch-ROCm ops / copies (%.0f us)" % (
        "V2VNet" if v2v else "FaFNet", frames * 5, ours_n, ours_t, tot_n, tot_t))
    print("%5s %9s  %-34s %-72s %s" % ("n", "us", "aten op", "device kernel", "issued from"))
    for (op, kern, where), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%5d %9.1f  %-34s %-72s %s" % (n, t, op[:34], kern, where))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "faf", int(sys.argv[2]) if len(sys.argv) > 2 else 2)
