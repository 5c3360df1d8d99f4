#!/usr/bin/env python3
"""Where do the small PyTorch launches of a HIP-graph training step come from (row f-3)?  One eager FaFNet / V2VNet step under torch.profiler
with Python stacks: device kernels that are not this library's, grouped by the innermost frame inside this repository.
usage: python tools/train_small_ops.py [FaFNet|V2VNet] [capturable]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from v2x_sim_amd import tuning  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import FaFNet, V2VNet  # noqa: E402
from v2x_sim_amd.train import detection_loss, train_forward  # noqa: E402
from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device  # noqa: E402

family = sys.argv[1] if len(sys.argv) > 1 else "FaFNet"
dev = torch.device("cuda:0")
cfg = Config("train")
data = synthetic_batch_on_device(cfg, 2, 5, seed=1, device=dev)
cls, kw = (FaFNet, dict(kd_flag=0, num_agent=5)) if family == "FaFNet" else (V2VNet, dict(num_agent=5))
model = init_for_training(cls(cfg, **kw), seed=0).to(dev).train()
capturable = len(sys.argv) > 2 and sys.argv[2] == "capturable"      # the optimizer of a captured step (train/graph_step.py)
opt = (torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True, fused=True) if capturable
       else torch.optim.Adam(model.parameters(), lr=1e-4, fused=True))
tuning.set("TRAIN_HIP", 1)


def step():
    res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
    loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by_site = collections.Counter()
by_site_us = collections.Counter()
for ev in prof.events():
    if not ev.kernels or ev.cpu_parent is not None and ev.cpu_parent.kernels:
        continue                                    # innermost op that launched device work
    name = ev.name
    if not name.startswith("aten::"):
        continue
    site = "?"
    for fr in ev.stack or []:
        if "/v2x" in fr or "/tools/" in fr:
            site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            break
    if site == "?":                                  # no Python stack: name the enclosing ops instead
        chain, par = [], ev.cpu_parent
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        site = "inside " + " < ".join(chain) if chain else "top level"
    n = len(ev.kernels)
    us = sum(k.duration for k in ev.kernels)
    by_site[(name, site)] += n
    by_site_us[(name, site)] += us
print("%s: device launches of torch ops in one training step, by op and innermost repository frame" % family)
tot = 0
for (name, site), n in sorted(by_site.items(), key=lambda kv: -by_site_us[kv[0]]):
    print("  %3d launches %8.1f us  %-28s %s" % (n, by_site_us[(name, site)], name, site[:150]))
    tot += n
print("  total %d launches, %.0f us" % (tot, sum(by_site_us.values())))
