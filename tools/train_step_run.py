#!/usr/bin/env python3
"""Runs the FaFNet (default) or V2VNet (`v2v`) training step on the HIP kernels a few times, eagerly (so that a kernel trace sees every launch):
    rocprofv3 --kernel-trace --stats -d out -o t --output-format csv -- python3 tools/train_step_run.py [faf|v2v] [frames (x 5 agents), default 2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from v2x_sim_amd import packing, tuning  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import FaFNet, V2VNet  # noqa: E402
from v2x_sim_amd.train import detection_loss, train_forward  # noqa: E402
from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device  # noqa: E402

dev = torch.device("cuda:0")
cfg = Config("train")
v2v = len(sys.argv) > 1 and sys.argv[1] == "v2v"
model = init_for_training(V2VNet(cfg, num_agent=5) if v2v else FaFNet(cfg, kd_flag=0, num_agent=5), seed=0).to(dev).train()
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
data = synthetic_batch_on_device(cfg, frames, 5, seed=1, device=dev)
opt = packing.watch_optimizer(torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True, fused=True))   # the hook the training loops install: a fused step must invalidate the packed-weight caches
tuning.set("TRAIN_HIP", 1)
STEPS = int(os.environ.get("V2X_TRAIN_RUN_STEPS", "24"))
for it in range(STEPS):    # 24 steps: the first one (lazy packings, optimizer state) weighs 4 % in the per-step averages
    res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], frames)
    loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("loss", float(loss))
