#!/usr/bin/env python3
"""Where the HOST time of an eager training step goes (cProfile over 20 FaFNet steps at 10 maps, device work overlapped): python tools/train_host_profile.py [faf|v2v]"""
import cProfile
import os
import pstats
import sys
import time

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
data = synthetic_batch_on_device(cfg, 2, 5, seed=1, device=dev)
opt = packing.watch_optimizer(torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True, fused=True))
tuning.set("TRAIN_HIP", 1)


def step():
    res = train_forward(model, data["bev_seq"], data["trans_matrices"], data["num_agent"], 2)
    loss = detection_loss(res, data["labels"], data["reg_targets"], data["reg_loss_mask"])[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("20 eager steps: host issue %.2f ms per step, device drained %.2f ms later in total" % ((t1 - t0) * 50, (t2 - t1) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
