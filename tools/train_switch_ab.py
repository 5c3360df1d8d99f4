#!/usr/bin/env python3
"""Paired timing of the training-step switches (row f-3): one captured step per switch setting, replays interleaved round-robin in ONE process (box and clock
drift cancel).  usage: python tools/train_switch_ab.py [FaFNet|V2VNet] [frames] [SETTING ...]   with SETTING = NAME=v[,NAME=v...] (tuning switches; "default" = none
changed); without settings: TRAIN_PACK_BATCH=0 against TRAIN_PACK_BATCH=1 (round 4's comparison)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
from v2x_sim_amd import tuning  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import FaFNet, V2VNet  # noqa: E402
from v2x_sim_amd.train import hip_graph  # noqa: E402
from v2x_sim_amd.train.graph_step import GraphedTrainStep  # noqa: E402
from v2x_sim_amd.train.loop import init_for_training, synthetic_batch_on_device  # noqa: E402


def main(family="FaFNet", frames=2, settings=("TRAIN_PACK_BATCH=0", "TRAIN_PACK_BATCH=1"), agents=5, rounds=5, reps=20):
    dev = torch.device("cuda:0")
    cfg = Config("train")
    data = synthetic_batch_on_device(cfg, frames, agents, seed=1, device=dev)
    cls, kw = (FaFNet, dict(kd_flag=0, num_agent=agents)) if family == "FaFNet" else (V2VNet, dict(num_agent=agents))
    base = init_for_training(cls(cfg, **kw), seed=0).to(dev).train()
    tuning.set("TRAIN_HIP", 1)
    steps = {}
    for setting in settings:
        pairs = [] if setting == "default" else [kv.split("=") for kv in setting.split(",")]
        saved = {k: tuning.get(k) for k, _ in pairs}
        for k, v in pairs:
            tuning.set(k, int(v))
        hip_graph._CACHE.clear()
        hip_graph._PLANS.clear()
        m = copy.deepcopy(base)
        opt = torch.optim.Adam(m.parameters(), lr=torch.tensor(1e-4, device=dev), capturable=True, fused=True)
        steps[setting] = GraphedTrainStep(m, opt, data, frames)      # the switches are read while the step is captured: the graph keeps its kernels
        for k, v in saved.items():
            tuning.set(k, v)
    times = {k: [] for k in steps}
    for _ in range(rounds):
        for k, g in steps.items():
            for _ in range(3):
                g(data)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g(data)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / reps)
    print("%s, %d maps per step, captured step replayed (ms, median of %d rounds x %d replays; min):" % (family, frames * agents, rounds, reps))
    for setting, t in times.items():
        t = sorted(t)
        print("  %-48s : %.3f  (min %.3f)" % (setting, t[len(t) // 2], t[0]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "FaFNet", int(sys.argv[2]) if len(sys.argv) > 2 else 2,
         tuple(sys.argv[3:]) if len(sys.argv) > 3 else ("TRAIN_PACK_BATCH=0", "TRAIN_PACK_BATCH=1"))
