#!/usr/bin/env python3
"""Trains the same detector for --steps steps with each training engine (torch = fp32 PyTorch-ROCm graph, hip = bf16 NHWC graph on the
hand-written kernels) and reports the loss tail and how many anchors of fresh scenes pass the score threshold on the HIP inference path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from v2x_sim_amd import ops, tuning
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, train_synthetic
    from v2x_sim_amd.utils import synthetic_scene
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    dev = torch.device("cuda:0")
    for engine in sys.argv[2:] or ["torch", "hip"]:
        tuning.set("TRAIN_HIP", 0 if engine.startswith("torch") else 1)
        config = Config("train", binary=True, only_det=True)
        model = V2VNet(config, gnn_iter_times=1, layer=3, num_agent=5)
        init_for_training(model, seed=0)
        model.to(dev)
        opt, sched = make_optimizer(model, 1e-3, steps)
        if engine.endswith("nofused"):
            opt, sched = torch.optim.Adam(model.parameters(), lr=1e-3), None
        hist = train_synthetic(model, config, steps, 2, 1e-3, seed=1, log=None, opt=opt, sched=sched)
        h = np.array([x[0] for x in hist])
        print("%s: loss first 10 %.3f | steps 40-60 %.3f | last 20 %.3f | max %.3f | nan %d" % (engine, h[:10].mean(), h[40:60].mean(), h[-20:].mean(), np.nanmax(h), int(np.isnan(h).sum())))
        model.eval()
        grid = ops.VoxelGrid()
        for f in range(2):
            sc = synthetic_scene.make_scene(5, seed=9000 + f)
            bits = ops.voxelize_bits(torch.from_numpy(sc["points"]).to(dev), torch.from_numpy(sc["n_pts"]).to(dev), grid)
            bev = ops.bits_to_dense(bits, grid.dims[2]).unsqueeze(1).float()
            with torch.no_grad():
                res = model(bev, torch.from_numpy(sc["trans"][None].astype(np.float32)).to(dev) if sc["trans"].ndim == 4 else torch.from_numpy(sc["trans"].astype(np.float32)).to(dev),
                            torch.full((1, 5), 5, dtype=torch.int32, device=dev), batch_size=1)
            cls = res["cls"].float()
            p = torch.softmax(cls.reshape(5, -1, 2), -1)[..., 1]
            print("   scene %d: anchors >= 0.7 per agent %s   finite %s" % (f, (p >= 0.7).sum(1).tolist(), bool(torch.isfinite(cls).all())))


if __name__ == "__main__":
    main()
