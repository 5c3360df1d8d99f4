#!/usr/bin/env python3
"""Segmentation training driver -- the flag surface of upstream coperception/tools/seg/train_seg.py
(/root/reference/README.md:101 points at it; the script itself is not in the reference tree).

    python tools/seg/train_seg.py --data synthetic --com v2v --steps 300 --batch 2 --logpath out/
    python tools/seg/train_seg.py --data /path/V2X-Sim-seg/train --com v2v --nepoch 5 --batch 2 --logpath out/

Synthetic scenes (utils/synthetic_scene.py: vehicle footprints as class 1), PyTorch-ROCm autograd over the HIP engine's
parameter tree, checkpoint with upstream's 'model_state_dict' key for tools/seg/test_seg.py --resume."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def seg_batch(config, frames, agents, seed, device, grid):
    from v2x_sim_amd import ops
    from v2x_sim_amd.utils import synthetic_scene
    b = synthetic_scene.make_batch(frames, agents, seed=seed)
    bits = ops.voxelize_bits(torch.from_numpy(b["points"]).to(device), torch.from_numpy(b["n_pts"]).to(device), grid)
    labels = np.stack([synthetic_scene.seg_labels(b["gt_boxes"][a][f], config) for a in range(agents) for f in range(frames)])
    return {"bev_seq": ops.bits_to_dense(bits, grid.dims[2])[:, None], "trans_matrices": torch.from_numpy(b["trans"]).to(device),
            "num_agent": torch.from_numpy(b["num_agent"]), "labels": torch.from_numpy(labels).to(device)}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", default="synthetic", type=str)
    ap.add_argument("--com", default="v2v", choices=["lowerbound", "upperbound", "v2v"])
    ap.add_argument("--batch", default=2, type=int)
    ap.add_argument("--steps", default=300, type=int)
    ap.add_argument("--nepoch", default=1, type=int)
    ap.add_argument("--lr", default=1e-3, type=float)
    ap.add_argument("--num_agent", default=5, type=int)
    ap.add_argument("--logpath", default="", type=str)
    ap.add_argument("--seed", default=0, type=int)
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--rsu", default=1, type=int, help="parsed dataset: 1 = agent0 (the RSU) takes part, 0 = vehicles only")
    ap.add_argument("--resume", default="", type=str)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.seg import FaFNetSeg, V2VNetSeg
    from v2x_sim_amd.train.loop import init_for_training
    from v2x_sim_amd.utils.SegModule import SegModule
    if not torch.cuda.is_available():
        raise SystemExit("train_seg.py needs the MI355X")
    device = torch.device("cuda:0")
    config = Config("train", binary=True, only_det=True)
    A = args.num_agent
    model = V2VNetSeg(config, num_agent=A) if args.com == "v2v" else FaFNetSeg(config, num_agent=A)
    ckpt = None
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu")
        model.load_state_dict(ckpt.get("model_state_dict", ckpt), strict=True)
    else:
        init_for_training(model, seed=args.seed)
    model.to(device)
    loader = None
    if args.data != "synthetic":
        # parsed dataset in the README.md:66-79 layout: <data>/agent{k}/{scene}_{frame}/0.npy with 'bev_seg' (agent0 = RSU)
        from torch.utils.data import DataLoader
        from v2x_sim_amd.datasets import V2XSimSeg, seg_batch_on_device
        first = 0 if args.rsu else 1
        roots = [os.path.join(args.data, "agent%d" % k) for k in range(first, first + A)]
        dataset = V2XSimSeg(dataset_roots=roots, config=config, split="train", densify="none")
        loader = DataLoader(dataset, batch_size=args.batch, shuffle=True, collate_fn=lambda x: x,
                            generator=torch.Generator().manual_seed(args.seed))
        print("training on %d frames x %d agents from %s" % (len(dataset), A, args.data))
    # one optimizer for the whole run, kept in the checkpoint (same rule as tools/det/train_codet.py)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    start = 1
    if ckpt is not None and "optimizer_state_dict" in ckpt:
        opt.load_state_dict(ckpt["optimizer_state_dict"])
    if ckpt is not None and "epoch" in ckpt:
        start = int(ckpt["epoch"]) + 1
    module = SegModule(model, None, config, opt, 0)
    grid = ops.VoxelGrid(config.voxel_size, config.area_extents)
    for epoch in range(start, args.nepoch + 1):
        losses = []
        if loader is not None:
            for samples in loader:
                losses.append(module.step(seg_batch_on_device(samples, grid, device), len(samples[0]), len(samples)))
        else:
            for it in range(args.steps):
                losses.append(module.step(seg_batch(config, args.batch, A, (args.seed + epoch) * 1000003 + it, device, grid), A, args.batch))
                if args.log and it % 20 == 0:
                    print("step %4d  loss %.4f" % (it, losses[-1]), flush=True)
        print("epoch %d: mean loss of the last 20 steps %.4f" % (epoch, float(np.mean(losses[-20:]))))
        if args.logpath:
            os.makedirs(args.logpath, exist_ok=True)
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict()},
                       os.path.join(args.logpath, "epoch_%d.pth" % epoch))
    model.eval()
    return model


if __name__ == "__main__":
    main()
