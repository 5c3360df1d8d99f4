#!/usr/bin/env python3
"""Detection training driver -- the flag surface of upstream coperception/tools/det/train_codet.py
(/root/reference/README.md:101 points at it; the script itself is not in the reference tree).

    python tools/det/train_codet.py --data synthetic --com v2v --nepoch 1 --steps 400 --batch 2 --logpath out/
    python tools/det/train_codet.py --data /path/V2X-Sim-det/train --com v2v --nepoch 10 --batch 2 --logpath out/

Training = PyTorch-ROCm autograd graph over the parameter tree of the HIP engine (v2x_sim_amd/train/); the sweeps are
voxelised by the HIP voxeliser.  `--data synthetic` draws fresh synthetic scenes (the V2X-Sim download is not
available offline, README.md:42-48); a directory trains on a parsed dataset in the README.md:66-79 layout (sparse sweeps
densified and anchor targets scattered on the GPU, train/loop.py::dataset_batch_on_device).  Writes epoch_{n}.pth with 'model_state_dict' (upstream's checkpoint key), which
tools/det/test_codet.py --resume loads into the HIP inference path."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", default="synthetic", type=str)
    ap.add_argument("--com", default="v2v", choices=["lowerbound", "upperbound", "v2v", "when2com", "sum", "mean", "max", "cat", "disco"])
    ap.add_argument("--batch", default=2, type=int, help="frames per step")
    ap.add_argument("--nepoch", default=1, type=int)
    ap.add_argument("--steps", default=400, type=int, help="steps per epoch (synthetic data has no natural epoch)")
    ap.add_argument("--lr", default=1e-3, type=float)
    ap.add_argument("--num_agent", default=5, type=int)
    ap.add_argument("--layer", default=3, type=int)
    ap.add_argument("--gnn_iter_times", default=1, type=int)
    ap.add_argument("--resume", default="", type=str)
    ap.add_argument("--logpath", default="", type=str)
    ap.add_argument("--seed", default=0, type=int)
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--nworker", default=0, type=int, help="DataLoader workers (parsed dataset)")
    ap.add_argument("--rsu", default=1, type=int, help="parsed dataset: 1 = agent0 (the RSU) takes part, 0 = vehicles only")
    ap.add_argument("--engine", default="", choices=["", "torch", "hip", "hip-graph"],
                    help="training graph: torch = PyTorch-ROCm ops (fp32, MIOpen); hip = bf16 NHWC graph on the hand-written kernels "
                         "(V2X_TRAIN_HIP=1); hip-graph = the same with every step replayed as one hipGraph (V2X_TRAIN_GRAPH=1; FaFNet, and "
                         "V2VNet while the agent table does not change).  Default: whatever the environment variables say")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.engine:
        from v2x_sim_amd import tuning
        tuning.set("TRAIN_HIP", int("0" if args.engine == "torch" else "1"))
        tuning.set("TRAIN_GRAPH", int("1" if args.engine == "hip-graph" else "0"))
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import CatFusion, DiscoNet, FaFNet, MaxFusion, MeanFusion, SumFusion, V2VNet, When2com
    from v2x_sim_amd.train.loop import init_for_training, make_optimizer, train_dataset, train_synthetic
    if not torch.cuda.is_available():
        raise SystemExit("train_codet.py needs the MI355X")
    config = Config("train", binary=True, only_det=True)
    A = args.num_agent
    if args.com == "v2v":
        model = V2VNet(config, gnn_iter_times=args.gnn_iter_times, layer=args.layer, num_agent=A)
    elif args.com == "when2com":
        model = When2com(config, layer=args.layer, num_agent=A)
    elif args.com in ("sum", "mean", "max", "cat", "disco"):
        cls = {"sum": SumFusion, "mean": MeanFusion, "max": MaxFusion, "cat": CatFusion, "disco": DiscoNet}[args.com]
        model = cls(config, layer=args.layer, kd_flag=0, num_agent=A)      # DiscoNet without the distillation teacher
    else:
        model = FaFNet(config, layer=args.layer, kd_flag=0, num_agent=A)
    ckpt = None
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu")
        model.load_state_dict(ckpt.get("model_state_dict", ckpt), strict=True)  # a key mismatch must not train from init silently
    else:
        init_for_training(model, seed=args.seed)
    dataset = None
    if args.data != "synthetic":
        # parsed dataset in the README.md:66-79 layout: <data>/agent{k}/{scene}_{frame}/0.npy (agent0 = RSU)
        from v2x_sim_amd.datasets import V2XSimDet
        first = 0 if args.rsu else 1
        roots = [os.path.join(args.data, "agent%d" % k) for k in range(first, first + A)]
        dataset = V2XSimDet(dataset_roots=roots, config=config, split="train", densify="none")
        print("training on %d frames x %d agents from %s" % (len(dataset), A, args.data))
    # ONE optimizer and ONE scheduler for the whole run (upstream keeps both across epochs): Adam's moments and the
    # decay survive epoch boundaries, and both travel in the checkpoint.  --nepoch is the run's LAST epoch number, so a
    # resumed run continues the numbering (and the schedule) where the checkpoint stopped.
    model.to("cuda:0")
    per_epoch = args.steps if dataset is None else (len(dataset) + args.batch - 1) // args.batch
    opt, sched = make_optimizer(model, args.lr, args.nepoch * per_epoch)
    start = 1
    if ckpt is not None and "optimizer_state_dict" in ckpt:
        opt.load_state_dict(ckpt["optimizer_state_dict"])
        sched.load_state_dict(ckpt["scheduler_state_dict"])
    if ckpt is not None and "epoch" in ckpt:
        start = int(ckpt["epoch"]) + 1
    if start > args.nepoch:
        print("checkpoint is at epoch %d: nothing left to train up to --nepoch %d" % (start - 1, args.nepoch))
    for epoch in range(start, args.nepoch + 1):
        if dataset is not None:
            hist = train_dataset(model, config, dataset, 1, args.batch, args.lr, seed=args.seed + epoch,
                                 log=20 if args.log else None, num_workers=args.nworker, opt=opt, sched=sched)
        else:
            hist = train_synthetic(model, config, args.steps, args.batch, args.lr, seed=args.seed + epoch,
                                   log=20 if args.log else None, opt=opt, sched=sched)
        tail = hist[-20:]
        print("epoch %d (lr %.2e): mean loss of the last %d steps %.4f" % (epoch, opt.param_groups[0]["lr"], len(tail),
                                                                           sum(h[0] for h in tail) / len(tail)))
        if args.logpath:
            os.makedirs(args.logpath, exist_ok=True)
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
                        "scheduler_state_dict": sched.state_dict()}, os.path.join(args.logpath, "epoch_%d.pth" % epoch))
    return model


if __name__ == "__main__":
    main()
