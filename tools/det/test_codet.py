#!/usr/bin/env python3
"""Detection evaluation driver -- the flag surface of upstream coperception/tools/det/test_codet.py
(/root/reference/README.md:101 points at it; the script itself is not in the reference tree) on top of the
MI355X hot path:  parsed dataset (README.md:66-79 layout) -> model -> FaFModule.predict_all -> eval_map.

    python tools/det/test_codet.py --data /path/V2X-Sim-det/test --com v2v --resume ckpt.pth --num_agent 5

Prints per-agent and mean mAP@0.5 / mAP@0.7 like upstream.  --com: lowerbound | upperbound (FaFNet), v2v
(V2VNet), when2com (inference 'activated'), who2com ('argmax_test').  The agent directories agent0..agentN-1
under --data are used (agent0 = RSU, README.md:70; pass --rsu 0 to skip it)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--data", required=True, type=str, help="the {split} directory holding agent{k}/")
    ap.add_argument("--com", default="v2v", choices=["lowerbound", "upperbound", "v2v", "when2com", "who2com", "sum", "mean", "max", "cat", "disco"])
    ap.add_argument("--resume", default="", type=str, help="checkpoint with 'model_state_dict' (or a bare state_dict)")
    ap.add_argument("--num_agent", default=5, type=int)
    ap.add_argument("--rsu", default=1, type=int, help="1: agent0 (the RSU) takes part, 0: vehicles only")
    ap.add_argument("--layer", default=3, type=int)
    ap.add_argument("--gnn_iter_times", default=1, type=int)
    ap.add_argument("--inference", default=None, type=str, help="softmax | activated | argmax_test")
    ap.add_argument("--warp_flag", default=1, type=int)
    ap.add_argument("--batch", default=1, type=int)
    ap.add_argument("--score_thr", default=0.7, type=float)
    ap.add_argument("--seed", default=0, type=int, help="synthetic-weights seed when --resume is not given")
    ap.add_argument("--log", action="store_true")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    from v2x_sim_amd.configs import Config, ConfigGlobal
    from v2x_sim_amd.datasets import V2XSimDet, collate_dense
    from v2x_sim_amd.models.det import CatFusion, DiscoNet, FaFNet, MaxFusion, MeanFusion, SumFusion, V2VNet, When2com
    from v2x_sim_amd.utils import postprocess as P
    from v2x_sim_amd.utils.CoDetModule import FaFModule
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights

    if not torch.cuda.is_available():
        raise SystemExit("test_codet.py needs the MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda:0")
    config, config_global = Config("test", binary=True, only_det=True), ConfigGlobal("test", binary=True, only_det=True)
    first = 0 if args.rsu else 1
    roots = [os.path.join(args.data, "agent%d" % k) for k in range(first, first + args.num_agent)]
    dataset = V2XSimDet(dataset_roots=roots, config=config, config_global=config_global, split="test", val=True)
    A = args.num_agent
    if args.com in ("lowerbound", "upperbound"):
        model = FaFNet(config, layer=args.layer, kd_flag=0, num_agent=A)
    elif args.com == "v2v":
        model = V2VNet(config, gnn_iter_times=args.gnn_iter_times, layer=args.layer, layer_channel=256, num_agent=A)
    elif args.com in ("sum", "mean", "max", "cat", "disco"):
        model = {"sum": SumFusion, "mean": MeanFusion, "max": MaxFusion, "cat": CatFusion, "disco": DiscoNet}[args.com](
            config, layer=args.layer, kd_flag=0, num_agent=A)
    else:
        model = When2com(config, layer=args.layer, warp_flag=args.warp_flag, num_agent=A)
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu")
        # strict: a checkpoint whose keys do not match must not be scored as if it had been loaded
        model.load_state_dict(ckpt.get("model_state_dict", ckpt), strict=True)
        model.eval()
    else:
        print("no --resume given: evaluating seeded synthetic weights (there is no released checkpoint in this tree)")
        init_synthetic_weights(model, seed=args.seed)
    model = model.to(device)
    module = FaFModule(model, None, config, None, 0)
    module.score_thr = args.score_thr
    inference = args.inference or ("argmax_test" if args.com == "who2com" else "activated")

    det_results = [[] for _ in range(A)]
    annotations = [[] for _ in range(A)]
    for start in range(0, len(dataset), args.batch):
        samples = [dataset[i] for i in range(start, min(start + args.batch, len(dataset)))]
        B = len(samples)
        bevs, trans, nat = collate_dense(samples)
        data = {"bev_seq": bevs.to(device), "trans_matrices": trans.to(device), "num_agent": nat}
        # one rule for every --com: predict_all returns a slot per (agent, frame), None where the agent's sweep is empty
        _, _, _, seq = module.predict_all(data, B, validation=False, num_agent=A, inference=inference)
        for k in range(A):
            for b in range(B):
                if seq[k][b] is not None:   # paired BY FRAME: frame b's detections meet frame b's ground truth
                    det_results[k].append(seq[k][b])
                    annotations[k].append(P.box_corners(samples[b][k][12].astype(np.float64)))
        if args.log:
            print("frames %d-%d done" % (start, start + B - 1))

    means = {0.5: [], 0.7: []}
    for k in range(A):
        line = "agent%d:" % (k + first)
        for iou in (0.5, 0.7):
            ap, info = P.eval_map(det_results[k], annotations[k], iou)
            means[iou].append(ap)
            line += "  mAP@%.1f %.2f" % (iou, 100 * ap)
        print(line + "  (%d frames, %d gt)" % (len(det_results[k]), info["num_gt"]))
    print("average local mAP@0.5 %.2f  mAP@0.7 %.2f" % (100 * float(np.mean(means[0.5])), 100 * float(np.mean(means[0.7]))))
    return {iou: float(np.mean(v)) for iou, v in means.items()}


if __name__ == "__main__":
    main()
