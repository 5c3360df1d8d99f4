"""Inference half of upstream coperception/utils/CoDetModule.py::FaFModule (absent from
/root/reference; README.md:101 points at tools/det/test_codet.py which drives it).

`predict_all` keeps the upstream call shape: run the model on the agent-major batch, then per
agent apply the 'faf' decode + NMS (utils/postprocess.py) unless that agent's BEV is empty.
`step` (training: losses, backward, optimiser) is SURVEY.md row f-3 and not built this round.
"""
import numpy as np
import torch

from . import postprocess


class FaFModule(object):
    def __init__(self, model, teacher, config, optimizer, kd_flag):
        if kd_flag:
            raise NotImplementedError("knowledge distillation is out of scope (DESIGN.md section 8)")
        self.model = model
        self.config = config
        self.optimizer = optimizer
        self.anchors = postprocess.build_anchor_map(config)
        self.score_thr = 0.7
        self.nms_thr = 0.01

    def step(self, data, batch_size, num_agent=5):
        raise NotImplementedError("training step (losses/backward/Adam) is SURVEY.md row f-3: not built yet")

    def predict_all(self, data, batch_size, validation=True, num_agent=5):
        """data: dict with 'bev_seq' (A*B, 1, X, Y, Z), 'trans_matrices' (B, A, A, 4, 4),
        'num_agent' (B, A).  -> (loss, cls_loss, loc_loss, seq_results) with the losses None
        (no labels are consumed at inference) and seq_results[k] = list of detections of agent k."""
        bev_seq = data["bev_seq"]
        with torch.no_grad():
            if hasattr(self.model, "fuse") or hasattr(self.model, "handshake"):
                result = self.model(bev_seq, data["trans_matrices"], data["num_agent"], batch_size=batch_size)
            else:
                result = self.model(bev_seq)
        cls = result["cls"].float().cpu().numpy()
        loc = result["loc"].float().cpu().numpy()
        occupied = (bev_seq.reshape(bev_seq.shape[0], -1) != 0).any(dim=1).cpu().numpy()
        seq_results = [[] for _ in range(num_agent)]
        for k in range(num_agent):
            for b in range(batch_size):
                row = k * batch_size + b
                if not occupied[row]:
                    continue
                seq_results[k].append(postprocess.apply_nms_det(loc[row], cls[row], self.anchors, self.score_thr,
                                                                self.nms_thr))
        return None, None, None, seq_results
