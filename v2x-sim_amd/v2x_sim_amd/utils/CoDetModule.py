"""Mirror of upstream coperception/utils/CoDetModule.py::FaFModule (absent from /root/reference; README.md:101
points at tools/det/{train,test}_codet.py which drive it).

`predict_all` keeps the upstream call shape: run the model (HIP engine) on the agent-major batch, then per
agent apply the 'faf' decode + NMS (utils/postprocess.py) unless that agent's BEV is empty.
`step` is the training step: PyTorch-ROCm autograd graph (train/graph.py) over the same
parameter tree, focal + smooth-L1 losses (train/loss.py), optimizer step; the HIP engine re-packs lazily afterwards.
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
        from .. import packing
        packing.watch_optimizer(optimizer)   # fused optimizers update the parameters without bumping their version counters
        self.anchors = postprocess.build_anchor_map(config)
        self.score_thr = 0.7
        self.nms_thr = 0.01
        self.device_postprocess = True    # False: upstream's host-side numpy path (utils/postprocess.apply_nms_det)
        self.fused_detection = True       # with device_postprocess: score threshold in the heads' epilogue, logits never stored
        self.cap = 4096                   # candidate capacity per map of the device path
        self._anchors_dev = None
        # oracle/ASSUMPTIONS.md rows 48 / 49 as switches (configs.Config): which box extent runs along the heading, the loss normaliser
        self.wh_axis = getattr(config, "box_wh_axis", "w_along_heading")
        self.loss_normalizer = getattr(config, "loss_normalizer", "positives")
        if self.wh_axis not in postprocess.WH_AXES:
            raise ValueError("config.box_wh_axis must be one of %s" % (postprocess.WH_AXES,))
        self._graphed = None              # (batch-shape key, GraphedTrainStep) when V2X_TRAIN_GRAPH=1

    def step(self, data, batch_size, num_agent=5):
        """One optimisation step.  data: 'bev_seq' (A*B, 1, X, Y, Z), 'labels' (A*B, X, Y, A', 2), 'reg_targets'
        (A*B, X, Y, A', 1, 6), 'reg_loss_mask' (A*B, X, Y, A', 1), 'trans_matrices' (B, A, A, 4, 4), 'num_agent' (B, A)
        -> (loss, cls_loss, loc_loss) python floats, as upstream."""
        from .. import packing
        from ..train import detection_loss, train_forward
        if self.optimizer is None:
            raise RuntimeError("FaFModule.step needs an optimizer")
        bev = data["bev_seq"]
        if not bev.is_cuda or next(self.model.parameters()).device != bev.device:
            raise RuntimeError("FaFModule.step trains on the MI355X: move the model and the batch to 'cuda'")
        self.model.train()
        if self._graph_ok(data, batch_size):
            # V2X_TRAIN_HIP=1 V2X_TRAIN_GRAPH=1: the whole step as one hipGraph (train/graph_step.py), rebuilt when the batch shape changes
            from ..train.graph_step import GraphedTrainStep
            # The captured steps live ON THE OPTIMIZER (one per batch shape), not on this module: the training loops build a new FaFModule
            # per epoch around the run's one optimizer, and a step re-captured every epoch (or for the partial last batch and back) would
            # pay its warm-up each time.  (The warm-up itself restores whatever optimizer state it finds: graph_step.py.)
            key = self._graph_key(data, batch_size)
            cache = self.optimizer.__dict__.setdefault("_v2x_graphed_steps", {})
            if key not in cache:
                cache[key] = GraphedTrainStep(self.model, self.optimizer, data, batch_size, normalizer=self.loss_normalizer)
            self._graphed = (key, cache[key])
            loss, cls_loss, loc_loss = cache[key](data)
            return loss.item(), cls_loss.item(), loc_loss.item()
        result = train_forward(self.model, bev, data.get("trans_matrices"), data.get("num_agent"), batch_size)
        loss, cls_loss, loc_loss = detection_loss(result, data["labels"], data["reg_targets"], data["reg_loss_mask"], normalizer=self.loss_normalizer)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        packing.stepped(self.optimizer)       # (an optimizer without step hooks: stamp the parameters here)
        return loss.item(), cls_loss.item(), loc_loss.item()

    def _graph_key(self, data, batch_size):
        return (id(self.model),) + tuple(tuple(data[k].shape) for k in ("bev_seq", "labels", "reg_targets", "reg_loss_mask")) + (batch_size, self.loss_normalizer)

    def _graph_ok(self, data, batch_size):
        from .. import tuning
        if tuning.get("TRAIN_HIP") != 1 or tuning.get("TRAIN_GRAPH") != 1:
            return False
        if hasattr(self.model, "outc"):
            return False
        if hasattr(self.model, "convgru"):          # V2VNet: the frame plan is baked into the graph -- only for the agent table it was captured with
            g = self.optimizer.__dict__.get("_v2x_graphed_steps", {}).get(self._graph_key(data, batch_size))
            if g is not None and not torch.equal(data["num_agent"].cpu(), g.num_agent):
                return False
        elif not hasattr(self.model, "stpn"):       # FaFNet has no per-batch host plan; the other baselines stay eager
            return False
        return all(g.get("capturable", True) for g in self.optimizer.param_groups)

    def postprocess(self, result):
        """Device-side score / threshold / decode / NMS (v2x_det_postprocess) for every map of `result` -> list of
        dict(boxes, corners, scores) like postprocess.apply_nms_det; maps with more than `cap` candidates fall back to the
        host function (with random weights tens of thousands of anchors can pass the threshold).
        result = {'det': ...} (the heads already thresholded, DetModelBase.detections): only sort + decode + NMS are left;
        -> None if a map overflowed `cap` there (the caller re-runs the logits path: no logits exist to fall back on)."""
        from .. import ops
        dev = (result["det"][0] if "det" in result else result["cls"]).device
        # "h_along_heading": the kernels lay w along the heading; a box with the extents exchanged is what they compute from anchors with
        # (w, h) exchanged and codes with (dw, dh) exchanged -- w' = a_h exp(dh) = h, h' = a_w exp(dw) = w -- so the second reading costs one
        # channel permutation of the codes and a column swap of the result, no second kernel (the corners below follow self.wh_axis)
        swap = self.wh_axis == "h_along_heading"
        perm = [0, 1, 3, 2, 4, 5]
        if self._anchors_dev is None or self._anchors_dev.device != dev:
            a = np.ascontiguousarray(self.anchors.reshape(-1, 6))
            self._anchors_dev = torch.from_numpy(np.ascontiguousarray(a[:, perm]) if swap else a).to(dev)

        def unswap(b):
            return b[..., [0, 1, 3, 2, 4]] if swap else b
        if "det" in result:
            keys, codes, counts = result["det"]
            boxes, scores, _, count = ops.det_nms_candidates(keys, codes[..., perm].contiguous() if swap else codes, counts, self._anchors_dev, self.nms_thr)
            boxes = unswap(boxes)
            count = count.cpu().numpy()
            if (count < 0).any():
                return None
            kmax = int(max(1, count.max()))
            boxes, scores = boxes[:, :kmax].cpu().numpy(), scores[:, :kmax].cpu().numpy()
            return [{"boxes": boxes[i, :count[i]], "scores": scores[i, :count[i]],
                     "corners": postprocess.box_corners(boxes[i, :count[i]], self.wh_axis) if count[i] else np.zeros((0, 4, 2), np.float32)}
                    for i in range(keys.shape[0])]
        cls, loc = result["cls"], result["loc"]
        boxes, scores, _, count = ops.det_postprocess(cls.contiguous(), (loc.reshape(cls.shape[0], -1, 6)[..., perm] if swap else loc).contiguous(), self._anchors_dev,
                                                      self.score_thr, self.nms_thr, self.cap)
        boxes = unswap(boxes)
        count = count.cpu().numpy()
        kmax = int(max(1, count.max()))
        boxes, scores = boxes[:, :kmax].cpu().numpy(), scores[:, :kmax].cpu().numpy()
        out = []
        for i in range(cls.shape[0]):
            if count[i] < 0:
                out.append(postprocess.apply_nms_det(loc[i].float().cpu().numpy(), cls[i].float().cpu().numpy(), self.anchors,
                                                     self.score_thr, self.nms_thr, wh_axis=self.wh_axis))
                continue
            b = boxes[i, :count[i]]
            out.append({"boxes": b, "corners": postprocess.box_corners(b, self.wh_axis) if count[i] else np.zeros((0, 4, 2), np.float32),
                        "scores": scores[i, :count[i]]})
        return out

    def predict_all(self, data, batch_size, validation=True, num_agent=5, inference="activated"):
        """data: dict with 'bev_seq' (A*B, 1, X, Y, Z), 'trans_matrices' (B, A, A, 4, 4),
        'num_agent' (B, A).  -> (loss, cls_loss, loc_loss, seq_results) with the losses None
        (no labels are consumed at inference) and seq_results[k][b] = detections of agent k in frame b, or None when
        that agent's BEV is empty (upstream skips such agents; the slot is kept so that callers pair detections with
        ground truth BY FRAME, never by list position).  when2com / who2com models run in `inference` mode."""
        bev_seq = data["bev_seq"]

        def run():
            with torch.no_grad():
                if hasattr(self.model, "handshake"):
                    return self.model(bev_seq, data["trans_matrices"], data["num_agent"], training=False, inference=inference,
                                      batch_size=batch_size)
                if hasattr(self.model, "fuse"):
                    return self.model(bev_seq, data["trans_matrices"], data["num_agent"], batch_size=batch_size)
                return self.model(bev_seq)
        dets = None
        if self.device_postprocess and self.fused_detection and hasattr(self.model, "detections"):
            # detections without the logits round trip: the heads' epilogue thresholds the scores (DetModelBase.detections)
            with self.model.detections(self.score_thr, self.cap):
                result = run()
            dets = self.postprocess(result) if "det" in result else None
            if dets is None and "det" in result:
                result = run()              # a map overflowed the candidate capacity: the logits path and its host fallback
        else:
            result = run()
        occupied = (bev_seq.reshape(bev_seq.shape[0], -1) != 0).any(dim=1).cpu().numpy()
        seq_results = [[] for _ in range(num_agent)]
        if dets is None:
            dets = self.postprocess(result) if self.device_postprocess else None
        if dets is None:
            cls = result["cls"].float().cpu().numpy()
            loc = result["loc"].float().cpu().numpy()
        for k in range(num_agent):
            for b in range(batch_size):
                row = k * batch_size + b
                if not occupied[row]:
                    seq_results[k].append(None)
                    continue
                seq_results[k].append(dets[row] if dets is not None else
                                      postprocess.apply_nms_det(loc[row], cls[row], self.anchors, self.score_thr, self.nms_thr))
        self.last_result = result
        return None, None, None, seq_results
