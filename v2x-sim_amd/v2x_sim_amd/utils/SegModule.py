"""Mirror of upstream coperception/utils/SegModule.py (absent from /root/reference; README.md:101 points at tools/seg/
{train,test}_seg.py which drive it): `step` = one optimisation step with pixel-wise cross entropy (PyTorch-ROCm autograd
graph over the engine's parameter tree, train/graph.py), `predict` = HIP inference + argmax / confusion matrix on the device
(v2x_seg_argmax_confusion)."""
import torch
import torch.nn.functional as F


class SegModule(object):
    def __init__(self, model, teacher, config, optimizer, kd_flag=0):
        if kd_flag:
            raise NotImplementedError("knowledge distillation is out of scope (DESIGN.md section 8)")
        self.model, self.config, self.optimizer = model, config, optimizer
        from .. import packing
        packing.watch_optimizer(optimizer)   # fused optimizers update the parameters without bumping their version counters

    def step(self, data, num_agent=5, batch_size=1):
        """data: 'bev_seq' (A*B, 1, X, Y, Z), 'labels' (A*B, X, Y) uint8/int64, 'trans_matrices', 'num_agent' -> loss (float)."""
        from ..train import train_forward
        bev = data["bev_seq"]
        if not bev.is_cuda:
            raise RuntimeError("SegModule.step trains on the MI355X: move the model and the batch to 'cuda'")
        self.model.train()
        logits = train_forward(self.model, bev, data.get("trans_matrices"), data.get("num_agent"), batch_size)
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), data["labels"].reshape(-1).long())
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        from .. import packing
        packing.stepped(self.optimizer)       # (an optimizer without step hooks: stamp the parameters here)
        return loss.item()

    def predict(self, data, batch_size=1, label=None):
        """-> (pred (A*B, X, Y) uint8, confusion matrix int64 [n_cls, n_cls] or None) on the HIP path."""
        from .. import ops
        self.model.eval()
        with torch.no_grad():
            x0 = self.model._input_nhwc(data["bev_seq"])
            if hasattr(self.model, "fuse"):
                logits = self.model.forward_nhwc(x0, data["trans_matrices"], data["num_agent"], batch_size=batch_size)
            else:
                logits = self.model.forward_nhwc(x0)
        return ops.seg_argmax_confusion(logits, label)


def iou_from_confusion(conf):
    """rows = label, cols = prediction -> per-class IoU (nan where the class is absent)."""
    conf = conf.double()
    tp = conf.diag()
    denom = conf.sum(0) + conf.sum(1) - tp
    return torch.where(denom > 0, tp / denom, torch.full_like(tp, float("nan")))
