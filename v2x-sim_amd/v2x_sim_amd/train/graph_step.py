"""One training step -- forward, loss, backward, optimizer -- captured as ONE hipGraph and replayed.

On the bf16 NHWC graph of train/hip_graph.py a 10-map FaFNet step is ~110 kernel launches of this library plus ~600 small PyTorch-ROCm
ops (concats, casts, the weight re-packing, the loss, Adam): 13.5 ms of host time around 10.7 ms of GPU time.  Nothing in the step depends
on values the host must see (the loss uses a masked sum, the optimizer is `capturable`), so the whole step is recorded once and replayed
with one launch; the host only copies the next batch into the static input tensors and reads the loss when it wants to.

Scope: models whose graph has no host-side plan that changes per batch -- FaFNet (lowerbound / upperbound) always; V2VNet when every batch
has the same `num_agent` table (the frame plan is baked at capture time; the constructor checks, __call__ re-checks).
"""
import torch

from .loss import detection_loss

_KEYS = ("bev_seq", "labels", "reg_targets", "reg_loss_mask", "trans_matrices")


class GraphedTrainStep:
    def __init__(self, model, optimizer, data, batch_size, warmup=3, normalizer="positives"):
        """model on the MI355X in train mode; optimizer: torch.optim.Adam(..., capturable=True) (lr may be a device tensor, see set_lr) or one without host-side state (plain SGD);
        data: one batch in FaFModule.step's format -- its shapes are the shapes of every later batch."""
        from .graph import train_forward
        if not data["bev_seq"].is_cuda:
            raise RuntimeError("GraphedTrainStep runs on the MI355X")
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"]:
                raise ValueError("GraphedTrainStep needs an optimizer created with capturable=True (its step counter lives on the host otherwise)")
        from .optim import use_hip_adam
        use_hip_adam(optimizer)          # (a plain torch.optim.Adam steps on the library's kernel: train/optim.py)
        self.model, self.optimizer, self.batch_size = model, optimizer, batch_size
        self.normalizer = normalizer     # configs.Config.loss_normalizer (oracle/ASSUMPTIONS.md row 49)
        self.static = {k: data[k].clone() for k in _KEYS if data.get(k) is not None}
        self.num_agent = None if data.get("num_agent") is None else data["num_agent"].clone().cpu()
        self._forward = train_forward
        model.train()
        # side-effect-free warm-up: run real steps (allocator, lazy optimizer state, kernel attributes), then put parameters, buffers and
        # optimizer state back -- IN PLACE, the captured graph must see the very tensors the warm-up created
        snap = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # the optimizer may already carry a run's moments and step counters (a later epoch, a partial last batch, --resume): keep them
        opt_snap = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in optimizer.state.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(snap[k])
            for p, st in optimizer.state.items():
                before = opt_snap.get(p)
                for k, v in st.items():
                    if not torch.is_tensor(v):
                        continue
                    if before is not None and torch.is_tensor(before.get(k)):
                        v.copy_(before[k])           # what the run had accumulated so far
                    else:
                        v.zero_()                    # state the warm-up itself created (a fresh optimizer)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.losses = self._step_body()
        from . import hip_graph
        # the captured step rebuilds its packed weights IN PLACE in the buffers the warm-up steps allocated (hip_graph._repack_stale): they must
        # live as long as the graph does, whatever happens to the module-level caches
        self._repack_plans = list(hip_graph._PLANS.values())
        hip_graph._PLANS.clear()
        hip_graph._CACHE.clear()     # the packed weights recorded in the graph are rewritten by every replay; eager callers re-pack

    def _step_body(self):
        s = self.static
        res = self._forward(self.model, s["bev_seq"], s.get("trans_matrices"), self.num_agent, self.batch_size)
        loss, cls_loss, loc_loss = detection_loss(res, s["labels"], s["reg_targets"], s["reg_loss_mask"], normalizer=self.normalizer)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss.detach(), cls_loss.detach(), loc_loss.detach()

    def set_lr(self, lr):
        """Learning rate of the captured optimizer step: only a TENSOR lr (torch.optim.Adam(lr=torch.tensor(..., device=...))) can change
        after capture."""
        for g in self.optimizer.param_groups:
            if not torch.is_tensor(g["lr"]):
                raise ValueError("the captured step has its float lr baked in; create the optimizer with a device-tensor lr")
            g["lr"].fill_(lr)

    def __call__(self, data):
        """Copies the batch into the static inputs and replays the step.  -> (loss, cls_loss, loc_loss) device tensors (valid until the next
        call; .item() them when the numbers are needed)."""
        if self.num_agent is not None and data.get("num_agent") is not None and not torch.equal(data["num_agent"].cpu(), self.num_agent):
            raise ValueError("GraphedTrainStep: this batch's num_agent table differs from the captured one (the frame plan is baked in)")
        for k, v in self.static.items():
            if data[k].shape != v.shape:
                raise ValueError("GraphedTrainStep: %s has shape %s, captured %s" % (k, tuple(data[k].shape), tuple(v.shape)))
            v.copy_(data[k])
        self.graph.replay()
        # a replay changes the parameters without bumping their version counters, which is what the inference engine's packed-weights
        # cache looks at (DetModelBase.packed): drop that cache so that the next model(...) call re-packs
        if hasattr(self.model, "repack"):
            self.model.repack()
        # ... and the same for the training graph's own packed-weight cache (hip_graph._layer): an EAGER step after replays (another batch shape,
        # the partial last batch) must not find the packings of the last eager step
        from .. import packing
        for g in self.optimizer.param_groups:
            packing.note_params_changed(g["params"])
        return self.losses
