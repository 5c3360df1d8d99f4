"""The optimizer step of the training graph on the library's kernel.

Upstream's train scripts (/root/reference/README.md:101) build `torch.optim.Adam(model.parameters(), lr=args.lr)` and hand it to FaFModule / SegModule; that
stays as it is.  `use_hip_adam(optimizer)` -- called by packing.watch_optimizer, i.e. by FaFModule, SegModule, make_optimizer and GraphedTrainStep -- switches a
plain torch.optim.Adam INSTANCE to the subclass below: same constructor arguments, same state (`step`, `exp_avg`, `exp_avg_sq`: state_dict() /
load_state_dict() and checkpoints are interchangeable), same hooks; only step() differs: one launch of v2x_adam_step_f32 per <= 72 parameter tensors
(FaFNet: 2 launches; torch's fused step: 3 launches at 1.8 TB/s) + one counter increment.  Anything the kernel does not cover -- amsgrad, maximize, decoupled
weight decay, a GradScaler's grad_scale / found_inf, differentiable, complex / non-fp32 / CPU / sparse parameters, tensor betas -- takes torch's own
step (super().step()), silently: the result is the same update either way (tests/test_gpu_train_kernels.py::test_hip_adam_*).  TRAIN_ADAM_HIP 0 = never switch."""
import ctypes as C

import torch

from .. import _lib, tuning
from .._launch import _stream


class HipAdam(torch.optim.Adam):
    def _hip_ok(self, group):
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or group.get("decoupled_weight_decay"):
            return False
        if getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return False
        b1, b2 = group["betas"]
        if torch.is_tensor(b1) or torch.is_tensor(b2) or torch.is_tensor(group["eps"]) or torch.is_tensor(group["weight_decay"]):
            return False
        lr = group["lr"]
        if torch.is_tensor(lr) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()
                    and not g.is_sparse and p.numel() < (1 << 31)):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if tuning.get("TRAIN_ADAM_HIP") == 0 or not all(self._hip_ok(g) for g in self.param_groups):
            return super().step(closure)
        self._cuda_graph_capture_health_check()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        work = []
        for group in self.param_groups:          # gather first: nothing is updated before every group is known to take the kernel
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            try:
                self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)   # torch's own lazy state initialisation (a private method:
            except TypeError:                                                                  #  another torch version may spell it differently)
                super().step(None)
                return loss
            if not params:
                continue
            if any(st.dtype != torch.float32 for st in steps):
                super().step(None)
                return loss
            if not steps[0].is_cuda and len({float(st) for st in steps}) > 1:      # host counters at different steps (parameters that sat out earlier
                super().step(None)                                                  #  steps): torch's per-tensor bias corrections
                return loss
            work.append((group, params, grads, exp_avgs, exp_avg_sqs, steps))
        for group, params, grads, exp_avgs, exp_avg_sqs, steps in work:
            dev_steps = steps[0].is_cuda
            torch._foreach_add_(steps, 1)
            step_host = 0.0 if dev_steps else float(steps[0])
            lr = group["lr"]
            lr_dev, lr_f = (C.c_void_p(lr.data_ptr()), 0.0) if torch.is_tensor(lr) else (None, float(lr))
            b1, b2 = group["betas"]
            stream = _stream()
            for i0 in range(0, len(params), _lib.ADAM_MAX_TENSORS):
                tb = _lib.AdamTensors()
                n = min(_lib.ADAM_MAX_TENSORS, len(params) - i0)
                for k in range(n):
                    i = i0 + k
                    m, v = exp_avgs[i], exp_avg_sqs[i]
                    if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32):
                        raise RuntimeError("HipAdam: optimizer state of an unexpected layout (load_state_dict from another dtype?)")
                    tb.param[k], tb.grad[k], tb.exp_avg[k], tb.exp_avg_sq[k] = params[i].data_ptr(), grads[i].data_ptr(), m.data_ptr(), v.data_ptr()
                    tb.step[k] = steps[i].data_ptr() if dev_steps else None
                    tb.numel[k] = params[i].numel()
                _lib.check(lib.v2x_adam_step_f32(C.byref(tb), n, lr_dev, lr_f, float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                 step_host, stream), "v2x_adam_step_f32")
        return loss


def use_hip_adam(opt):
    """A plain torch.optim.Adam instance -> the same optimizer stepping on v2x_adam_step_f32 (see the module docstring); anything else is returned as it is."""
    if opt is None or type(opt) is not torch.optim.Adam or tuning.get("TRAIN_ADAM_HIP") == 0:
        return opt
    if not all(p.is_cuda and p.dtype == torch.float32 for g in opt.param_groups for p in g["params"]):
        return opt
    opt.__class__ = HipAdam
    opt._patch_step_function()           # torch wraps a class's step() with the hook / profiler shell once per class
    return opt
