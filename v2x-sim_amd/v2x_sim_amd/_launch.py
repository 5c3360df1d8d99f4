"""Launch plumbing shared by the wrapper modules (ops.py: the forward hot path, ops_train.py: row f-3, ops_post.py: row f-1): the current HIP
stream, the device-tensor argument check (there is deliberately no eager / CPU fallback) and the optional per-launch instrumentation."""
import ctypes as C

import torch


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the handle without a torch.cuda.Stream object around it (what torch's own
                                                                         # compiled-kernel launchers call): 0.4 us instead of 6.5 us per launch


def _stream():
    """The current HIP stream of the current device, as the C ABI takes it."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-launch instrumentation used by bench.py's roofline pass: when PROFILE is a list,
# every wrapper appends (kernel_name, algorithmic_flops, algorithmic_bytes, start_event, end_event, layer_name)
# with HIP events recorded on the stream the kernel is launched on.
PROFILE = None   # read and set as `v2x_sim_amd.ops.PROFILE` (a property of that module, see the end of ops.py)


def profile_list():
    return PROFILE


class _Prof:
    __slots__ = ("rec",)

    def __init__(self, name, flops, nbytes, layer=None):
        self.rec = None
        if profile_list() is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.rec = (name, float(flops), float(nbytes), e0, e1, layer or name)
            e0.record(torch.cuda.current_stream())

    def done(self):
        if self.rec is not None:
            self.rec[4].record(torch.cuda.current_stream())
            profile_list().append(self.rec)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the MI355X (cuda) device; the v2x_sim_amd hot path has no CPU "
                           "fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return C.c_void_p(t.data_ptr())


def _dev_opt(t, dtype, name):
    return None if t is None else _dev(t, dtype, name)
