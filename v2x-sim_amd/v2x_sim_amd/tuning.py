"""Experiment / engine switches of the package, read from the environment ONCE (at import for the host-side ones, at first use inside
libv2x_amd.so for the kernel-selection ones) and changed afterwards only through set() -- nothing on the hot path calls os.environ.

Kernel-selection switches live in the library (include/v2x_amd.h: v2x_tuning_set / v2x_tuning_get; defaults = the measured-fastest forms):
    STREAM_WAVES STREAM_G STREAM_WT STORE_X4 STREAM_PERSIST STREAM_WIDE WIDE3 HALO_PP VOXELIZE_LDS WARP_LDS S2_G GRU_XCD_WALK HALO_XCD WGRAD_TR BN_PARTIAL_T WGRAD_REDUCE4 CONV1X1
    (STORE_X4 1: 16-byte output stores -- two channel tiles exchanged between the k-slot quarters with v_permlane16_swap_b32 -- in every bf16 epilogue
     that has the form; 0: 8-byte stores, same bytes and values)
Host-side switches (this module):
    CONV_PAIR   1  conv_pre_1 -> conv_pre_2 as one launch from the bit grid; 0: two launches
    SMALL_BATCH 2  latency dispatch: split-K for the streamed layers when a launch has fewer tiles than CUs (ops.small_batch_splitk) and the 1-tap
                   stride-2 kernel below four tiles per CU; results differ from the default kernels by fp32 summation order -- and, for conv5_1 / conv6_1,
                   by the weight form: a split launch multiplies the 9-tap bf16 weights, the throughput launch the pre-summed parity-class weights
                   (ops.py header; bounded in tests/test_gpu_bits_input.py).  0: never; 1: every
                   launch of the process (explicit pin, also for the sharded runners); 2: only inside `with ops.latency_dispatch():`, which the
                   plain single-GPU model classes enter in forward() -- the sharded runners never do (R-rank == 1-rank bitwise)
    SPLITK_TARGET 320  the number of workgroups a declared latency launch's split-K aims at (ops.small_batch_splitk; launches with >= max(200, 5/8 of it) tiles
                   are never split); the training graph passes its own (TRAIN_SPLITK)
    TRAIN_HIP   1  training graph on the hand-written kernels (train/hip_graph.py: bf16 NHWC activations, fp32 master weights); 0: the fp32
                   PyTorch-ROCm (MIOpen) graph of train/graph.py -- upstream's precision, 5x slower
    TRAIN_GRAPH 1  with TRAIN_HIP: FaFModule.step replays the whole step (forward, loss, backward, optimizer) as ONE hipGraph where that is possible -- FaFNet
                   always, V2VNet while the agent table equals the captured one, an optimizer that is capturable (train.loop.make_optimizer builds one; a plain
                   `optim.Adam(params, lr=float)` is not: eager) -- one captured step per batch shape, kept on the optimizer; anything else runs eagerly.  0: always eager
                   (round 6 made 1 the default: 4.1-4.3 ms instead of 5.0-6.4 ms per 10-map FaFNet step, the eager step is host-bound)
    WARP_HIP    1  with TRAIN_HIP: the cross-agent warp of the fusion stage (forward + data gradient) on v2x_warp_affine_f32 / _bwd_f32;
                   0: F.grid_sample and its atomic-scatter backward
    WARP_XCD    1  inference warp + fuse: the output maps of one frame on one XCD (v2x_warp_fuse_ordered: each source map is fetched once per frame
                   instead of once per ego; bit-identical); 0: the plain grid
    SEG_FUSE    1  segmentation models: conv8_2 and the 1x1 class head as one halo launch (the 32-channel map never reaches HBM; bit-identical); 0: two layers
    TRAIN_GATES_HIP 1  with TRAIN_HIP: the ConvGRU's gate arithmetic of the fusion stage as one launch forward and one backward (csrc/gru_train.hip)
    TRAIN_LOSS_HIP 1  with TRAIN_HIP: the detection loss and its gradients as three launches of csrc/det_loss.hip (0: ~45 PyTorch-ROCm ops)
    TRAIN_PACK_BATCH 1  with TRAIN_HIP: the packed weights of all layers rebuilt by ONE launch after an optimizer step (0: one launch per layer)
    PARITY_CLASS 3  pack the decoder `_1` layers the parity-class kernels cover with pre-summed 2x2-tap weights for the x2-upsampled source (-37 % MACs;
                   the sums are formed in fp32 and rounded to bf16 once): 2 = conv8_1 (w_layout 3, resident weights) and conv5_1 / conv6_1 (w_layout 4,
                   streamed weights); 3 = also conv7_1 (w_layout 4 with 64-row tiles, the two-tiles-per-workgroup kernel); 1 = conv8_1 only; 0 = the 9-tap forms
                   everywhere.  Read when a model is packed
    TAIL_FUSE 1  conv8_2 and the detection heads as ONE launch (conv_tail.hip: conv8_2's output never leaves the CU; bit-identical to the two launches);
                   0 = two launches.  Read at every forward
    TRAIN_HEAD_PACK 1  with TRAIN_HIP: the heads' fp32 logit gradients -> bf16, channel padding and bias gradient in one pass (v2x_cast_pad_chsum_f32); 0: torch ops
    TRAIN_BN_BIAS_ZERO 1  with TRAIN_HIP: the bias gradient of a convolution in front of a batch-statistics BatchNorm is returned as its exact value, 0 (the BN removes
                   the batch mean: the computed value is the rounding residue of a cancelling sum); 0: the residue, accumulated by the BN backward kernel
    TRAIN_UPCAT_CONV 1  with TRAIN_HIP: conv8_1 (64 upsampled + 32 skip channels -> 32, full resolution) forward on the two-source halo kernel and its data gradient
                   as two halo launches (32 -> 64, 32 -> 32) instead of the gather kernel both ways; 0: the gather kernel on the concatenated map
    TRAIN_SPLITK 480  with TRAIN_HIP: the 3x3 layers' forward and data-gradient launches are declared small-batch launches (train/hip_graph.py::_run_layer): a streamed
                   layer with fewer tiles than 5/8 of this many workgroups splits its chunk ranges to reach about this many (the deep layers of a 10- or 20-map
                   step; swept 160 ... 800, profiles/r06_train_switch_ab_4.txt); 1: the one-frame inference rule (320 / 200); 0: one workgroup per tile at any batch
    TRAIN_V2V_NHWC 1  with TRAIN_HIP: V2VNet's message-passing rounds on the bf16 NHWC maps (csrc/v2v_train.hip: message, input convolution, gates -- four launches
                   forward, four backward per round) when every frame of the batch has the same number of agents; 0: the fp32 NCHW graph around the warp / gates kernels
    TRAIN_ADAM_HIP 1  a plain torch.optim.Adam handed to FaFModule / SegModule / make_optimizer / GraphedTrainStep steps on v2x_adam_step_f32 (train/optim.py: same state,
                   same update; one launch per 72 tensors); 0: torch's own step
    TRAIN_HIP_CONV 0  only the eligible 3x3 layers of the fp32 graph on the kernels (the first step of row f-3, kept for its tests)
Retired in round 6 (their alternate forms had been measured slower for two rounds or more and no test or tool exercised them): S2_RESIDENT, S2_T16, PP_64, UPCAT_HIP.
The tests use the `tune` fixture (tests/conftest.py), which restores every value it touched."""
import ctypes as C
import os

_HOST_DEFAULTS = {"CONV_PAIR": 1, "TRAIN_HIP": 1, "TRAIN_GRAPH": 1, "TRAIN_HIP_CONV": 0, "SMALL_BATCH": 2, "WARP_HIP": 1, "WARP_XCD": 1, "SEG_FUSE": 1, "TRAIN_PACK_BATCH": 1, "TRAIN_LOSS_HIP": 1, "TRAIN_GATES_HIP": 1, "PARITY_CLASS": 3, "TAIL_FUSE": 1, "TRAIN_HEAD_PACK": 1, "TRAIN_BN_BIAS_ZERO": 1, "TRAIN_UPCAT_CONV": 1, "TRAIN_SPLITK": 480, "TRAIN_V2V_NHWC": 1, "TRAIN_ADAM_HIP": 1, "SPLITK_TARGET": 320}
LIBRARY_SWITCHES = ("STREAM_WAVES", "STREAM_G", "STREAM_WT", "STORE_X4", "STREAM_PERSIST", "STREAM_WIDE", "WIDE3", "HALO_PP",
                    "VOXELIZE_LDS", "WARP_LDS", "S2_G", "GRU_XCD_WALK", "HALO_XCD", "WGRAD_TR", "BN_PARTIAL_T", "WGRAD_REDUCE4", "CONV1X1")


def _env_int(name, default):
    v = os.environ.get("V2X_" + name, "")
    try:
        return int(v) if v != "" else default
    except ValueError:
        return default


_host = {k: _env_int(k, d) for k, d in _HOST_DEFAULTS.items()}


def _norm(name):
    name = name.upper()
    return name[4:] if name.startswith("V2X_") else name


def get(name):
    name = _norm(name)
    if name in _host:
        return _host[name]
    from . import _lib
    out = C.c_int(0)
    _lib.check(_lib.load().v2x_tuning_get(name.encode(), C.byref(out)), "v2x_tuning_get(%s)" % name)
    return out.value


def set(name, value):  # noqa: A001 - mirrors the C entry's name
    """-> the previous value."""
    name = _norm(name)
    old = get(name)
    if name in _host:
        _host[name] = int(value)
    else:
        from . import _lib
        _lib.check(_lib.load().v2x_tuning_set(name.encode(), int(value)), "v2x_tuning_set(%s)" % name)
    return old
