"""ctypes binding of libv2x_amd.so (the C ABI declared in include/v2x_amd.h).

There is NO fallback: if the HIP library is missing or a call fails the product
path raises.  (The CPU oracle under /oracle is test infrastructure and is never
imported from this package.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libv2x_amd.so")

V2X_EPI_BF16, V2X_EPI_F32, V2X_EPI_GRU, V2X_EPI_DET = 0, 1, 2, 3
V2X_FUSE_WSUM, V2X_FUSE_MEAN, V2X_FUSE_MAX = 0, 1, 2
ABI_VERSION = 18


class ConvDesc(C.Structure):
    """Mirror of `struct v2x_conv_desc` (include/v2x_amd.h)."""
    _fields_ = [
        ("in0", C.c_void_p), ("in1", C.c_void_p),
        ("C0", C.c_int32), ("C1", C.c_int32), ("up0", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("Cout", C.c_int32), ("w_rows", C.c_int32), ("w_kpad", C.c_int32),
        ("weight", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("epilogue", C.c_int32), ("relu", C.c_int32),
        ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
        ("out2", C.c_void_p), ("split", C.c_int32), ("out2_cstride", C.c_int32),
        ("w_layout", C.c_int32), ("Cout2", C.c_int32),
        ("weight2", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("relu2", C.c_int32),
        ("in_format", C.c_int32), ("in_zbits", C.c_int32),
        ("det_counts", C.c_void_p), ("det_thr", C.c_float), ("det_cap", C.c_int32),
        ("splitk", C.c_int32), ("splitk_ws", C.c_void_p), ("small_batch", C.c_int32),
    ]


class PackJob(C.Structure):
    """Mirror of `struct v2x_pack_job` (include/v2x_amd.h): one packing of the batched device packer."""
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p),
                ("rows_src", C.c_int32), ("cin", C.c_int32), ("cin_p", C.c_int32), ("taps", C.c_int32), ("K", C.c_int32), ("w_kpad", C.c_int32),
                ("tile", C.c_int32), ("cout", C.c_int32), ("layout", C.c_int32), ("transform", C.c_int32),
                ("src_stride", C.c_int32), ("reserved0", C.c_int32),
                ("groups", C.c_int64), ("data_groups", C.c_int64), ("block_begin", C.c_int64)]


class PackSpec(C.Structure):
    """Mirror of `struct v2x_pack_spec` (include/v2x_amd.h)."""
    _fields_ = [("Cout", C.c_int32), ("Cin", C.c_int32), ("ksize", C.c_int32), ("cin_pad", C.c_int32),
                ("w_layout", C.c_int32), ("epilogue", C.c_int32), ("chain", C.c_int32), ("c_up", C.c_int32),
                ("src_rows", C.c_int32), ("src_row0", C.c_int32)]


ADAM_MAX_TENSORS = 72     # V2X_ADAM_MAX_TENSORS


class AdamTensors(C.Structure):
    """Mirror of `struct v2x_adam_tensors` (include/v2x_amd.h): a host table of device pointers."""
    _fields_ = [("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS), ("step", C.c_void_p * ADAM_MAX_TENSORS), ("numel", C.c_int64 * ADAM_MAX_TENSORS)]


# name -> (restype, argtypes); every symbol include/v2x_amd.h declares
SIGNATURES = {
    "v2x_abi_version": (C.c_int, []),
    "v2x_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "v2x_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "v2x_last_error": (C.c_char_p, []),
    "v2x_voxelize_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                     C.c_void_p, C.c_void_p]),
    "v2x_voxelize_fused_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "v2x_bits_to_dense_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_bits_to_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "v2x_dense_f32_to_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p]),
    "v2x_bits_to_indices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_indices_to_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "v2x_pack_conv_size": (C.c_size_t, [C.POINTER(PackSpec), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "v2x_pack_conv_device": (C.c_int, [C.POINTER(PackSpec), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_zero_insert_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_gru_gates_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_gru_gates_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_det_loss_workspace_size": (C.c_longlong, [C.c_longlong]),
    "v2x_det_loss_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_det_loss_backward": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong, C.c_float, C.c_float] + [C.c_void_p] * 7),
    "v2x_pack_conv_device_job": (C.c_int, [C.POINTER(PackSpec), C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(PackJob), C.POINTER(C.c_int64)]),
    "v2x_pack_conv_device_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "v2x_channel_sum_workspace_size": (C.c_longlong, [C.c_longlong, C.c_int]),
    "v2x_channel_sum_bf16": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_cast_pad_chsum_workspace_size": (C.c_longlong, [C.c_longlong, C.c_int]),
    "v2x_cast_pad_chsum_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_adam_step_f32": (C.c_int, [C.POINTER(AdamTensors), C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "v2x_v2v_message_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]),
    "v2x_v2v_message_bwd_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_gru_gates_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_gru_gates_nhwc_workspace_size": (C.c_longlong, [C.c_longlong, C.c_int]),
    "v2x_gru_gates_nhwc_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_warp_affine_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_warp_affine_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_upcat_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_upcat_bwd_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_conv3x3_wgrad_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_pack_conv": (C.c_int, [C.POINTER(PackSpec), C.c_void_p, C.c_void_p]),
    "v2x_pack_chain_1x1": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_pack_gru_bias": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_fold_bn": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                              C.c_void_p, C.c_void_p]),
    "v2x_conv_tile_rows": (C.c_int, [C.c_int, C.c_int]),
    "v2x_conv_stream_tile_rows": (C.c_int, [C.c_int, C.c_int]),
    "v2x_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "v2x_conv2d_pair": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_void_p]),
    "v2x_conv3x3_wgrad_splits": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "v2x_conv3x3_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "v2x_bn_train_workspace_size": (C.c_longlong, [C.c_longlong, C.c_int]),
    "v2x_bn_train_forward": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_bn_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_bn_dxsum_workspace_size": (C.c_longlong, [C.c_longlong, C.c_int]),
    "v2x_bn_train_backward_dxsum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_warp_fuse": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_warp_fuse_ordered": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "v2x_attn_handshake": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_pixel_weighted_fuse": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_det_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_det_postprocess_rotated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_det_nms_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_rotated_iou": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "v2x_match_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2x_calib_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "v2x_calib_mfma": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "v2x_seg_argmax_confusion": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
}

_lib = None


class V2XLibraryError(RuntimeError):
    pass


def load():
    """Load libv2x_amd.so, bind every declared symbol and check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; it must be the one already resident when libv2x_amd.so resolves
    # libamdhip64, otherwise the process holds two runtimes and every pointer torch allocated is foreign to ours
    # (observed: hipMemsetAsync fails with "invalid value" when this library was loaded before `import torch`).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise V2XLibraryError(
            "libv2x_amd.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C v2x-sim_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise V2XLibraryError("libv2x_amd.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.v2x_abi_version()
    if ver == -ABI_VERSION and os.environ.get("V2X_ALLOW_PROBE_BUILD") == "1":
        ver = ABI_VERSION       # an instrumented timing build (tools/probes/): garbage results by design, loaded only on request (the probe scripts set this)
    if ver == -ABI_VERSION:
        raise V2XLibraryError("libv2x_amd.so contains an INSTRUMENTED kernel object (a tools/probes/ timing build: its results are garbage) -- rebuild the product "
                              "(`make -C v2x-sim_amd/csrc` after removing build/<name>.o), or set V2X_ALLOW_PROBE_BUILD=1 for a timing experiment")
    if ver != ABI_VERSION:
        raise V2XLibraryError("ABI mismatch: library %d, binding %d" % (lib.v2x_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().v2x_last_error()
        raise V2XLibraryError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
