"""Calibration probes (include/v2x_amd.h: v2x_calib_stream / v2x_calib_mfma): what THIS box sustains on a pure streaming kernel at the read :
write mixes of the HBM-bound layers and on a register-resident MFMA loop with random operands.  bench.py prints the result as `calibration`
so that roofline fractions and rounds can be compared box-free; DESIGN.md section 6 grades the HBM-bound kernels against the measured mixes
next to the 8 TB/s datasheet figure."""
import ctypes as C

import torch

from . import _lib

MIXES = {"copy_1_1": (1, 1), "copy_1_3": (1, 3), "copy_2_1": (2, 1), "copy_4_1": (4, 1), "read_only": (1, 0), "write_only": (0, 1)}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_rate(n_read, n_write, total_bytes=4 << 30, nontemporal=False, wg_per_cu=8, reps=3, device=None, bufs=None):
    """-> TB/s of (n_read + n_write) * units * 16 bytes moved by one launch, best of `reps` (HIP events on the launch stream)."""
    lib = _lib.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    units = total_bytes // (16 * (n_read + n_write))
    if bufs is None:
        src = torch.empty(max(n_read, 1) * units * 4, dtype=torch.int32, device=device).random_()
        dst = torch.empty(max(n_write, 1) * units * 4, dtype=torch.int32, device=device)
    else:
        src, dst = bufs
    best = None
    for _ in range(reps + 1):       # (the first launch is the warm-up)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.v2x_calib_stream(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), units, n_read, n_write, int(bool(nontemporal)),
                                        wg_per_cu, _stream()), "v2x_calib_stream")
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or _ == 1 else min(best, ms)
    return (n_read + n_write) * units * 16 / (best * 1e-3) / 1e12


def mfma_rate(ms_budget=120.0, random_operands=True, shape32=False, device=None):
    """-> (TFLOP/s, sustained shader clock in MHz) of the register-resident MFMA loop run back to back for ~ms_budget (the power manager
    settles within tens of ms; a 5-ms burst would read the boost clock)."""
    lib = _lib.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    scratch = torch.zeros(512, dtype=torch.float32, device=device)
    clocks = torch.zeros(2, dtype=torch.int64, device=device)
    iters = 20000                   # ~5 ms per launch at 2.2 PFLOP/s
    fl = C.c_double(0.0)

    def launch(seed):
        _lib.check(lib.v2x_calib_mfma(C.c_void_p(scratch.data_ptr()), iters, seed if random_operands else 0, int(bool(shape32)),
                                      C.c_void_p(clocks.data_ptr()), C.byref(fl), _stream()), "v2x_calib_mfma")
    launch(1)
    torch.cuda.synchronize()
    n = max(2, int(ms_budget / 5.0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(n // 2):         # first half: settle the clocks, untimed
        launch(100 + i)
    e0.record()
    for i in range(n - n // 2):
        launch(200 + i)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1)
    c = clocks.tolist()
    return fl.value * (n - n // 2) / (ms * 1e-3) / 1e12, (100.0 * c[0] / c[1]) if c[1] else None


def calibrate(device=None, total_bytes=2 << 30, mfma_ms=120.0):
    """The record bench.py prints: {"mfma_tflops", "sclk_mhz", "copy_1_1_tbs", "copy_1_3_tbs", "copy_2_1_tbs", "copy_4_1_tbs"} (+ method)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    # one pair of buffers large enough for every mix at this byte count
    src = torch.empty(total_bytes // 4, dtype=torch.int32, device=device).random_()
    dst = torch.empty(total_bytes // 4, dtype=torch.int32, device=device)
    out = {}
    for name in ("copy_1_1", "copy_1_3", "copy_2_1", "copy_4_1"):
        r, w = MIXES[name]
        # the ceiling is the best persistent-grid size: 1:1 peaks at 4 workgroups per CU, the write-heavy mixes at 2 (profiles/r04_hbm_mix_probe.txt)
        out[name + "_tbs"] = max(stream_rate(r, w, total_bytes, False, wg, 2, device, (src, dst)) for wg in (2, 4, 8))
    del src, dst
    tf, mhz = mfma_rate(mfma_ms, True, False, device)
    out["mfma_tflops"] = tf
    out["sclk_mhz"] = mhz
    out["method"] = ("v2x_calib_stream: %d MiB per launch, 16 B per lane, R read + W write streams, default cache policy, best of 2 / 4 / 8 workgroups per CU x 2 launches "
                     "after a warm-up; v2x_calib_mfma: register-resident v_mfma_f32_16x16x32_bf16 loop, random operands, 2 waves per SIMD, ~%d ms "
                     "back to back (second half timed); sclk = s_memtime / s_memrealtime over one wave's loop" % (total_bytes >> 20, int(mfma_ms)))
    return out
