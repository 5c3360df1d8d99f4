"""CPU (cross-compile) checks of code-generation invariants the kernels rely on.

conv_stream.hip counts its own LDS-DMA operations with `s_waitcnt vmcnt(N)`: any compiler-inserted scratch
(spill) load/store inside the kernel would join the same counter and silently corrupt the pipeline, so the
build must keep every conv kernel spill-free.  The halo / gather kernels must really use LDS-DMA
(global_load_lds) -- a silent fallback to register staging is how the first version lost 2x."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2x-sim_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _asm(src, tmp_path):
    out = os.path.join(str(tmp_path), src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                           os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm):
    """name -> (body, metadata dict) for every amdhsa kernel in the file."""
    res = {}
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", asm, flags=re.S | re.M):
        res[m.group(1)] = m.group(0)
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(.*?)\.wavefront_size", asm, flags=re.S):
        meta[m.group(1)] = m.group(2)
    return res, meta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,kernel_prefix,min_kernels", [
    ("conv_stream.hip", "_Z21conv3x3_stream_kernel", 6),
    ("conv_stream.hip", "_Z22conv3x3_stream8_kernel", 3),
    ("conv_stream.hip", "_Z23conv3x3_stream8g_kernel", 2),
    ("conv_stream_s2.hip", "_Z24conv3x3_s2_stream_kernel", 2),
    ("conv_stream_s2.hip", "_Z18conv3x3_s2g_kernel", 2),
    ("conv_stream.hip", "_Z19conv3x3_wide_kernel", 2),
    ("conv_stream.hip", "_Z20conv3x3_wide3_kernel", 1),
    ("conv_stream_s2.hip", "_Z26conv3x3_s2_resident_kernel", 1),
    ("conv_halo.hip", "_Z19conv3x3_halo_kernel", 3),
    ("conv_halo.hip", "_Z22conv3x3_halo_sb_kernel", 2),
    ("conv_halo.hip", "_Z22conv3x3_halo_pp_kernel", 3),
    ("conv_halo_pair.hip", "_Z24conv3x3_pair_bits_kernel", 1),
    ("conv_igemm.hip", "_Z17conv_igemm_kernel", 9),
])
def test_conv_kernels_are_spill_free_and_use_lds_dma(tmp_path, src, kernel_prefix, min_kernels):
    asm = _asm(src, tmp_path)
    bodies, meta = _kernels(asm)
    names = [n for n in bodies if n.startswith(kernel_prefix)]
    assert len(names) >= min_kernels, names
    for n in names:
        body = bodies[n]
        assert "scratch_" not in body, "%s touches scratch (spill) -- breaks the vmcnt bookkeeping" % n
        assert "global_load_lds_dwordx4" in body, "%s lost its LDS-DMA loads" % n
        assert "v_mfma_f32_16x16x32_bf16" in body
        md = meta.get(n, "")
        seg = re.search(r"\.private_segment_fixed_size:\s+(\d+)", md)
        assert seg and int(seg.group(1)) == 0, (n, seg and seg.group(1))
        vg = re.search(r"\.vgpr_count:\s+(\d+)", md)
        # two waves per SIMD -> 256 registers; the chained 64 -> 64 -> 64 4-wave halo kernel (the odd-tile-count fallback of the ping-pong form)
        # holds 158 KiB of LDS = one workgroup per CU = one wave per SIMD and may take a whole SIMD's 512
        limit = 512 if n.startswith("_Z19conv3x3_halo_kernelILi0ELi64ELi64ELi64ELi1E") else 256
        assert vg and int(vg.group(1)) <= limit, (n, vg and vg.group(1))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_stream_kernel_waits_are_counted(tmp_path):
    """Inside the streamed kernel the only vmcnt waits are the hand-placed counted ones (0/2/4/6); in particular the
    step barrier is a bare s_barrier (a __syncthreads-style `s_waitcnt vmcnt(0)` right before it would drain the ring)."""
    asm = _asm("conv_stream.hip", tmp_path)
    bodies, _ = _kernels(asm)
    name = [n for n in bodies if n.startswith("_Z21conv3x3_stream_kernelILi128ELi8ELi32ELi0E")][0]
    body = bodies[name]
    waits = set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
    assert {"6", "4", "0"} <= waits and waits <= {"0", "2", "4", "6"}, waits
    # exactly one barrier in the step loop, and the instruction before it is not a vmcnt(0) drain
    lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(";")]
    idx = [i for i, ln in enumerate(lines) if ln.startswith("s_barrier")]
    assert len(idx) >= 1
    for i in idx:
        assert "vmcnt(0)" not in lines[i - 1], lines[i - 3:i + 1]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_pair_kernel_never_drains_its_stores_inside_the_tile_loop(tmp_path):
    """conv_halo_pair.hip: vmcnt is in-order, so a vmcnt(0) inside the tile loop would wait for the previous tile's output
    stores (that was 6 us per tile before the loop was restructured).  Past the prologue barrier the only vector-memory
    waits are for the next tile's two occupancy words, requested before this tile's stores: vmcnt(9) / vmcnt(8) with the eight
    8-byte stores, vmcnt(5) / vmcnt(4) with the four 16-byte ones (the store form is a TEMPLATE parameter for exactly this
    reason: behind a run-time branch the compiler could not count and drained the stores in every tile)."""
    asm = _asm("conv_halo_pair.hip", tmp_path)
    bodies, _ = _kernels(asm)
    names = sorted(n for n in bodies if n.startswith("_Z24conv3x3_pair_bits_kernel"))
    assert len(names) == 2, names
    for n, allowed in zip(names, ({"8", "9"}, {"4", "5"})):          # <false> (ILb0E) sorts before <true>
        body = bodies[n]
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(";")]
        end = next(i for i, ln in enumerate(lines) if ln.startswith("s_endpgm"))
        bars = [i for i, ln in enumerate(lines[:end]) if ln.startswith("s_barrier")]
        assert len(bars) == 4, bars                       # LUT, prologue, and the two per-tile barriers
        loop = lines[bars[1]:end]
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", "\n".join(loop))
        assert waits and set(waits) <= allowed, (n, waits)
        assert sum(ln.startswith("v_mfma_f32_16x16x32_bf16") for ln in loop) == 132   # 5 x 6 x 2 (layer A) + 3 x 24 (layer B)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_streaming_conv1x1_keeps_its_weights_in_registers(tmp_path):
    """conv1x1.hip (round 6): every instantiation is spill-free, touches no LDS (the weights are MFMA fragments in registers, the pixels come straight from HBM), has no
    barrier, and issues exactly KS x CT MFMAs per 16-pixel fragment in flight (PF of them per loop iteration)."""
    asm = _asm("conv1x1.hip", tmp_path)
    bodies, meta = _kernels(asm)
    names = [n for n in bodies if n.startswith("_Z21conv1x1_stream_kernel")]
    assert len(names) == 36, len(names)            # KS in {1, 2, 4} x CT in {1, 2, 3, 4, 6, 8} x {bf16, fp32}
    for n in names:
        body = bodies[n]
        m = re.match(r"_Z21conv1x1_stream_kernelILi(\d+)ELi(\d+)ELb([01])ELi(\d+)E", n)
        ks, ct, pf = int(m.group(1)), int(m.group(2)), int(m.group(4))
        assert "scratch_" not in body and "ds_read" not in body and "ds_write" not in body and "s_barrier" not in body, n
        assert len(re.findall(r"^\s+v_mfma_f32_16x16x32_bf16", body, flags=re.M)) == ks * ct * pf, n
        seg = re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta.get(n, ""))
        assert seg and int(seg.group(1)) == 0, n
