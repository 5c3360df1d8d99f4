"""Host-side sanitizer run ("sanitizers on the CPU build only" -- GPU ASan / XNACK are not available on this pool).

The library's HOST code (argument validation, the weight packers of pack.hip, the error channel of api.hip) is rebuilt with
-fsanitize=address,undefined (the device halves of the files ignore the flag), the plain-C caller tests/c_abi_smoke.c is built the same way and
its host-only part runs: every element of the three packed weight layouts is written and checked against the index formulas of
include/v2x_amd.h, plus the entry points' rejection paths.  Any out-of-bounds write in a packer, signed overflow in an index computation or
misaligned access fails the run."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2x-sim_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang"
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs hipcc and its clang")
def test_host_code_under_asan_and_ubsan(tmp_path):
    tmp = str(tmp_path)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    procs = []
    for i, src in enumerate(srcs):
        obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
        procs.append((src, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-option-ignored"] + SAN +
                                            ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        if len(procs) % 8 == 0:          # 8 CPUs here
            for _, p in procs[-8:]:
                p.wait()
    for src, p in procs:
        out, err = p.communicate()
        assert p.returncode == 0, (src, err[-2000:])
    lib = os.path.join(tmp, "libv2x_amd.so")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC"] + SAN + sorted(glob.glob(os.path.join(tmp, "*.o"))) + ["-o", lib],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    exe = os.path.join(tmp, "c_abi_smoke")
    p = subprocess.run([CLANG, "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")] + SAN +
                       [os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-L" + tmp, "-lv2x_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                        "-Wl,-rpath," + tmp, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, "--pack-only"], capture_output=True, text=True, timeout=300, env=env)
    report = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0 and "pack OK" in p.stdout, report
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, report
