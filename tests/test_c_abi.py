"""tests/c_abi_smoke.c -- a compiled, non-Python caller of libv2x_amd.so.  Built with gcc as C99
against include/v2x_amd.h: proves the header is C-clean, the packers give a host without torch everything v2x_conv2d
needs, and (on the GPU) that the three weight layouts drive three kernels to the same result."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "v2x-sim_amd", "v2x_sim_amd", "lib")


def build(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    if not os.path.exists(os.path.join(LIBDIR, "libv2x_amd.so")):
        pytest.fail("libv2x_amd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(str(tmp_path), "c_abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-L" + LIBDIR, "-lv2x_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_c_caller_builds_and_packs_on_the_host(tmp_path):
    p = subprocess.run([build(tmp_path), "--pack-only"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "pack OK" in p.stdout


@pytest.mark.gpu
def test_c_caller_runs_three_layouts_through_v2x_conv2d(tmp_path):
    p = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=300)
    print(p.stdout)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "C ABI smoke OK" in p.stdout
