import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "v2x-sim_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_c_lib():
    """Builds (if needed) and loads the plain-C voxelizer restatement under oracle/."""
    import ctypes
    so = os.path.join(ROOT, "oracle", "_build", "libvoxelize_ref.so")
    src = os.path.join(ROOT, "oracle", "voxelize_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    lib.oracle_voxelize_occupy.restype = ctypes.c_int64
    lib.oracle_occupancy_indices.restype = ctypes.c_int64
    return lib


@pytest.fixture(scope="session")
def device():
    import torch
    return torch.device("cuda:0")


class _Tune:
    """tune("STREAM_G", 0) sets a kernel-selection / engine switch (v2x_sim_amd.tuning); tune.reset("STREAM_G") puts back the value it
    had when the test started.  Everything touched is restored at teardown."""

    def __init__(self):
        self._orig = {}

    def __call__(self, name, value):
        from v2x_sim_amd import tuning
        old = tuning.set(name, int(value))
        self._orig.setdefault(name.upper(), old)

    def reset(self, name):
        from v2x_sim_amd import tuning
        if name.upper() in self._orig:
            tuning.set(name, self._orig[name.upper()])

    def restore(self):
        from v2x_sim_amd import tuning
        for k, v in self._orig.items():
            tuning.set(k, v)
        self._orig.clear()


@pytest.fixture
def tune():
    t = _Tune()
    yield t
    t.restore()
