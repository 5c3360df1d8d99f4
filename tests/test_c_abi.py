"""tests/c_abi_smoke.c -- a compiled, non-Python caller of libv2x_amd.so.  Built with gcc as C99
against include/v2x_amd.h: proves the header is C-clean, the packers give a host without torch everything v2x_conv2d
needs, and (on the GPU) that the three weight layouts drive three kernels to the same result."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "v2x-sim_amd", "v2x_sim_amd", "lib")


def build(tmp_path, src="c_abi_smoke"):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    if not os.path.exists(os.path.join(LIBDIR, "libv2x_amd.so")):
        pytest.fail("libv2x_amd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(str(tmp_path), src)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", src + ".c"), "-L" + LIBDIR, "-lv2x_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
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


# ---------------------------------------------------------------------------------------------------------------------------------------
# tests/c_abi_frame.c: the WHOLE path (points -> voxel scatter -> encoder -> warp + ConvGRU -> decoder -> heads -> post-processing) driven from
# raw checkpoint tensors by a C99 host through the C ABI alone.  The tensor-list file format is documented in that source.
_DT = {"float32": 0, "float64": 1, "int32": 2}


def _write_tensors(path, tensors):
    import numpy as np
    import struct
    with open(path, "wb") as f:
        f.write(b"V2XF" + struct.pack("<i", len(tensors)))
        for name, a in tensors:
            a = np.ascontiguousarray(a)
            assert a.dtype.name in _DT and a.ndim <= 6 and len(name) < 64, (name, a.dtype, a.shape)
            f.write(name.encode().ljust(64, b"\0"))
            f.write(struct.pack("<8i", _DT[a.dtype.name], a.ndim, *(list(a.shape) + [0] * (6 - a.ndim))))
            f.write(a.tobytes())


def _read_tensors(path):
    import numpy as np
    import struct
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"V2XF"
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            name = f.read(64).rstrip(b"\0").decode()
            hdr = struct.unpack("<8i", f.read(32))
            dt = {0: np.float32, 1: np.float64, 2: np.int32}[hdr[0]]
            shape = hdr[2:2 + hdr[1]]
            out[name] = np.frombuffer(f.read(int(np.prod(shape)) * np.dtype(dt).itemsize), dtype=dt).reshape(shape)
    return out


def test_c_frame_caller_builds(tmp_path):
    """CPU box: the whole-path C host compiles as strict C99 against the header and links against the library (no run: it needs the GPU)."""
    assert os.path.exists(build(tmp_path, "c_abi_frame"))


@pytest.mark.gpu
@pytest.mark.parametrize("A,X,n_pts", [(3, 64, 4096), (5, 256, 65536)])
def test_c_caller_runs_the_whole_path_from_points_to_detections(tmp_path, tune, A, X, n_pts):
    """One collaborative V2VNet frame (3 agents on a 64 x 64 x 13 grid; 5 agents x 65 536 points on the full 256 x 256 x 13 grid) run by the C host from the raw
    fp32 checkpoint: its occupancy grid is bit-exact against the oracle's voxeliser, its logits and detections are BIT-IDENTICAL to the Python
    host's (same kernels, same packing -- the C packers and the C restatement of ops.run_layer's shape rules agree with packing.py / ops.py)
    and inside the end-to-end tolerance of the bf16-emulating oracle (tests/test_gpu_models.py: TOL_EMU)."""
    import numpy as np
    import torch
    from oracle import coperception_ref as R
    from oracle import voxelize_ref as VR
    from v2x_sim_amd import ops
    from v2x_sim_amd.configs import Config
    from v2x_sim_amd.models.det import V2VNet
    from v2x_sim_amd.utils import postprocess
    from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_poses
    dev = torch.device("cuda:0")
    cfg = Config("test")
    half = X * 0.25 / 2
    ext = ((-half, half), (-half, half), (-3.0, 2.0))
    grid = ops.VoxelGrid(area_extents=ext)
    assert grid.dims == (X, X, 13)
    pm = init_synthetic_weights(V2VNet(cfg, num_agent=A), seed=3)
    rng = np.random.default_rng(A)
    pts = np.zeros((A, n_pts, 4), np.float32)
    pts[..., :2] = rng.uniform(-half * 1.2, half * 1.2, (A, n_pts, 2))
    pts[..., 2] = rng.uniform(-5, 4, (A, n_pts))
    cnt = np.full((A,), n_pts, np.int32)
    cnt[1] = n_pts - 7                                            # a ragged cloud
    T = synthetic_poses(1, A, seed=1)
    T[..., :2, 3] *= X / 256.0
    # anchors of the (possibly small) grid, as utils.postprocess.build_anchor_map lays them out
    class _G:                                                     # noqa: E306
        map_dims, voxel_size, area_extents, anchor_size = (X, X, 13), cfg.voxel_size, ext, cfg.anchor_size
    anchors = postprocess.build_anchor_map(_G).reshape(-1, 6).astype(np.float32)
    nms, cap = 0.01, 4096
    # the Python host first (kernel selection by shape only: the latency forms off) -- its logits also fix a score threshold that a few hundred anchors
    # per map pass (random weights: most foreground probabilities sit near 0.5)
    tune("SMALL_BATCH", 0)
    pmd = pm.to(dev)
    nat = torch.full((1, A), A)
    with torch.no_grad():
        py = pmd.forward_points(torch.from_numpy(pts).to(dev), torch.from_numpy(cnt).to(dev), torch.from_numpy(T).to(dev), nat, batch_size=1, grid=grid)
        fg = torch.softmax(py["cls"], -1)[..., 1]
        thr = float(fg.flatten().topk(300 * A).values[-1])
        boxes, scores, index, count = ops.det_postprocess(py["cls"].contiguous(), py["loc"].contiguous(), torch.from_numpy(anchors).to(dev), thr, nms, cap)
    tensors = [("config", np.array([A, 1, X, X, 13, n_pts, 4, cap], np.int32)), ("points", pts), ("n_pts", cnt),
               ("extents", np.array([x for lohi in ext for x in lohi], np.float64)), ("voxel", np.array(grid.voxel, np.float64)),
               ("trans", T.astype(np.float32)), ("anchors", anchors), ("thresholds", np.array([thr, nms], np.float32))]
    tensors += [(k, v.detach().float().cpu().numpy()) for k, v in pm.state_dict().items() if v.dtype.is_floating_point]
    fin, fout = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    _write_tensors(fin, tensors)
    p = subprocess.run([build(tmp_path, "c_abi_frame"), fin, fout], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0 and "C ABI frame OK" in p.stdout, p.stdout + p.stderr
    got = _read_tensors(fout)

    # (1) a1: bit-exact occupancy against the oracle's voxeliser
    ref_bev = np.stack([VR.voxelize_occupy(pts[a, :cnt[a]], VR.VOXEL_SIZE, np.asarray(ext)) for a in range(A)])
    dense = ((got["bits"][..., None].astype(np.int64) >> np.arange(13)) & 1).astype(np.float32)
    assert np.array_equal(dense, ref_bev), "C host: voxel occupancy is not bit-exact"

    # (2) the Python host on the same checkpoint and points: identical bits
    assert np.array_equal(got["cls"].reshape(A, -1, 2), py["cls"].cpu().numpy()), "C host logits differ from the Python host's"
    assert np.array_equal(got["loc"].reshape(A, X, X, 6, 1, 6), py["loc"].cpu().numpy())
    cn = count.cpu().numpy()
    assert np.array_equal(got["count"], cn) and (cn > 0).all() and (cn <= cap).all(), cn
    for a in range(A):
        k = int(cn[a])
        assert np.array_equal(got["boxes"][a, :k], boxes[a, :k].cpu().numpy()) and np.array_equal(got["scores"][a, :k], scores[a, :k].cpu().numpy())
        assert np.array_equal(got["index"][a, :k], index[a, :k].cpu().numpy())

    # (3) parity: the bf16-emulating oracle on the oracle's own occupancy grid, at the end-to-end tolerance of tests/test_gpu_models.py
    om = R.V2VNet(num_agent=A).eval()
    om.load_state_dict(pm.state_dict())
    om.emulate_bf16 = True
    with torch.no_grad():
        ref = om(torch.from_numpy(ref_bev)[:, None], torch.from_numpy(T), nat, batch_size=1)
    for key, g in (("cls", got["cls"].reshape(A, -1, 2)), ("loc", got["loc"].reshape(A, X, X, 6, 1, 6))):
        r = ref[key].numpy()
        scale = float(np.abs(r).max())
        d = np.abs(g - r)
        assert d.max() <= 3e-2 * scale and d.mean() <= 3e-3 * scale, (key, d.max() / scale, d.mean() / scale)
