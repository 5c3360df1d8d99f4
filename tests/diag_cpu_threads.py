"""Diagnostic: oracle V2VNet frame time vs torch thread count on this host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import numpy as np, torch
from oracle import coperception_ref as R, voxelize_ref as VR
from v2x_sim_amd.utils.synthetic import synthetic_points, synthetic_poses
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
om = R.V2VNet().eval()
pts = synthetic_points(5, 65536, seed=1)
T = torch.from_numpy(synthetic_poses(1, 5, seed=2)); nat = torch.full((1, 5), 5)
t0 = time.perf_counter(); bev = np.stack([VR.voxelize_occupy(p) for p in pts])[:, None]; print("voxelize 5 sweeps: %.3fs" % (time.perf_counter() - t0))
bev = torch.from_numpy(bev)
for th in (int(a) for a in sys.argv[1:]):
    torch.set_num_threads(th)
    with torch.no_grad():
        om(bev, T, nat, batch_size=1)
        t0 = time.perf_counter(); om(bev, T, nat, batch_size=1); dt = time.perf_counter() - t0
    print("threads %3d: %.2f s/frame" % (th, dt), flush=True)
