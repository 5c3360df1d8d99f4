"""Per-layer timing of one V2VNet step (HIP events around every launch). usage: layer_profile.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import numpy as np, torch
from v2x_sim_amd import ops
from v2x_sim_amd.configs import Config
from v2x_sim_amd.models.det import V2VNet
from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet
from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses
Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
model = init_synthetic_weights(V2VNet(Config("test")), seed=0).to(dev)
shard = AgentShard(5, Bt, 0, 1); runner = ShardedV2VNet(model, shard)
points = torch.from_numpy(synthetic_points(5 * Bt, 65536, seed=1)).to(dev)
n_pts = torch.full((5 * Bt,), 65536, dtype=torch.int32, device=dev)
trans = torch.from_numpy(synthetic_poses(Bt, 5, seed=2)).to(dev)
plan = shard.fusion_plan(torch.full((Bt, 5), 5), dev)
# monkeypatch conv2d to tag records with the layer name
orig = ops.conv2d
names = []
def conv2d(pc, *a, **k):
    names.append(pc.name); return orig(pc, *a, **k)
ops.conv2d = conv2d
import v2x_sim_amd.models.det.base as B
with torch.no_grad():
    for _ in range(3): runner.forward_points(points, n_pts, trans, plan)
    torch.cuda.synchronize()
    reps = 5; agg = {}
    for _ in range(reps):
        names.clear(); ops.PROFILE = []
        runner.forward_points(points, n_pts, trans, plan)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        ci = 0
        for name, fl, by, e0, e1 in recs:
            key = name
            if name.startswith("conv"):
                key = names[ci] + "  " + name.replace("conv_igemm_kernel", "").replace("conv3x3_halo_sb_kernel", "halo_sb").replace("conv3x3_halo_kernel", "halo"); ci += 1
            a = agg.setdefault(key, [0.0, 0.0, 0.0]); a[0] += e0.elapsed_time(e1) / reps; a[1] += fl / reps; a[2] += by / reps
tot = sum(v[0] for v in agg.values())
print("frames/step %d  total kernel time %.3f ms" % (Bt, tot))
for k, (ms, fl, by) in agg.items():
    print("%-46s %8.1f us  %7.1f TF/s  %7.1f GB/s  AI %6.0f" % (k, ms * 1e3, fl / ms / 1e9, by / ms / 1e6, fl / max(by, 1)))
