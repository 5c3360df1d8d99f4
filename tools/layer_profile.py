"""Per-layer timing of one V2VNet forward (HIP events around every launch). usage: layer_profile.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch  # noqa: E402
from v2x_sim_amd import ops  # noqa: E402
from v2x_sim_amd.configs import Config  # noqa: E402
from v2x_sim_amd.models.det import V2VNet  # noqa: E402
from v2x_sim_amd.parallel import AgentShard, ShardedV2VNet  # noqa: E402
from v2x_sim_amd.utils.synthetic import init_synthetic_weights, synthetic_points, synthetic_poses  # noqa: E402

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
model = init_synthetic_weights(V2VNet(Config("test")), seed=0).to(dev)
shard = AgentShard(5, Bt, 0, 1)
runner = ShardedV2VNet(model, shard)
points = torch.from_numpy(synthetic_points(5 * Bt, 65536, seed=1)).to(dev)
n_pts = torch.full((5 * Bt,), 65536, dtype=torch.int32, device=dev)
trans = torch.from_numpy(synthetic_poses(Bt, 5, seed=2)).to(dev)
plan = shard.fusion_plan(torch.full((Bt, 5), 5), dev)
with torch.no_grad():
    for _ in range(3):
        runner.forward_points(points, n_pts, trans, plan)
    torch.cuda.synchronize()
    reps, agg = 5, {}
    for _ in range(reps):
        ops.PROFILE = []
        runner.forward_points(points, n_pts, trans, plan)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        for name, fl, by, e0, e1, layer in recs:
            short = name.replace("conv_igemm_kernel", "igemm").replace("conv3x3_halo_sb_kernel", "halo_sb").replace(
                "conv3x3_halo_kernel", "halo").replace("conv3x3_", "").replace("_kernel", "")
            key = layer if layer == name else "%s  %s" % (layer, short)
            a = agg.setdefault(key, [0.0, 0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1) / reps
            a[1] += fl / reps
            a[2] += by / reps
            a[3] += 1
tot = sum(v[0] for v in agg.values())
print("frames/launch %d  total kernel time %.3f ms  (%.0f frames/s by kernel time)" % (Bt, tot, Bt / tot * 1e3))
for k, (ms, fl, by, cnt) in agg.items():
    print("%-52s %8.1f us x%d %7.1f TF/s  %7.1f GB/s  AI %6.0f  %4.1f %%" % (
        k, ms * 1e3, cnt // reps, fl / ms / 1e9, by / ms / 1e6, fl / max(by, 1), 100 * ms / tot))
