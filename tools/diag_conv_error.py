"""Diagnostic: error of the MFMA conv (fp32 epilogue) vs an fp64 reference, next to torch-CPU fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "v2x-sim_amd"))
import torch, torch.nn.functional as F
from v2x_sim_amd import ops, packing
dev = torch.device("cuda:0")
def bf(x): return x.to(torch.bfloat16).float()
for (N, Cin, Cout, H, W) in ((1, 32, 32, 64, 64), (1, 128, 128, 32, 32), (1, 512, 512, 16, 16)):
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(N, Cin, H, W, generator=g).relu())
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5)
    ref64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    ref32 = F.conv2d(x, w, None, 1, 1)
    pc = packing.pack_conv("t", w, torch.ones(Cout), torch.zeros(Cout), relu=False, epilogue=ops.V2X_EPI_F32, device=dev)
    y = ops.conv2d(pc, x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)).cpu().permute(0, 3, 1, 2)
    e_hip = (y.double() - ref64).abs(); e_cpu = (ref32.double() - ref64).abs()
    print("Cin=%d: |ref| mean %.3f  HIP err max %.2e mean %.2e | CPU fp32 err max %.2e mean %.2e" % (
        Cin, ref64.abs().mean(), e_hip.max(), e_hip.mean(), e_cpu.max(), e_cpu.mean()))
    pcb = packing.pack_conv("t", w, torch.ones(Cout), torch.zeros(Cout), relu=False, device=dev)
    yb = ops.conv2d(pcb, x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)).float().cpu().permute(0, 3, 1, 2)
    flips = (yb != bf(ref32)).float().mean()
    print("   bf16-epilogue elements differing from bf16(torch fp32): %.4f%%" % (100 * flips))
