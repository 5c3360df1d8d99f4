#!/usr/bin/env python3
"""Host-link cost of one bench step's input (128 frames x 5 agents x 65536 points x 16 B = 671 MB) from pinned host memory,
for the PCIe-inclusive figure in DESIGN.md.  usage: python tools/pcie_rate.py"""
import torch
n = 128 * 5
host = torch.empty((n, 65536, 4), dtype=torch.float32).pin_memory()
dev = torch.empty_like(host, device="cuda:0")
for _ in range(2):
    dev.copy_(host, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    dev.copy_(host, non_blocking=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("H2D %.1f MB in %.2f ms = %.1f GB/s" % (host.numel() * 4 / 1e6, ms, host.numel() * 4 / ms / 1e6))
