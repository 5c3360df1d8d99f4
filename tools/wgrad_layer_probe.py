#!/usr/bin/env python3
"""Per-layer time of the weight-gradient kernel at a training batch (default 40 maps), against both roofs, and its sensitivity to the number of workgroups that share a
(co tile, ci tile) pair (n_split): which layers are far from what, and is the grid the lever?   python3 tools/wgrad_layer_probe.py [maps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from v2x_sim_amd import _lib  # noqa: E402

LAYERS = [("conv_pre_2 / conv8_2 / heads 32->32 @256", 32, 32, 256), ("conv8_1 96->32 @256", 96, 32, 256), ("conv1_1 (zero-inserted dy) 32->64 @256", 32, 64, 256),
          ("conv1_2 / conv7_2 64->64 @128", 64, 64, 128), ("conv7_1 192->64 @128", 192, 64, 128), ("conv2_1 (zero-ins.) 64->128 @128", 64, 128, 128),
          ("conv2_2 / conv6_2 128->128 @64", 128, 128, 64), ("conv6_1 384->128 @64", 384, 128, 64), ("conv3_1 (zero-ins.) 128->256 @64", 128, 256, 64),
          ("conv3_2 / conv5_2 256->256 @32", 256, 256, 32), ("conv5_1 768->256 @32", 768, 256, 32), ("conv4_1 (zero-ins.) 256->512 @32", 256, 512, 32)]


def main():
    maps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("%-44s %7s %9s %8s %8s %9s   n_split sweep (us)" % ("layer (%d maps)" % maps, "n_split", "us", "TFLOP/s", "unique", "GB/s uniq"))
    for name, cin, cout, hw in LAYERS:
        x = torch.randn(maps, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn(maps, hw, hw, cout, generator=g).to(torch.bfloat16).to(dev)
        ns0 = lib.v2x_conv3x3_wgrad_splits(maps, hw, hw, cin, cout)
        tiles = maps * (hw // 8) * (hw // 32)
        res = {}
        rows32 = cout % 64 != 0
        for ns in sorted(set([ns0, max(2 if rows32 else 1, ns0 // 2), min(ns0 * 2, (2 if rows32 else 1) * tiles), min(ns0 * 4, (2 if rows32 else 1) * tiles)])):
            if rows32 and ns % 2:
                continue
            ws = torch.empty((ns, cout, 3, 3, cin), dtype=torch.float32, device=dev)
            for _ in range(2):
                rc = lib.v2x_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), maps, hw, hw, cin, cout, ws.data_ptr(), ns, s)
            assert rc == 0, lib.v2x_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.v2x_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), maps, hw, hw, cin, cout, ws.data_ptr(), ns, s)
            e1.record()
            torch.cuda.synchronize()
            res[ns] = e0.elapsed_time(e1) * 200.0
            del ws
        us = res[ns0]
        flops = 2.0 * maps * hw * hw * cout * 9 * cin
        uniq = (x.numel() + dy.numel()) * 2
        print("%-44s %7d %9.1f %8.0f %7.0fM %9.0f   %s" % (name, ns0, us, flops / us / 1e6, uniq / 1e6, uniq / us / 1e3, "  ".join("%d: %.1f" % kv for kv in sorted(res.items()))))
        del x, dy


if __name__ == "__main__":
    main()
