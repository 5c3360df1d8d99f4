#!/usr/bin/env python3
"""Paired in-process A/B of two builds of libv2x_amd.so: the same launches alternate between the two libraries inside ONE process, so box
drift (clock, temperature: +-4 % between runs on these boxes) cancels.  tools/ab_inproc.sh builds the two variants.
    python3 tools/ab_inproc.py libA.so libB.so
Prints, per layer shape, the mean time of each variant and the paired difference with its standard error."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "v2x-sim_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from v2x_sim_amd import _lib, ops, packing  # noqa: E402


def load_variant(path):
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.v2x_abi_version() == _lib.ABI_VERSION
    return lib


def main():
    libs = [load_variant(sys.argv[1]), load_variant(sys.argv[2])]
    # optional: per-variant tuning switches, e.g.  A:STREAM_WT=1 B:STREAM_WT=2  (include/v2x_amd.h: v2x_tuning_set; set on the variant's OWN
    # library handle at every switch, so the same .so may be passed twice)
    envs = [{}, {}]
    only = None
    for a in sys.argv[3:]:
        if a.startswith("only="):       # only=s2 : run the cases whose name contains the text
            only = a[5:]
            continue
        v, kv = a.split(":", 1)
        k, val = kv.split("=", 1)
        envs["AB".index(v)][k.upper().replace("V2X_", "")] = int(val)
    defaults = {}
    for k in set(envs[0]) | set(envs[1]):
        out = C.c_int(0)
        assert libs[0].v2x_tuning_get(k.encode(), C.byref(out)) == 0, k
        defaults[k] = out.value

    def use(v):
        _lib._lib = libs[v]
        for k, d in defaults.items():
            assert libs[v].v2x_tuning_set(k.encode(), envs[v].get(k, d)) == 0
    _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    cases = [("conv6_1: (256 half-res + 128) -> 128 @64", 256, 128, 128, 64, 1, False), ("conv5_1: (512 half-res + 256) -> 256 @32", 512, 256, 256, 32, 1, False),
             ("conv3_2: 256 -> 256 @32", 256, 0, 256, 32, 0, False), ("conv6_2: 128 -> 128 @64", 128, 0, 128, 64, 0, False),
             ("ConvGRU 512 -> 3x256 @32", 256, 256, 256, 32, 0, True), ("conv7_1: (128 half-res + 64) -> 64 @128", 128, 64, 64, 128, 1, False), ("conv7_2-like: 64 -> 64 @128 (streamed)", 64, 0, 64, 128, 0, False),
             ("conv2_2 -> conv3d_2 chain: 128 -> 128 -> 128 @64", 128, 0, 128, 64, 0, "chain"),
             ("halo conv8_2: 32 -> 32 @256", 32, 0, 32, 256, 0, "halo"), ("halo conv7_2: 64 -> 64 @128", 64, 0, 64, 128, 0, "halo"),
             ("halo conv8_1: (64 half-res + 32) -> 32 @256", 64, 32, 32, 256, 1, "halo"),
             ("halo conv8_1 parity-class: (64 half-res + 32) -> 32 @256", 64, 32, 32, 256, 1, "ppc"),
             ("halo conv1_2 -> conv3d_1 chain: 64 -> 64 -> 64 @256... at 128", 64, 0, 64, 128, 0, "halochain"),
             ("halo heads: 32 -> 64 -> 12 | 36 fp32 @256", 32, 0, 64, 256, 0, "heads"),
             ("s2 conv2_1: 64 -> 128 @128 -> 64", 64, 0, 128, 128, 0, "s2"), ("s2 conv3_1: 128 -> 256 @64 -> 32", 128, 0, 256, 64, 0, "s2"),
             ("s2 conv4_1: 256 -> 512 @32 -> 16", 256, 0, 512, 32, 0, "s2"), ("pair conv_pre_1 -> conv_pre_2 from the bit grid @256", 32, 0, 32, 256, 0, "pair")]
    if only:
        cases = [c for c in cases if only in c[0]]
    n = 320
    for name, c0, c1, cout, hw, up, gru in cases:
        split = 0
        if gru == "pair":
            pa = packing.pack_conv_halo(name + ".a", torch.randn(32, 32, 3, 3, generator=g) * 0.05, torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1, relu=True, device=dev)
            pb = packing.pack_conv_halo(name + ".b", torch.randn(32, 32, 3, 3, generator=g) * 0.05, torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1, relu=True, device=dev)
            bits = (torch.rand(n // 8, hw, hw, generator=g) < 0.3).to(torch.int32) * (torch.randint(0, 1 << 13, (n // 8, hw, hw), generator=g, dtype=torch.int32))
            bits = bits.to(dev)
            outs = []
            for v in (0, 1):
                use(v)
                outs.append(ops.conv2d_pair(pa, pb, bits, 13).clone())
            same = torch.equal(outs[0], outs[1])
            reps = 40
            t = np.zeros((reps, 2))
            for r in range(-3, reps):
                for v in ((0, 1) if r % 2 == 0 else (1, 0)):
                    use(v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.conv2d_pair(pa, pb, bits, 13)
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 0:
                        t[r, v] = e0.elapsed_time(e1) * 1e3
            d = t[:, 1] - t[:, 0]
            print("%-44s A %.1f us  B %.1f us  B-A %+.1f us (%+.2f %%, s.e. %.2f %%)  outputs %s" % (
                name, t[:, 0].mean(), t[:, 1].mean(), d.mean(), 100 * d.mean() / t[:, 0].mean(), 100 * d.std() / np.sqrt(reps) / t[:, 0].mean(),
                "bit-identical" if same else "DIFFER"))
            continue
        if gru == "heads":
            from v2x_sim_amd._lib import V2X_EPI_F32
            w = torch.randn(64, 32, 3, 3, generator=g) * 0.05
            w2 = torch.zeros(48, 64, 1, 1)
            w2[:12, :32] = torch.randn(12, 32, 1, 1, generator=g) * 0.1
            w2[12:, 32:] = torch.randn(36, 32, 1, 1, generator=g) * 0.1
            pc = packing.pack_conv_halo(name, w, torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, relu=True,
                                        chain=(w2, torch.ones(48), torch.randn(48, generator=g) * 0.1, False), epilogue=V2X_EPI_F32, device=dev)
            split = 12
        elif gru == "halo":
            w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
            pc = packing.pack_conv_halo(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0 if c1 else c0 + c1, C1=c1, relu=True,
                                        device=dev)
        elif gru == "ppc":
            w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
            pc = packing.pack_conv_halo_parity(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0, C1=c1, relu=True, device=dev)
        elif gru == "halochain":
            w = torch.randn(cout, c0, 3, 3, generator=g) * 0.05
            ch = (torch.randn(cout, cout, 1, 1, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, True)
            pc = packing.pack_conv_halo(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, relu=True, chain=ch, device=dev)
        elif gru == "chain":
            w = torch.randn(cout, c0, 3, 3, generator=g) * 0.05
            ch = (torch.randn(cout, cout, 1, 1, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, True)
            pc = packing.pack_conv_stream(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0, relu=True, chain=ch, device=dev)
        elif gru == "s2":
            w = torch.randn(cout, c0, 3, 3, generator=g) * 0.05
            pc = packing.pack_conv_stream(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0, relu=True, stride=2, device=dev)
        elif gru:
            w = torch.randn(3 * cout, c0 + c1, 3, 3, generator=g) * 0.02
            pc = packing.pack_gru_stream(name, w, torch.randn(3 * cout, generator=g) * 0.1, torch.randn(3 * cout, generator=g) * 0.1, C0=c0, C1=c1, device=dev)
        else:
            w = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
            pc = packing.pack_conv_stream(name, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, C0=c0 if c1 else c0 + c1, C1=c1, up0=up,
                                          relu=True, device=dev)
        nn = n if hw <= 64 else (n // 2 if hw == 128 else n // 8)
        if c1:
            h0 = hw // 2 if up else hw
            # post-ReLU-like operands (half of the entries zero), as inside the network
            x0 = torch.relu(torch.randn(nn, h0, h0, c0, generator=g)).to(torch.bfloat16).to(dev)
            x1 = torch.relu(torch.randn(nn, hw, hw, c1, generator=g)).to(torch.bfloat16).to(dev)
        else:
            x0, x1 = torch.relu(torch.randn(nn, hw, hw, c0, generator=g)).to(torch.bfloat16).to(dev), None
        outs = []
        for v in (0, 1):
            use(v)
            outs.append((lambda r: torch.cat([t.reshape(-1) for t in r]) if isinstance(r, tuple) else r.clone())(ops.conv2d(pc, x0, x1, split=split)))
        same = torch.equal(outs[0], outs[1])
        reps = 40
        t = np.zeros((reps, 2))
        for r in range(-3, reps):
            for v in ((0, 1) if r % 2 == 0 else (1, 0)):
                use(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv2d(pc, x0, x1, split=split)
                e1.record()
                torch.cuda.synchronize()
                if r >= 0:
                    t[r, v] = e0.elapsed_time(e1) * 1e3
        d = t[:, 1] - t[:, 0]
        print("%-44s A %.1f us  B %.1f us  B-A %+.1f us (%+.2f %%, s.e. %.2f %%)  outputs %s" % (
            name, t[:, 0].mean(), t[:, 1].mean(), d.mean(), 100 * d.mean() / t[:, 0].mean(), 100 * d.std() / np.sqrt(reps) / t[:, 0].mean(),
            "bit-identical" if same else "DIFFER (max |d| %.3g of max |y| %.3g, %.2f %% of the entries)" % (
                float((outs[0].float() - outs[1].float()).abs().max()), float(outs[0].float().abs().max()), 100 * float((outs[0] != outs[1]).float().mean()))))


if __name__ == "__main__":
    main()
