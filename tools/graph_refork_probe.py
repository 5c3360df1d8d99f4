#!/usr/bin/env python3
"""Does a stream capture honour a SECOND fork of the same side stream?  main: k1 -> [fork] -> long work -> k3 -> [fork again] -> ... -> join.
The side stream's second piece of work reads what k3 wrote: if the replayed graph drops the second dependency it reads stale data.
usage: python tools/graph_refork_probe.py   (prints, per variant, how many of 50 replays produced the right value)"""
import torch

dev = torch.device("cuda:0")
N = 1 << 20


def build(refork_same_stream, long_side=0):
    a = torch.zeros(N, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    c = torch.zeros(N, device=dev)
    b = torch.zeros(N, device=dev)
    d = torch.zeros(N, device=dev)
    e = torch.zeros(N, device=dev)
    side1, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    s2 = side1 if refork_same_stream else side2
    g = torch.cuda.CUDAGraph()
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        (big @ big).sum()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        a.add_(1.0)                                   # k1
        side1.wait_stream(main)
        with torch.cuda.stream(side1):
            u = big
            for _ in range(long_side):                # side piece 1: LONG when long_side > 0 (is it still an ancestor of the join?)
                u = (u @ big) * 1e-3
            b.copy_(a * 2.0 + u.flatten()[:N] * 0.0)  # reads a
        t = big
        for _ in range(6 if not long_side else 1):    # ~ms of work on main
            t = (t @ big) * 1e-3
        c.copy_(a + t.flatten()[:N] * 0.0 + 5.0)      # k3: c = a + 5, after the long chain
        s2.wait_stream(main)                          # second fork
        with torch.cuda.stream(s2):
            d.copy_(c * 3.0)                          # side piece 2: reads c
        main.wait_stream(side1)
        if s2 is not side1:
            main.wait_stream(s2)
        e.copy_(b + d)                                # after the join: needs BOTH side pieces
    return g, a, b, c, d, e


for same, long_side in ((True, 0), (False, 0), (True, 12), (False, 12)):
    g, a, b, c, d, e = build(same, long_side)
    ok = 0
    for it in range(50):
        b.zero_()
        c.zero_()
        d.zero_()
        g.replay()
        torch.cuda.synchronize()
        want_b, want_d = (it + 1) * 2.0, (it + 1 + 5.0) * 3.0      # the capture itself runs nothing: a = it + 1 in replay `it`
        good = bool((b == want_b).all()) and bool((d == want_d).all()) and bool((e == want_b + want_d).all())
        ok += good
    print("second fork on %s, first side piece %s: %d / 50 replays correct (last e[0] = %.1f, want %.1f)"
          % ("the SAME side stream" if same else "a second side stream", "LONG" if long_side else "short", ok, float(e[0]), want_b + want_d))
