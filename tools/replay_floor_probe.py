#!/usr/bin/env python3
"""What a replayed evaluation costs as a function of its launch count (the question behind folding the small producers
and the top-level log-sum-exp into the fused plate step's launch):
    python3 tools/replay_floor_probe.py [replays]
Period of back-to-back replays of graphs holding (a) n dependent trivial kernels, n = 1 .. 4, (b) the fused plate step
alone at K = 30 / 100, (c) the fused plate step between n trivial kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def period(fn):
    s = t.cuda.Stream()
    with t.cuda.stream(s):
        for _ in range(3):
            fn()
        t.cuda.synchronize()
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g, stream=s):
            fn()
        for _ in range(20):
            g.replay()
        t.cuda.synchronize()
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(n_rep):
            g.replay()
        b.record()
        t1 = time.perf_counter()
        t.cuda.synchronize()
        return a.elapsed_time(b) / n_rep * 1e3, (t1 - t0) / n_rep * 1e6


x = t.zeros(64, device="cuda")


def trivial(n):
    def fn():
        for _ in range(n):
            x.add_(1.0)
    return fn


for n in (1, 2, 3, 4, 8):
    dev, host = period(trivial(n))
    print(f"{n} trivial dependent kernels: period {dev:.2f} us (host submit {host:.2f} us per replay)", flush=True)

for M, K in ((300, 30), (300, 100), (38, 100)):
    g = t.Generator(device="cuda").manual_seed(0)
    pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
    z = t.randn(M, K, 18, device="cuda", generator=g)
    mu = t.randn(K, 18, device="cuda", generator=g)
    raw = 0.3 * t.randn(K, 18, device="cuda", generator=g)
    sm = [(t.randn(M, K, device="cuda", generator=g), (pl, Kz)) for _ in range(2)]
    nl = lambda: E.normal_lse((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz, log_scale=True, partials=True)
    for pre, post in ((0, 0), (1, 0), (1, 1), (1, 2)):
        def fn():
            for _ in range(pre):
                x.add_(1.0)
            nl()
            for _ in range(post):
                x.add_(1.0)
        dev, host = period(fn)
        print(f"M={M} K={K}: {pre} trivial + fused plate step + {post} trivial: period {dev:.2f} us (host {host:.2f})", flush=True)
