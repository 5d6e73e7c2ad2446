#!/usr/bin/env python3
"""The fused plate step through its plain launch and through the chained kernel with nothing chained (same tiles, plus
the arrival counter): replay periods, to tell the chained kernel's own overhead from what the prelude / tail cost.
    python3 tools/chain_probe.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E, native as N
from alan_amd.dims import Dim


def period(fn, n_rep=1000):
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
        a.record()
        for _ in range(n_rep):
            g.replay()
        b.record()
        t.cuda.synchronize()
        return a.elapsed_time(b) / n_rep * 1e3


L = N.lib()
for M, K in ((300, 30), (300, 100), (38, 100)):
    g = t.Generator(device="cuda").manual_seed(0)
    pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
    z = t.randn(M, K, 18, device="cuda", generator=g)
    mu = t.randn(K, 18, device="cuda", generator=g)
    raw = 0.3 * t.randn(K, 18, device="cuda", generator=g)
    sm = [(t.randn(M, K, device="cuda", generator=g), (pl, Kz)) for _ in range(2)]
    a = E._normal_lse_args((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz)
    d = E._normal_lse_desc(a, True)
    d.out = d.value
    n_parts = int(L.alan_normal_lse_n_partials(C.byref(d)))
    out1 = t.empty(n_parts, K, K, device="cuda")
    out2 = t.empty(n_parts, K, K, device="cuda")
    state = t.zeros(4, dtype=t.int32, device="cuda")
    d.o_sl, d.o_ss, d.add_const, d.keep_partials = K, 1, 0.0, 1

    def plain():
        d.out = out1.data_ptr()
        N.check(L.alan_normal_lse(C.byref(d), None, 0, N.current_stream(z.device)), "plain")

    def chained():
        d.out = out2.data_ptr()
        N.check(L.alan_normal_lse_chained(C.byref(d), None, 0, None, 0, state.data_ptr(), N.current_stream(z.device)), "chained")

    p1, p2 = period(plain), period(chained)
    p1b, p2b = period(plain), period(chained)
    print(f"M={M} K={K}: plain {p1:.2f} / {p1b:.2f} us   chained kernel, nothing chained {p2:.2f} / {p2b:.2f} us   "
          f"max |diff| {float((out1 - out2).abs().max()):.3g}  state {state.tolist()}", flush=True)
