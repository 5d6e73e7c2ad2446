"""Reductions inside the outer-product producer's backward: torch's (two-level, to stay on single-block kernels inside
replayed graphs: dist._sum_leading) against alan_reduce(mode=SUM).  Usage: python tools/sum_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as t
from alan_amd import engine as E
from alan_amd.dist import _sum_leading


def bench(name, fn):
    g = t.cuda.CUDAGraph()
    s = t.cuda.Stream()
    s.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        with t.cuda.graph(g, stream=s):
            for _ in range(10):
                out = fn()
    t.cuda.synchronize()
    g.replay()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    t.cuda.synchronize()
    print(f"{name:64s} {(time.perf_counter() - t0) / 300 * 1e6:7.2f} us", flush=True)
    return out


G = t.randn(300, 30, 30, 30, device="cuda")                       # [m, Kmu, Kpsi, Kz]
T = t.randn(9000, 30, 18, device="cuda")                          # [value row, loc row, event]
Gp = G.permute(0, 3, 1, 2).reshape(270000, 30)
with t.no_grad():
    a = bench("S0: _sum_leading(Gp)                     [270000,30] -> [30]", lambda: _sum_leading(Gp))
    b = bench("S0: alan SUM over (m, Kmu, Kz) of G", lambda: E.reduce_factors([(G, ("m", "a", "b", "z"))], plate=("m", "a", "z"))[0])
    print("   max diff", (a - b).abs().max().item())
    c = bench("gl: _sum_leading(T.view(9000, 540))      -> [540]", lambda: _sum_leading(T.view(9000, 540)))
    d = bench("gl: alan SUM over v of T", lambda: E.reduce_factors([(T, ("v", "l", "e"))], plate=("v",))[0])
    print("   max diff", (c - d.reshape(-1)).abs().max().item())
    e = bench("gv: T.sum(1)                             -> [9000,18]", lambda: T.sum(1))
    f = bench("gv: alan SUM over l of T", lambda: E.reduce_factors([(T, ("v", "l", "e"))], plate=("l",))[0])
    print("   max diff", (e - f).abs().max().item())
