#!/usr/bin/env python3
"""Kernel-level timing of the S-ML plate step (F[M,K,K,K] + g[M,K] -> lse Kz -> sum M) at literal and
scaled plate sizes.  GPU only.  Usage: python tools/microbench.py [K ...]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E


def time_graph(fn, iters):
    """Capture `iters` back-to-back launches in a HIP graph; return ms per launch."""
    fn()
    t.cuda.synchronize()
    g = t.cuda.CUDAGraph()
    s = t.cuda.Stream()
    with t.cuda.stream(s):
        fn()
        with t.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    t.cuda.synchronize()
    g.replay()
    t.cuda.synchronize()
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    t.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    Ks = [int(a) for a in sys.argv[1:]] or [10, 30, 100]
    dev = "cuda"
    for K in Ks:
        for scale in (1, 8, 64):
            M = 300 * scale
            nbytes = 4 * (M * K ** 3 + M * K)
            if nbytes > 6e9:
                continue
            g = t.Generator(device=dev).manual_seed(1234)
            F = -0.5 * t.randn(M, K, K, K, device=dev, generator=g) ** 2 - 0.9189 - math.log(K)
            gz = -0.5 * t.randn(M, K, device=dev, generator=g) ** 2 - 0.9189 - math.log(K)
            fac = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
            iters = 50 if nbytes < 5e8 else 10
            for name, fn in [
                ("lse+platesum", lambda: E.reduce_factors(fac, reduce=("z",), plate=("m",))),
                ("lse only", lambda: E.reduce_factors(fac, reduce=("z",))),
                ("lse F only", lambda: E.reduce_factors(fac[:1], reduce=("z",))),
            ]:
                ms = time_graph(fn, iters)
                print(f"K={K:4d} M={M:6d} {nbytes/1e6:9.1f} MB  {name:14s} {ms*1e3:9.1f} us  "
                      f"{nbytes/ms/1e9:8.1f} GB/s  ({nbytes/ms/1e9/8000*100:5.1f}% of 8 TB/s)", flush=True)
            del F, gz
            t.cuda.empty_cache()


if __name__ == "__main__":
    main()
