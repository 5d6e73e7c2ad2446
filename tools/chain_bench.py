#!/usr/bin/env python3
"""chain_logmmexp (+ logsumexp) alone:  python3 tools/chain_bench.py [T] [K ...]  -- per-evaluation time by HIP events
over graph replays, and a sanity check against an fp64 left-to-right log-matvec scan on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import native as N
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Ks = [int(k) for k in sys.argv[2:]] or [30, 64, 100]
for K in Ks:
    g = t.Generator().manual_seed(K)
    ms = (-0.5 * t.randn(T, K, K, generator=g) ** 2 - 0.92 - t.log(t.tensor(float(K)))).cuda()
    vec, _, _ = N.chain_logmmexp(ms)
    u = t.logsumexp(ms[-1].double(), -1)                        # right-to-left scan: u_t[i] = LSE_j(M_t[i,j] + u_{t+1}[j])
    for step in range(T - 2, -1, -1):
        u = t.logsumexp(ms[step].double() + u[None, :], -1)
    err = float((vec.double() - u).abs().max())
    for _ in range(3):
        N.chain_logmmexp(ms)
    t.cuda.synchronize()
    gr = t.cuda.CUDAGraph()
    with t.cuda.graph(gr):
        N.chain_logmmexp(ms)
    for _ in range(3):
        gr.replay()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        gr.replay()
    b.record()
    t.cuda.synchronize()
    print(f"T={T} K={K}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per chain (graph replay), max |err| vs fp64 scan {err:.2e}")
