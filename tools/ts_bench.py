"""Timeseries config (C5: Kalman T=1000, K=30): evals/s eager and as a replayed HIP graph, forward and a whole
elbo_rws backward.  Usage: python tools/ts_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as t
import alan_amd as alan
import bench

for T, K in ((1000, 30), (1000, 100), (100, 30)):
    prob = bench.build_timeseries_problem("cuda", T=T)
    t.manual_seed(0)
    s = prob.sample(K, reparam=False)
    for graph in (False, True):
        for _ in range(3):
            v = s.elbo_nograd(alan.no_checkpoint, graph=graph)
        t.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            v = s.elbo_nograd(alan.no_checkpoint, graph=graph)
        t.cuda.synchronize()
        d = (time.perf_counter() - t0) / n
        print(f"T={T} K={K} graph={graph}: {d * 1e6:8.1f} us/eval  elbo {float(v):.4f}", flush=True)
    if K <= 30:
        from alan_amd import OptParam
        # gradient of the chain alone on this model's factor shape
        from alan_amd.contract import chain_logmmexp_lse
        ms = (-0.5 * t.randn(T, K, K, device="cuda") ** 2 - 3.0).requires_grad_(True)
        g = t.rand(K, device="cuda")
        for _ in range(3):
            t.autograd.grad(chain_logmmexp_lse(ms), ms, g)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            t.autograd.grad(chain_logmmexp_lse(ms), ms, g)
        t.cuda.synchronize()
        print(f"T={T} K={K} chain forward + backward (eager): {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us", flush=True)
