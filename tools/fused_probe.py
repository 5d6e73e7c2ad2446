#!/usr/bin/env python3
"""Graph-replay time of the movielens ELBO with and without the fused plate step (dist.FUSE_PLATE_STEP)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
from alan_amd import dist as D
prob = bench.build_problem("cuda")
for K in (30, 100, 10):
    strat = bench.strategy_for(1, K)
    for fused in (False, True):
        D.FUSE_PLATE_STEP = fused
        s = bench.draw(prob, K)
        for _ in range(3): v = s.elbo_nograd(strat, graph=True)
        t.cuda.synchronize(); t0 = time.perf_counter()
        n = 200 if K < 100 else 30
        for _ in range(n): v = s.elbo_nograd(strat, graph=True)
        t.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"K={K:3d} fused={fused!s:5s}: {dt*1e6:8.1f} us/eval  {1/dt:9.1f} evals/s  elbo {float(v):.4f}", flush=True)
