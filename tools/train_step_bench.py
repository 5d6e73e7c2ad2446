#!/usr/bin/env python3
"""Times one VI / RWS training iteration of the movielens model (sample -> elbo -> backward), the loop of
examples/basic_runner.py:81-112 of the reference, on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import bench
import alan_amd as alan

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prob = bench.build_problem("cuda")
opt = t.optim.Adam(prob.parameters(), lr=1e-3)
for mode in (() if os.environ.get("SKIP_EAGER") else ("rws", "vi")):
    for strat_name, strat in (("no_checkpoint", alan.no_checkpoint), ("checkpoint", alan.checkpoint)):
        def step():
            opt.zero_grad()
            s = prob.sample(K, reparam=(mode == "vi"))
            e = s.elbo_vi(strat) if mode == "vi" else s.elbo_rws(strat)
            (-e).backward()
            opt.step()
            return e
        for _ in range(3):
            step()
        t.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            e = step()
        t.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"K={K} {mode:3s} {strat_name:13s}: {dt*1e3:8.2f} ms/iter  ({1/dt:7.1f} it/s)  elbo {float(e):.2f}  "
              f"peak mem {t.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
        t.cuda.reset_peak_memory_stats()

# ---- the same iteration as one HIP-graph replay
from alan_amd.training import GraphedStep
for mode in (os.environ.get("GRAPH_MODES", "rws,vi").split(",")):
    prob = bench.build_problem("cuda")
    # as examples/basic_runner.py:76-79 of the reference: RWS updates Q with maximize=True on (-elbo)
    opt = (t.optim.Adam(prob.Q.parameters(), lr=1e-2, capturable=True, maximize=True) if mode == "rws"
           else t.optim.Adam(prob.parameters(), lr=1e-2, capturable=True))
    step = GraphedStep(prob, K, opt, method=mode)
    e0 = sum(float(step()) for _ in range(20)) / 20
    t.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        e = step()
    t.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    e1 = sum(float(step()) for _ in range(20)) / 20
    print(f"K={K} {mode:3s} graph replay  : {dt*1e3:8.3f} ms/iter  ({1/dt:7.1f} it/s)  mean elbo first 20: {e0:.1f} -> "
          f"after {n + 40} iters: {e1:.1f}", flush=True)
