#!/usr/bin/env python3
"""Throughput of sample.EvalPipeline (alan_pipeline_*) by lanes and issuing threads, against one evaluation after another.
    python3 tools/pipeline_probe.py [K] [M] [evaluations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
import models

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
strategy = alan.no_checkpoint
ref = float(sample.elbo_nograd(strategy, graph=False))
for _ in range(10):
    sample.elbo_nograd(strategy, graph=True)
t.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    sample.elbo_nograd(strategy, graph=True)
t.cuda.synchronize()
print(f"K={K} M={M}: one after another (recorded launches, one stream): {(time.perf_counter() - t0) / n * 1e6:.2f} us per evaluation", flush=True)
for lanes, threads in ((1, 0), (1, 1), (2, 0), (2, 1), (2, 2), (3, 1), (3, 3), (4, 2), (4, 4), (6, 3), (6, 6), (8, 4)):
    pipe = alan.sample.EvalPipeline(sample, strategy, lanes=lanes, threads=threads, results=(n + lanes - 1) // lanes + 8)
    pipe.run(64)
    t.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        vals = pipe.run(n)
        t.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert float((vals - ref).abs().max()) <= 2e-6 * abs(ref)
    # a short batch: what the ramp costs (20 evaluations, as the driver's bench line times)
    t0 = time.perf_counter()
    pipe.run(20)
    t.cuda.synchronize()
    d20 = time.perf_counter() - t0
    print(f"K={K} M={M}: {lanes} lane(s), {threads} issuing thread(s): {best / n * 1e6:.2f} us per evaluation ({n / best:.0f} evals/s); "
          f"a batch of 20: {d20 / 20 * 1e6:.2f} us per evaluation", flush=True)
    pipe.close()
