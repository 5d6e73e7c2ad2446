#!/usr/bin/env python3
"""One EvalPipeline run for profilers: python3 tools/pipeline_run.py K M lanes threads n"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
import models

K, M, lanes, threads, n = (int(x) for x in sys.argv[1:6])
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
strategy = alan.no_checkpoint if K < 100 or M < 100 else alan.Split("plate_1", 38)
ref = float(sample.elbo_nograd(strategy, graph=False))
pipe = alan.sample.EvalPipeline(sample, strategy, lanes=lanes, threads=threads, results=(n + lanes - 1) // lanes + 8)
pipe.run(64)
t.cuda.synchronize()
t0 = time.perf_counter()
vals = pipe.run(n)
t.cuda.synchronize()
dt = time.perf_counter() - t0
assert float((vals - ref).abs().max()) <= 2e-6 * abs(ref)
print(f"K={K} M={M} {lanes} lanes x {threads} threads: {dt / n * 1e6:.2f} us per evaluation", flush=True)
pipe.close()
