#!/usr/bin/env python3
"""tools/pipeline_probe.py for a few (lanes, threads) pairs only: for sweeps over the library's launch-geometry knobs.
    python3 tools/pipeline_probe_short.py [K] [M] [evaluations] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
import models

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
tag = sys.argv[4] if len(sys.argv) > 4 else ""
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
strategy = alan.no_checkpoint if K < 100 or M < 100 else alan.Split("plate_1", 38)
ref = float(sample.elbo_nograd(strategy, graph=False))
out = []
for lanes, threads in ((1, 0), (3, 3), (4, 4), (6, 6)):
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
    out.append(f"{lanes}x{threads}: {best / n * 1e6:.2f}")
    pipe.close()
print(f"{tag} K={K} M={M} us per evaluation by lanes x threads: " + "  ".join(out), flush=True)
