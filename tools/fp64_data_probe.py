#!/usr/bin/env python3
"""movielens K=30 with fp64 observations (what a real data file gives): graph-replay time of the ELBO."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
import alan_amd
prob32 = bench.build_problem("cuda")
# same model, data cast to fp64
from alan_amd import Problem
data64 = {k: getattr(prob32._data, f"t_{k}").double().refine_names(*prob32._data._names[k]) for k in prob32._data._keys}
prob64 = Problem(prob32.P, prob32.Q, data64).to("cuda")
for name, prob in (("fp32 obs", prob32), ("fp64 obs", prob64)):
    s = bench.draw(prob, 30)
    for _ in range(3): v = s.elbo_nograd(alan.no_checkpoint, graph=True)
    t.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): v = s.elbo_nograd(alan.no_checkpoint, graph=True)
    t.cuda.synchronize()
    print(name, f"{(time.perf_counter()-t0)/200*1e6:.1f} us/eval", v.dtype, float(v))
