#!/usr/bin/env python3
"""Where a SHORT pipelined batch spends its time (bench.py's headline times 20 evaluations between two synchronisations;
the steady state is 10 us per evaluation, a batch of 20 takes 17): host seconds of each phase of EvalPipeline.run(n).
    python3 tools/pipeline_batch_probe.py [n] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
from alan_amd import native as N
import models

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K, M = 30, 300
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
pipe = sample.pipeline(alan.no_checkpoint, lanes=lanes)
pipe.run(64)
t.cuda.synchronize()
L = N.lib()
rows = []
for rep in range(12):
    t.cuda.synchronize()
    p = [time.perf_counter()]
    ok = pipe.problem.memory_fingerprint() == pipe.fingerprint
    p.append(time.perf_counter())
    pipe._sync_noise(n)
    stream = t.cuda.current_stream().cuda_stream
    N.check(L.alan_pipeline_fence(pipe._h, stream), "fence")
    p.append(time.perf_counter())
    N.check(L.alan_pipeline_submit(pipe._h, n), "submit")
    pipe.total += n
    p.append(time.perf_counter())
    N.check(L.alan_pipeline_join(pipe._h, stream), "join")
    p.append(time.perf_counter())
    t.cuda.synchronize()
    p.append(time.perf_counter())
    pipe.first = pipe.total                 # (what results() does to the window)
    rows.append([(b - a) * 1e6 for a, b in zip(p, p[1:])] + [(p[-1] - p[0]) * 1e6])
    # the same through the public call
    t.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(n)
    t.cuda.synchronize()
    rows[-1].append((time.perf_counter() - t0) * 1e6)
import statistics
names = ["fingerprint", "noise + fence", "submit", "join (waits until every launch is issued)", "synchronize", "total", "pipe.run(n) + synchronize"]
for i, nm in enumerate(names):
    col = sorted(r[i] for r in rows[2:])
    print(f"{nm:48s} median {statistics.median(col):7.1f} us   min {col[0]:7.1f}")
print(f"n = {n}, lanes = {lanes}: steady state would be {n} x ~10 us")
