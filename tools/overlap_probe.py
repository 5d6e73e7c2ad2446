#!/usr/bin/env python3
"""Do INDEPENDENT evaluations overlap on the chip when their recorded launch lists are issued on different streams?
n copies of the same movielens evaluation (each captured on its own: its own intermediates, result ring and launch
list) are re-issued round-robin on n streams; the period per evaluation is the wall time of the whole run over the
number of evaluations (basic_runner.py:81-112 evaluates independent ELBOs one after another: a throughput).
    python3 tools/overlap_probe.py [K] [M] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
from alan_amd import sample as S
import models

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
strategy = alan.no_checkpoint
with t.no_grad():
    ref = float(sample.elbo_nograd(strategy, graph=False))
MAXN = 4
evs = [S._GraphedELBO(sample, strategy) for _ in range(MAXN)]
assert all(e.calls is not None for e in evs), "the evaluation did not record as library launches alone"
streams = [t.cuda.Stream() for _ in range(MAXN)]
t.cuda.synchronize()
for n in (1, 2, 3, 4):
    for i in range(40):
        evs[i % n].calls.replay(streams[i % n].cuda_stream)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_rep):
        evs[i % n].calls.replay(streams[i % n].cuda_stream)
    t1 = time.perf_counter()
    t.cuda.synchronize()
    t2 = time.perf_counter()
    vals = []
    for e in evs[:n]:
        if e.ring is None:
            vals.append(float(e.out))
        else:
            e.ring.sync_position()
            vals.append(float(e.ring.slots[(e.ring.pos - 1) % e.ring.n]))
    assert all(abs(v - ref) <= 1e-5 * abs(ref) for v in vals), (vals, ref)
    print(f"K={K} M={M}: {n} stream(s): {(t2 - t0) / n_rep * 1e6:.2f} us per evaluation ({n_rep / (t2 - t0):.0f} evals/s; host issue "
          f"{(t1 - t0) / n_rep * 1e6:.2f} us)  reference {ref:.4f}", flush=True)

# the same with each copy's captured GRAPH launched on its stream (one host call per evaluation instead of three)
for n in (1, 2, 3, 4):
    def go(i):
        with t.cuda.stream(streams[i % n]):
            evs[i % n].graph.replay()
    for i in range(40):
        go(i)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_rep):
        go(i)
    t1 = time.perf_counter()
    t.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K} M={M}: graphs on {n} stream(s): {(t2 - t0) / n_rep * 1e6:.2f} us per evaluation (host issue {(t1 - t0) / n_rep * 1e6:.2f} us)", flush=True)
