#!/usr/bin/env python3
"""Is the idle time between two replays of ONE captured graph (8-9 us on the GPU's timeline, tools/replay_trace.sh) tied to
the graph's executable object?  The same evaluation captured into 1, 2 and 4 graphs, replayed in turn; and the same
launches issued again kernel by kernel from the host for comparison.
    python3 tools/replay_alternate_probe.py [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
from alan_amd import native as N

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def capture(fn, s):
    g = t.cuda.CUDAGraph()
    with t.cuda.graph(g, stream=s):
        keep = fn()
    return g, keep


def run(fn, n_graphs, label):
    s = t.cuda.Stream()
    with t.cuda.stream(s):
        for _ in range(3):
            fn()
        t.cuda.synchronize()
        gs = [capture(fn, s) for _ in range(n_graphs)]
        for i in range(20):
            gs[i % n_graphs][0].replay()
        t.cuda.synchronize()
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for i in range(n_rep):
            gs[i % n_graphs][0].replay()
        b.record()
        t1 = time.perf_counter()
        t.cuda.synchronize()
    print(f"{label}: {n_graphs} graph(s) in turn: period {a.elapsed_time(b) / n_rep * 1e3:.2f} us (host {(t1 - t0) / n_rep * 1e6:.2f} us per replay)", flush=True)


x = t.zeros(64, device="cuda")


def trivial3():
    for _ in range(3):
        x.add_(1.0)


for n in (1, 2, 4):
    run(trivial3, n, "3 trivial dependent kernels")

import models
g = t.Generator().manual_seed(5)
xx = t.randn(300, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(300, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": 300, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(30, reparam=False)
with t.no_grad():
    ev = lambda: sample.elbo_nograd(graph=False)
    for n in (1, 2, 4):
        run(ev, n, "movielens K=30 evaluation (3 launches)")
