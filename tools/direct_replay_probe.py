#!/usr/bin/env python3
"""Would re-issuing an evaluation's library calls from the host (no HIP graph) beat the graph replay, whose launch leaves
the GPU idle for several microseconds?  One eager evaluation's calls into libalan_mi355.so are recorded (function + the
very argument objects) and issued again in a tight loop; the period is compared with the graph's.
    python3 tools/direct_replay_probe.py [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
from alan_amd import native as N
import models

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = t.Generator().manual_seed(5)
xx = t.randn(300, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(300, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": 300, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(30, reparam=False)
L = N.lib()
names = ("alan_reduce", "alan_reduce_batch", "alan_normal_lse")
with t.no_grad():
    for _ in range(3):
        ref = float(sample.elbo_nograd(graph=False))
    calls, keep = [], []

    class Rec:
        def __init__(self, name, fn):
            self.name, self.fn = name, fn

        def __call__(self, *a):
            calls.append((self.fn, a))
            return self.fn(*a)
    real = {n: getattr(L, n) for n in names}
    for n in names:
        setattr(L, n, Rec(n, real[n]))
    out = sample.elbo_nograd(graph=False)
    keep.append(out)
    for n in names:
        setattr(L, n, real[n])
    t.cuda.synchronize()
    print("recorded calls:", [f.__name__ for f, _ in calls], "value", float(out), "reference", ref)
    for fn, a in calls:
        fn(*a)
    t.cuda.synchronize()
    print("re-issued value", float(out))
    a_, b_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a_.record()
    for _ in range(n_rep):
        for fn, a in calls:
            fn(*a)
    b_.record()
    t1 = time.perf_counter()
    t.cuda.synchronize()
    print(f"host re-issue of the {len(calls)} library calls: period {a_.elapsed_time(b_) / n_rep * 1e3:.2f} us (host {(t1 - t0) / n_rep * 1e6:.2f} us per evaluation)")
    for _ in range(5):
        sample.elbo_nograd(graph=True)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    a_.record()
    for _ in range(n_rep):
        sample.elbo_nograd(graph=True)
    b_.record()
    t1 = time.perf_counter()
    t.cuda.synchronize()
    print(f"graph replay: period {a_.elapsed_time(b_) / n_rep * 1e3:.2f} us (host {(t1 - t0) / n_rep * 1e6:.2f} us per evaluation)")
