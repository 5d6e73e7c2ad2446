#!/usr/bin/env python3
"""Where the host's microseconds of one replayed evaluation go (movielens K=30, sample.elbo_nograd(graph=True)):
   python3 tools/host_path_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch as t
import alan_amd as alan
import models

g = t.Generator().manual_seed(5)
xx = t.randn(300, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(300, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": 300, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(30, reparam=False)
for _ in range(5):
    sample.elbo_nograd(graph=True)
ge = next(iter(sample.__dict__["_graphs"].values()))
N = 3000


def clock(label, fn, sync_every=200):
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        fn()
        if (i + 1) % sync_every == 0:
            t.cuda.synchronize()                    # (keeps the queue from filling: host cost, not back-pressure)
    dt = (time.perf_counter() - t0) / N * 1e6
    t.cuda.synchronize()
    print(f"{label:58s} {dt:6.2f} us per call", flush=True)


strat = alan.checkpoint
clock("sample.elbo_nograd(graph=True)  [everything]", lambda: sample.elbo_nograd(graph=True))
side = t.cuda.Stream()
side.wait_stream(t.cuda.current_stream())
with t.cuda.stream(side):                           # (not the legacy default stream)
    clock("  the same on a stream of its own", lambda: sample.elbo_nograd(graph=True))
t.cuda.current_stream().wait_stream(side)
clock("  _graph_key(strategy)", lambda: sample._graph_key(strat))
clock("  _graphed(strategy)  [key + dict]", lambda: sample._graphed(strat))
clock("  ring.claim()", lambda: ge.ring.claim())
clock("  torch.cuda.current_stream().cuda_stream", lambda: t.cuda.current_stream().cuda_stream)
st = t.cuda.current_stream().cuda_stream
if ge.calls is not None:
    clock("  calls.replay(stream)  [3 launches from C]", lambda: ge.calls.replay(st), sync_every=50)
clock("  graph.replay()", lambda: ge.graph.replay(), sync_every=50)
slot = ge.ring.claim()
clock("  slot.detach()", lambda: slot.detach())
