#!/usr/bin/env python3
"""What the producers' launch of a movielens evaluation (one reduce_small_multi_kernel launch: four problems) is made of:
each of its problems launched ALONE 200 times, then the batch 200 times -- for `rocprofv3 --kernel-trace --stats`, whose
per-kernel averages then tell the problems apart (alone they run as reduce_small_kernel<mode> / bernoulli_linear_kernel).
    python3 tools/producer_parts_probe.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch as t
import alan_amd as alan, bench
from alan_amd import native as N

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prob = bench.build_problem("cuda")
s = bench.draw(prob, K)
caught = []
real = N._flush_items


def spy(items):
    caught.append(list(items))
    return real(items)


N._flush_items = spy
with t.no_grad():
    s.elbo_nograd(alan.no_checkpoint, graph=False)
N._flush_items = real
items = max(caught, key=len)
L = N.lib()
stream = N.current_stream(t.device("cuda", 0))
print("problems of the batch:", [N._desc_info(d) for d, _, _ in items], flush=True)
for d, _, _ in items:
    for _ in range(200):
        N.check(L.alan_reduce(C.byref(d), None, 0, stream), "alan_reduce")
    t.cuda.synchronize()
arr = (C.POINTER(N.ReduceDesc) * len(items))(*[C.pointer(d) for d, _, _ in items])
for _ in range(200):
    N.check(L.alan_reduce_batch(arr, len(items), stream), "alan_reduce_batch")
t.cuda.synchronize()
print("done")
