#!/usr/bin/env python3
"""Which part of the chained launch costs what: the real movielens evaluation's queued pieces (producers, fused plate
step, final contraction) are intercepted at the moment they would be launched, and the chained entry point is replayed
with subsets of them.    python3 tools/chain_parts.py [K]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import alan_amd as alan
from alan_amd import native as N
import bench as B

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prob = B.build_problem("cuda")
sample = B.draw(prob, K)
strat = alan.no_checkpoint if K < 100 else alan.Split("plate_1", 38)
caught = []
real = N._launch_fused


def spy():
    caught.append(N._Q.fused)
    real()


N._launch_fused = spy
N.CHAIN_LAUNCHES = 2                                    # (every chained form: this tool measures their parts)
with t.no_grad():
    ref = float(sample.elbo_nograd(strat, graph=False))
N._launch_fused = real
f = caught[-1]
print(f"K={K}: prelude {len(f.prelude)} problems (modes {[d.mode for d, _, _ in f.prelude]}), tail {len(f.tail)}; elbo {ref:.4f}", flush=True)
L = N.lib()
state = t.zeros(4, dtype=t.int32, device="cuda")


def period(fn, n_rep=1000):
    s = t.cuda.Stream()
    with t.cuda.stream(s):
        for _ in range(3):
            fn()
        t.cuda.synchronize()
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g, stream=s):
            fn()
        for _ in range(20):
            g.replay()
        t.cuda.synchronize()
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n_rep):
            g.replay()
        b.record()
        t.cuda.synchronize()
        return a.elapsed_time(b) / n_rep * 1e3


def variant(pre, tail):
    PA, n, TA, m = f.arrays(pre, tail)
    rc = L.alan_normal_lse_chained_check(C.byref(f.desc), PA, n, TA, m)
    if rc < 0:
        return None

    def fn():
        N.check(L.alan_normal_lse_chained(C.byref(f.desc), PA, n, TA, m, state.data_ptr(), N.current_stream(t.device("cuda", 0))), "chained")
    return period(fn)


is_rec = lambda d: d.mode in (N.MODE_BERNOULLI_LINEAR,) or (d.mode in (N.MODE_NORMAL, N.MODE_NORMAL_LOGSCALE) and d.n_factors == 3)
rec = [it for it in f.prelude if is_rec(it[0])]
aux = [it for it in f.prelude if not is_rec(it[0])]
for name, pre, tail in (("body alone", [], []), ("body + tail", [], f.tail), ("body + every producer", f.prelude, []),
                        ("body + the [M,K] producers (in-tile at K <= 32)", rec, []), ("body + the other producers", aux, []),
                        ("everything", f.prelude, f.tail)):
    p = variant(pre, tail)
    print(f"  {name:55s} {'declined' if p is None else f'{p:7.2f} us'}   state {state.tolist()}", flush=True)
