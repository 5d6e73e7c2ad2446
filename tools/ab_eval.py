#!/usr/bin/env python3
"""A/B of routing switches on the movielens evaluation (graph replay, same process, same box):
    python3 tools/ab_eval.py [K] [switch=value ...]      e.g.  tools/ab_eval.py 30 logpq.PARTIAL_PLATE_SUMS=0
Prints microseconds per evaluation with the defaults and with the switches applied, alternating three times."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import alan_amd as alan
import bench as B

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
switches = []
for a in sys.argv[2:]:
    name, val = a.split("=")
    mod, attr = name.rsplit(".", 1)
    switches.append((importlib.import_module("alan_amd." + mod), attr, eval(val)))
strat = alan.no_checkpoint if K < 100 else alan.Split("plate_1", 38)


def run(apply):
    saved = [(m, a, getattr(m, a)) for m, a, _ in switches]
    if apply:
        for m, a, v in switches:
            setattr(m, a, v)
    try:
        prob = B.build_problem("cuda")
        sample = B.draw(prob, K)
        dt, val = B.timed_evals(sample, strat, 200, 20, 1, graph=True)
        return dt / 200 * 1e6, val
    finally:
        for m, a, v in saved:
            setattr(m, a, v)


for rep in range(3):
    a = run(False)
    b = run(True)
    print(f"K={K} rep {rep}: defaults {a[0]:.2f} us/eval (elbo {a[1]:.4f})   with {sys.argv[2:]}: {b[0]:.2f} us/eval (elbo {b[1]:.4f})", flush=True)
