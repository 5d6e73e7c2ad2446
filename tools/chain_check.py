#!/usr/bin/env python3
"""The chained launch (alan_normal_lse_chained: producers + fused plate step + final contraction in one launch) against
the separate launches, on movielens: eager values at K = 3 / 10 / 30 / 100, then tools/ab_eval-style replay periods.
    python3 tools/chain_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import alan_amd as alan
from alan_amd import native as N
import bench as B

for K in (3, 10, 30, 100):
    prob = B.build_problem("cuda")
    sample = B.draw(prob, K)
    strat = alan.no_checkpoint if K < 100 else alan.Split("plate_1", 38)
    vals = {}
    for chain in (0, 2):
        N.CHAIN_LAUNCHES = chain
        with t.no_grad():
            vals[chain] = [float(sample.elbo_nograd(strat, graph=False)) for _ in range(3)]
        t.cuda.synchronize()
    st = [x.tolist() for x in N._CHAIN_STATE.values()]
    print(f"K={K}: separate {vals[0]}  chained {vals[2]}  state {st}", flush=True)
MODES = {"separate": (0, False), "sync-free": (1, False), "prelude": (2, False), "full": (2, True)}
for K in (30, 100):
    strat = alan.no_checkpoint if K < 100 else alan.Split("plate_1", 38)
    for rep in range(2):
        res = []
        for name, (chain, tail) in MODES.items():
            N.CHAIN_LAUNCHES, N.CHAIN_TAIL = chain, tail
            prob = B.build_problem("cuda")
            sample = B.draw(prob, K)
            dt, val = B.timed_evals(sample, strat, 300, 20, 1, graph=True)
            res.append(f"{name} {dt / 300 * 1e6:.2f} us ({val:.4f})")
        print(f"K={K} rep {rep}: " + "   ".join(res), flush=True)
