#!/usr/bin/env python3
"""Replayed-graph ELBO == eager ELBO for every bench configuration, with and without host synchronisation between
replays (a replay that starts on an idle GPU exposes missing dependencies inside a captured graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
cases = [("movielens K=30", bench.build_problem, 30, alan.no_checkpoint),
         ("movielens K=100 Split", bench.build_problem, 100, alan.Split("plate_1", 38)),
         ("bus_breakdown K=30", bench.build_bus_problem, 30, alan.no_checkpoint),
         ("timeseries K=30", bench.build_timeseries_problem, 30, alan.no_checkpoint)]
ok = True
for name, build, K, strat in cases:
    prob = build("cuda")
    s = bench.draw(prob, K)
    eager = float(s.elbo_nograd(strat))
    vals_sync, vals_async = [], []
    for _ in range(30):
        vals_sync.append(float(s.elbo_nograd(strat, graph=True)))        # float() synchronises
    outs = [s.elbo_nograd(strat, graph=True) for _ in range(30)]          # back to back
    t.cuda.synchronize()
    vals_async = [float(o) for o in outs]
    rel = lambda v: abs(v - eager) / abs(eager)
    worst = max(max(map(rel, vals_sync)), max(map(rel, vals_async)))
    same = len(set(vals_sync)) == 1 and len(set(vals_async)) == 1 and vals_sync[0] == vals_async[0]
    ok = ok and same and worst < 1e-6
    print(f"{name:24s} eager {eager:.4f}  graph {vals_sync[0]:.4f}  identical across replays/sync modes: {same}  "
          f"max rel diff vs eager {worst:.1e}")
print("OK" if ok else "MISMATCH")
