#!/usr/bin/env python3
"""One workload replayed n times for rocprofv3 (graph replays, so the trace holds only the evaluation's own kernels):
    python3 tools/prof_case.py {ml,bus,ts,vi,rws} [K] [n]
ml = movielens elbo_nograd (Split('plate_1', 38) at K >= 100, as bench.py), bus / ts = bench.py's other configurations,
vi / rws = a whole training iteration (GraphedStep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
from alan_amd.training import GraphedStep
which = sys.argv[1] if len(sys.argv) > 1 else "ml"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
if which in ("vi", "rws"):
    prob = bench.build_problem("cuda")
    # as bench.py's training_iteration leg: the library's own Adam (ALAN_PROF_TORCH_ADAM=1: torch's fused capturable one)
    if os.environ.get("ALAN_PROF_TORCH_ADAM") == "1":
        opt = (t.optim.Adam(prob.Q.parameters(), lr=1e-2, capturable=True, fused=True, maximize=True) if which == "rws"
               else t.optim.Adam(prob.parameters(), lr=1e-2, capturable=True, fused=True))
    else:
        opt = alan.Adam(prob.Q.parameters(), lr=1e-2, maximize=True) if which == "rws" else alan.Adam(prob.parameters(), lr=1e-2)
    step = GraphedStep(prob, K, opt, method=which, computation_strategy=bench.strategy_for(1, K))
    for _ in range(n):
        v = step()
else:
    prob = {"ml": bench.build_problem, "bus": bench.build_bus_problem, "ts": bench.build_timeseries_problem}[which]("cuda")
    s = bench.draw(prob, K)
    strat = bench.strategy_for(1, K) if which == "ml" else alan.no_checkpoint
    for _ in range(3):
        s.elbo_nograd(strat, graph=True)
    t.cuda.synchronize()
    for _ in range(n):
        v = s.elbo_nograd(strat, graph=True)
t.cuda.synchronize()
print(which, K, n, float(v))
