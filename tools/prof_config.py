#!/usr/bin/env python3
"""Graph-replays one of bench.py's other configurations for rocprofv3:  python3 tools/prof_config.py {bus,ts} [n] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
which = sys.argv[1] if len(sys.argv) > 1 else "ts"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prob = (bench.build_timeseries_problem if which == "ts" else bench.build_bus_problem)("cuda")
s = bench.draw(prob, int(sys.argv[3]) if len(sys.argv) > 3 else 30)
for _ in range(3):
    s.elbo_nograd(alan.no_checkpoint, graph=True)
t.cuda.synchronize()
for _ in range(n):
    v = s.elbo_nograd(alan.no_checkpoint, graph=True)
t.cuda.synchronize()
print(which, float(v))
