#!/usr/bin/env python3
"""What ONE rank of an N-GPU C4 run computes, measured on one GPU: the movielens K=100 ELBO over M = ceil(300 / N) users
(the rank's slice of Split('plate_1', .)), graph replay, no collective.   python3 tools/rank_share.py [K] [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for n in ([int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]):
    M = -(-300 // n)
    prob = bench.build_problem("cuda", M=M)
    s = bench.draw(prob, K)
    strat = alan.no_checkpoint if n > 1 else bench.strategy_for(1, K)
    for _ in range(4):
        v = s.elbo_nograd(strat, graph=True)
    t.cuda.synchronize()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        v = s.elbo_nograd(strat, graph=True)
    b.record()
    t.cuda.synchronize()
    print(f"K={K} N={n}: M={M} users per rank, {a.elapsed_time(b) / 50 * 1e3:.1f} us per evaluation (no all-reduce), elbo {float(v):.3f}")
