#!/usr/bin/env python3
"""cProfile of the eager (non-graph) ELBO evaluation: where the host time per evaluation goes."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
prob = bench.build_problem("cuda"); s = bench.draw(prob, 30)
for _ in range(20): s.elbo_nograd(alan.no_checkpoint)
t.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): v = s.elbo_nograd(alan.no_checkpoint)
t.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / n * 1e6:.0f} us/eval")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): v = s.elbo_nograd(alan.no_checkpoint)
t.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
