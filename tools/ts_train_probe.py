"""A VI / RWS training iteration of the reference's Kalman-filter timeseries model (tests/timeseries.py:5-50, T = 1000)
with a learned Normal approximate posterior per timestep, replayed as one HIP graph: ms per iteration.
   python3 tools/ts_train_probe.py [K] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t

import alan_amd as alan
from bench import build_timeseries_train_problem

if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for method in ("vi", "rws"):
        prob = build_timeseries_train_problem("cuda")
        opt = t.optim.Adam(list(prob.parameters()), lr=1e-2, capturable=True, fused=True, maximize=(method == "rws"))
        step = alan.GraphedStep(prob, K, opt, method=method)
        first = [float(step()) for _ in range(5)]
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            v = step()
        t.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        print(f"timeseries T=1000 K={K} {method}: {ms:.3f} ms / iteration; elbo {first[0]:.1f} -> {float(v):.1f}", flush=True)
        del step, opt, prob
