#!/usr/bin/env python3
"""Runs the dominant reduce_Ks kernel (S-ML plate step) a few times at the literal movielens size and in
the bandwidth regime, for rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for M, iters in ((300, 20), (300 * 64 if K <= 30 else 300 * 2, 6)):
    g = t.Generator(device="cuda").manual_seed(1234)
    F = -0.5 * t.randn(M, K, K, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    gz = -0.5 * t.randn(M, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    fac = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
    for _ in range(iters):
        out, _ = E.reduce_factors(fac, reduce=("z",), plate=("m",))
    t.cuda.synchronize()
    print(f"K={K} M={M}: algorithmic bytes per launch = {4 * (M * K**3 + M * K + K * K)}", flush=True)
    del F, gz
