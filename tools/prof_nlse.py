#!/usr/bin/env python3
"""Runs the fused plate step's forward alone (no gradients) `iters` times at one S-ML shape, for rocprofv3:
    python3 tools/prof_nlse.py M K E iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim

M, K, Ev, iters = (int(x) for x in sys.argv[1:5])
g = t.Generator(device="cuda").manual_seed(0)
pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
z = t.randn(M, K, Ev, device="cuda", generator=g)
mu = t.randn(K, Ev, device="cuda", generator=g)
raw = 0.3 * t.randn(K, Ev, device="cuda", generator=g)
sm = [(t.randn(M, K, device="cuda", generator=g), (pl, Kz)) for _ in range(2)]
for _ in range(iters):
    E.normal_lse((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz, log_scale=True)
t.cuda.synchronize()
print("done", M, K, Ev, iters)
