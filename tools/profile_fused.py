#!/usr/bin/env python3
"""Runs the fused plate step (alan_normal_lse) and its backward a few times at the S-ML sizes of BASELINE.json, for
rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE passes):  K=30, M=300 and K=100, M=300."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _scale_table import force_scale_table
force_scale_table()             # (the gradient-free launches: the kernel an evaluation runs, its scale table built ahead)

for M, K, Ev, iters in ((300, 30, 18, 20), (300, 100, 18, 6)):
    g = t.Generator(device="cuda").manual_seed(0)
    pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
    z = t.randn(M, K, Ev, device="cuda", generator=g).requires_grad_(True)
    mu = t.randn(K, Ev, device="cuda", generator=g).requires_grad_(True)
    raw = (0.3 * t.randn(K, Ev, device="cuda", generator=g)).requires_grad_(True)
    sm = [(t.randn(M, K, device="cuda", generator=g).requires_grad_(True), (pl, Kz)) for _ in range(2)]
    G = t.randn(K, K, device="cuda", generator=g)
    for _ in range(iters):
        with t.no_grad():
            E.normal_lse((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz, log_scale=True)
    for _ in range(iters):
        out, _ = E.normal_lse((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz, log_scale=True)
        t.autograd.grad((out,), [z, mu, raw, sm[0][0], sm[1][0]], (G,))
    t.cuda.synchronize()
    inputs = 4 * (M * K * Ev + 2 * K * Ev + 2 * M * K)
    print(f"M={M} K={K} E={Ev}: input bytes {inputs}, output bytes {4 * K * K}, factor never materialised would be "
          f"{4 * M * K ** 3} bytes", flush=True)
