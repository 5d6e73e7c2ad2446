#!/usr/bin/env python3
"""Times the fused plate step (alan_normal_lse) and its backward on the S-ML shapes:
    python3 tools/nlse_bench.py [iters]
Per shape: forward alone, forward saving lse, backward (all gradients) and backward (small factors only), by HIP events
around `iters` back-to-back launches; plus the fp32 MFMA floor of the forward (padded tiles x steps x 64 cycles)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
shapes = [(300, 30, 18), (300, 100, 18), (38, 100, 18), (38, 30, 18), (4800, 30, 18)]
if len(sys.argv) > 2:
    shapes = [tuple(int(x) for x in sys.argv[2].split(","))]


def timed(fn):
    for _ in range(3):
        fn()
    t.cuda.synchronize()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    t.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for M, K, Ev in shapes:
    g = t.Generator(device="cuda").manual_seed(0)
    pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
    z = t.randn(M, K, Ev, device="cuda", generator=g)
    mu = t.randn(K, Ev, device="cuda", generator=g)
    raw = 0.3 * t.randn(K, Ev, device="cuda", generator=g)
    sm = [(t.randn(M, K, device="cuda", generator=g), (pl, Kz)) for _ in range(2)]
    G = t.randn(K, K, device="cuda", generator=g)
    args = lambda zz, mm, rr, ss: ((zz, (pl, Kz)), (mm, (dl,)), (rr, (ds,)), ss, pl, Kz)
    fwd = timed(lambda: E.normal_lse(*args(z, mu, raw, sm), log_scale=True))
    zr, mr, rr = (x.clone().requires_grad_(True) for x in (z, mu, raw))
    sr = [(x.clone().requires_grad_(True), d) for x, d in sm]
    out, _ = E.normal_lse(*args(zr, mr, rr, sr), log_scale=True)
    fwd_lse = timed(lambda: E.normal_lse(*args(zr, mr, rr, sr), log_scale=True))
    bwd = timed(lambda: t.autograd.grad((out,), [zr, mr, rr, sr[0][0], sr[1][0]], (G,), retain_graph=True))
    out2, _ = E.normal_lse(*args(z, mu, raw, sr), log_scale=True)
    bwd_small = timed(lambda: t.autograd.grad((out2,), [sr[0][0]], (G,), retain_graph=True))
    nt = M * math.ceil(K / 32) * K * math.ceil(K / 32)
    steps = (Ev + 2) // 2
    floor = nt * steps * 64 / 1024 / 2.4e3          # us: 1024 SIMDs, 64 cycles per 32x32x2 MFMA, 2.4 GHz
    flops = 2.0 * M * K * K * K * Ev
    print(f"M={M} K={K} E={Ev}: fwd {fwd:.1f} us (+lse {fwd_lse:.1f})  bwd {bwd:.1f} us  bwd small-only {bwd_small:.1f} us | "
          f"fwd MFMA floor {floor:.1f} us, algorithmic {flops / fwd / 1e6:.1f} TFLOP/s fwd")
