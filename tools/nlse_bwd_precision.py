#!/usr/bin/env python3
"""Gradient error of the fused plate step's backward (alan_normal_lse_backward) against fp64 autograd, per argument, on
the shapes of tests/test_gpu_fused_plate_step.py -- for the arithmetic this process's library runs its V / U products in
(ALAN_NLB_X2 unset: bf16 2-way split; ALAN_NLB_X2=0: fp32 MFMA; the knob is read once per process, so one run per form)
and, beside it, of fp32 autograd through torch.distributions on the same GPU: what the reference's backward is
(utils.py:218-220 + TorchDimDist.py:127-162 differentiated by autograd in fp32).
    python3 tools/nlse_bwd_precision.py
Per gradient: e_max = max |g - g64| / max |g64|, and the largest |g - g64| / (1e-5 max|g64| + 1e-4 |g64|) -- the reference's
own acceptance of a backward (tests/test_problem_vs_itself.py:71-88: rtol 1e-4, atol 1e-5 on O(1) moments), <= 1 passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim

SHAPES = [(300, 30, 30, 30, 18, 2, True), (38, 100, 100, 100, 18, 2, True), (7, 5, 4, 3, 3, 0, False),
          (11, 33, 9, 70, 20, 1, True), (5, 8, 40, 31, 1, 3, True), (3, 64, 2, 32, 31, 4, False),
          (4, 16, 5, 128, 9, 1, False), (6, 10, 7, 50, 18, 2, True), (9, 97, 3, 33, 17, 2, False),
          (2, 130, 2, 5, 30, 0, True), (40, 30, 30, 30, 18, 2, True), (5, 36, 3, 40, 7, 2, True),
          (300, 100, 100, 100, 18, 2, True)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]


def reference(z, mu, raw, smalls, log_scale, G, dtype, device):
    z, mu, raw = (x.detach().to(device, dtype).requires_grad_(True) for x in (z, mu, raw))
    sm = [x.detach().to(device, dtype).requires_grad_(True) for x, _ in smalls]
    M = z.shape[0]
    # (plate elements in blocks: the [M, NL, NS, NK] broadcast of the K=100 shape is 1.2 GB in fp32)
    tot_grads = None
    blk = max(1, min(M, int(2.5e8 // (mu.shape[0] * raw.shape[0] * z.shape[1] * z.shape[2]))))
    for m0 in range(0, M, blk):
        zs = z[m0:m0 + blk]
        sigma = raw.exp() if log_scale else raw
        lp = t.distributions.Normal(mu[None, :, None, None, :], sigma[None, None, :, None, :]).log_prob(zs[:, None, None, :, :]).sum(-1)
        tot = lp
        for x, (_, (_, kind)) in zip(sm, smalls):
            view = {"mk": lambda: x[m0:m0 + blk, None, None, :], "k": lambda: x[None, None, None, :],
                    "m": lambda: x[m0:m0 + blk, None, None, None], "km": lambda: x.t()[m0:m0 + blk, None, None, :]}[kind]()
            tot = tot + view
        mx = tot.amax(-1, keepdim=True)
        out = ((tot - mx).exp().sum(-1) + t.finfo(t.float32).eps).log() + mx.squeeze(-1)
        gs = t.autograd.grad((out.sum(0) * G.to(device, dtype)).sum(), [z, mu, raw, *sm], allow_unused=True)
        gs = [g if g is not None else t.zeros_like(p) for g, p in zip(gs, [z, mu, raw, *sm])]
        tot_grads = gs if tot_grads is None else [a + b for a, b in zip(tot_grads, gs)]
    return [g.double().cpu() for g in tot_grads]


form = "fp32 MFMA (ALAN_NLB_X2=0)" if os.environ.get("ALAN_NLB_X2") == "0" else "bf16 2-way split V / U (default until round 3)" \
    if os.environ.get("ALAN_NLB_X2") in (None, "1", "2") and os.environ.get("ALAN_NLB_X2") != "3" else "bf16 3-way split V / U"
print(f"# fused backward, V / U products on: {form}")
print("| shape (M, NK, NL, NS, E) | argument | fused: e_max | fused: worst / (1e-5 max + 1e-4 |g|) | torch fp32 autograd: e_max | torch fp32: worst |")
print("|---|---|---|---|---|---|")
worst_all = 0.0
for (M, NK, NL, NS, Ev, n_small, log_scale) in SHAPES:
    g = t.Generator().manual_seed(M * 7 + NK + NS)
    pl, K, dl, ds = Dim("plate", M), Dim("K", NK), Dim("Kl", NL), Dim("Ks", NS)
    z, mu, raw = t.randn(M, NK, Ev, generator=g), t.randn(NL, Ev, generator=g), 0.3 * t.randn(NS, Ev, generator=g)
    if not log_scale:
        raw = raw.exp()
    kinds = [((pl, K), "mk"), ((K,), "k"), ((pl,), "m"), ((K, pl), "km")]
    smalls = []
    for i in range(n_small):
        dims, kind = kinds[i]
        smalls.append((t.randn(*[d.size for d in dims], generator=g), (dims, kind)))
    G = t.randn(NL, NS, generator=g)
    ref = reference(z, mu, raw, smalls, log_scale, G, t.float64, "cuda")
    t32 = reference(z, mu, raw, smalls, log_scale, G, t.float32, "cuda")
    dev = [x.to("cuda").requires_grad_(True) for x in (z, mu, raw)]
    dsm = [x.to("cuda").requires_grad_(True) for x, _ in smalls]
    out, _ = E.normal_lse((dev[0], (pl, K)), (dev[1], (dl,)), (dev[2], (ds,)), [(x, d[0]) for x, (_, d) in zip(dsm, smalls)],
                          pl, K, log_scale=log_scale)
    grads = t.autograd.grad((out * G.to("cuda")).sum(), [*dev, *dsm])
    names = ["value", "loc", "log scale" if log_scale else "scale", *[f"small{i}" for i in range(n_small)]]
    for n, a, b, c in zip(names, grads, ref, t32):
        a = a.double().cpu()
        sc = float(b.abs().max()) + 1e-30
        def m(x):
            return float((x - b).abs().max()) / sc, float(((x - b).abs() / (1e-5 * sc + 1e-4 * b.abs())).max())
        (e1, w1), (e2, w2) = m(a), m(c)
        worst_all = max(worst_all, w1)
        print(f"| {M}, {NK}, {NL}, {NS}, {Ev} | {n} | {e1:.2e} | {w1:.3f} | {e2:.2e} | {w2:.3f} |", flush=True)
print(f"\nworst fused gradient against the reference's acceptance (rtol 1e-4, atol 1e-5 x max): {worst_all:.3f} (<= 1 passes)")
