"""The A = G @ (1/scale^2) product of the outer-product Normal producer's backward: [n_value*n_loc, n_scale] x
[n_scale, E] with E = 18 -- rocBLAS tiles it 256x128 and takes 80 us for 50 MB of traffic.  Other formulations?
Usage: python tools/outer_bwd_gemm_probe.py"""
import time
import torch as t

nV, nL, nS, E = 9000, 30, 30, 18
G = t.randn(nV * nL, nS, device="cuda")
w = t.rand(nS, E, device="cuda")
G3 = G.view(nV, nL, nS)


def bench(name, fn):
    for _ in range(3):
        out = fn()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        out = fn()
    t.cuda.synchronize()
    print(f"{name:58s} {(time.perf_counter() - t0) / 30 * 1e6:8.1f} us   out {tuple(out.shape)}", flush=True)
    return out


ref = bench("G @ w  (2-D GEMM, M = 270000, N = 18)", lambda: G @ w)
b = bench("bmm(G3, w.expand(nV, nS, E))  (9000 batches of 30x30x18)", lambda: t.bmm(G3, w.expand(nV, nS, E)))
print("   max diff", (b.reshape(-1, E) - ref).abs().max().item())
c = bench("(w.T @ G.T).T  (M = 18, N = 270000)", lambda: (w.t() @ G.t()).t())
print("   max diff", (c - ref).abs().max().item())
d = bench("einsum vls,se->vle", lambda: t.einsum("vls,se->vle", G3, w))
bench("G.view(1000, 270, nS) bmm", lambda: t.bmm(G.view(1000, 270, nS), w.expand(1000, nS, E)))
bench("G.view(100, 2700, nS) bmm", lambda: t.bmm(G.view(100, 2700, nS), w.expand(100, nS, E)))
t.backends.cuda.preferred_blas_library("hipblaslt")
bench("hipBLASLt: G @ w", lambda: G @ w)
bench("hipBLASLt: bmm 9000", lambda: t.bmm(G3, w.expand(nV, nS, E)))
