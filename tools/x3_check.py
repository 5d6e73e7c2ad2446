#!/usr/bin/env python3
"""alan_normal_lse against an fp64 torch evaluation of the same plate step, per (loc row, scale tile):
    python3 tools/x3_check.py M,NK,NL,NS,E[,n_small[,log_scale]] ...
Prints the largest absolute / relative error per scale tile of 32 and which loc rows hold mismatches (a debugging aid
for the tile / wave mapping of the fused kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.dims import Dim

shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(5, 37, 4, 128, 18, 3, 1)]
for shp in shapes:
    M, NK, NL, NS, Ev = shp[:5]
    n_small = shp[5] if len(shp) > 5 else 2
    log_scale = bool(shp[6]) if len(shp) > 6 else True
    g = t.Generator().manual_seed(M + NK + NS)
    pl, K, dl, ds = Dim("plate", M), Dim("K", NK), Dim("Kl", NL), Dim("Ks", NS)
    z = t.randn(M, NK, Ev, generator=g).cuda()
    mu = t.randn(NL, Ev, generator=g).cuda()
    raw = (0.3 * t.randn(NS, Ev, generator=g)).cuda()
    sc = raw if log_scale else raw.exp()
    small_dims = [(pl, K), (K,), (pl,), (K, pl)]
    smalls = [(t.randn(*[d.size for d in small_dims[i]], generator=g).cuda(), small_dims[i]) for i in range(n_small)]
    for rep in range(3):
        out, _ = E.normal_lse((z, (pl, K)), (mu, (dl,)), (sc, (ds,)), smalls, pl, K, log_scale=log_scale)
        lp = t.distributions.Normal(mu.double()[None, :, None, None, :], raw.double().exp()[None, None, :, None, :]).log_prob(
            z.double()[:, None, None, :, :]).sum(-1)                       # [M, NL, NS, NK]
        for i, (x, dims) in enumerate(smalls):
            xx = x.double()
            lp = lp + (lambda: xx[:, None, None, :], lambda: xx[None, None, None, :], lambda: xx[:, None, None, None],
                       lambda: xx.t()[:, None, None, :])[i]()
        want = t.logsumexp(lp, -1).sum(0)
        err = (out.double() - want).abs()
        rel = err / want.abs().clamp_min(1e-30)
        bad = (rel > 3e-5).nonzero()
        tiles = [float(rel[:, 32 * i:32 * i + 32].max()) for i in range((NS + 31) // 32)]
        print(f"{shp} rep {rep}: max abs {float(err.max()):.3e} rel {float(rel.max()):.3e}; per scale tile {['%.1e' % x for x in tiles]}; "
              f"bad rows {sorted(set(bad[:, 0].tolist()))} bad cols {sorted(set(bad[:, 1].tolist()))[:40]}", flush=True)
