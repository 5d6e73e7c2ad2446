#!/usr/bin/env python3
"""Ablation of the outer-product Normal producer at the movielens K=100 chunk shape (library built with
`make -C alan_amd/csrc ABLATE=1`; env ALAN_NORMAL_ABLATE: 1 prologue only, 2 no stores, 4 no MFMAs; ALAN_NORMAL_MFMA=0
selects the vector kernel).  Usage: python tools/ablate_normal.py [K] [M]"""
import os, subprocess, sys
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M = int(sys.argv[2]) if len(sys.argv) > 2 else 38
if os.environ.get("_CHILD") != "1":
    for mfma in ("1", "0"):
        for ab in ("0", "1", "2", "4"):
            if mfma == "0" and ab == "4":
                continue
            env = dict(os.environ, _CHILD="1", ALAN_NORMAL_ABLATE=ab, ALAN_NORMAL_MFMA=mfma)
            out = subprocess.run([sys.executable, __file__, str(K), str(M)], env=env, capture_output=True, text=True)
            print(f"MFMA={mfma} ABLATE={ab}: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as t
from alan_amd import engine as E, native as N
from alan_amd.profiling import KernelTimer
from alan_amd.dims import Dim
g = t.Generator().manual_seed(0)
E_ = 18
dm, dz, dmu, dpsi = Dim("plate_1", M), Dim("K_z", K), Dim("K_mu", K), Dim("K_psi", K)
z = t.randn(M, K, E_, generator=g).cuda()
mu = t.randn(K, E_, generator=g).cuda()
sc = (t.rand(K, E_, generator=g) + 0.5).cuda()
kt = KernelTimer(min_bytes=1 << 20)
for _ in range(3):
    E.normal_logprob((z, (dm, dz)), (mu, (dmu,)), (sc, (dpsi,)), (dm, dmu, dpsi, dz))
t.cuda.synchronize()
with kt:
    for _ in range(20):
        E.normal_logprob((z, (dm, dz)), (mu, (dmu,)), (sc, (dpsi,)), (dm, dmu, dpsi, dz))
    t.cuda.synchronize()
res = [m for mode, b, m in kt.results()]
print(f"{sum(res) / len(res) * 1e3:7.2f} us per launch ({len(res)} launches, by HIP events)")
