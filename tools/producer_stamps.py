#!/usr/bin/env python3
"""In-kernel timeline of normal_outer_kernel (ALAN_NORMAL_ABLATE=9 makes every workgroup write its begin /
after-prologue / end wall-clock stamps over the head of the output)."""
import os, sys
os.environ["ALAN_NORMAL_ABLATE"] = "9"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M, D = 300, 18
z = t.randn(M, K, D, device="cuda")
mu = t.randn(K, D, device="cuda")
sc = t.rand(K, D, device="cuda") + 0.5
for it in range(3):
    out = E.normal_logprob((z, ("m", "kz")), (mu, ("kmu",)), (sc, ("kpsi",)), ("m", "kmu", "kpsi", "kz"))
    t.cuda.synchronize()
nwg = ((M * K + 255) // 256) * K
st = out.view(-1).view(t.int64)[: 4 * nwg].view(nwg, 4).cpu()
t0 = int(st[:, 0].min())
b, p, e, sm = (st[:, 0] - t0).float() / 100, (st[:, 1] - t0).float() / 100, (st[:, 2] - t0).float() / 100, st[:, 3]
print(f"{nwg} workgroups; times in us since the first workgroup began (100 MHz clock)")
print(f"begin   : min {b.min():6.2f} median {b.median():6.2f} max {b.max():6.2f}")
print(f"prologue: median duration {(p - b).median():6.2f}  max {(p - b).max():6.2f}   (end: max {p.max():6.2f})")
print(f"main    : median duration {(e - p).median():6.2f}  max {(e - p).max():6.2f}")
print(f"end     : min {e.min():6.2f} median {e.median():6.2f} max {e.max():6.2f}")
print("distinct CU ids:", len(set(sm.tolist())))
order = b.argsort()
for i in list(range(0, nwg, max(1, nwg // 12))):
    j = int(order[i])
    print(f"  wg#{j:4d} begin {b[j]:6.2f} pro {p[j]-b[j]:5.2f} main {e[j]-p[j]:5.2f} end {e[j]:6.2f}")
