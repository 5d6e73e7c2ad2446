#!/usr/bin/env python3
"""chain_logmmexp forward + backward n times (for rocprofv3):  python3 tools/chain_bwd_prof.py [T] [K] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import native as N
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = t.Generator().manual_seed(K)
ms = (-0.5 * t.randn(T, K, K, generator=g) ** 2 - 0.92 - t.log(t.tensor(float(K)))).cuda()
gv = t.randn(K, generator=g).cuda()
for _ in range(n):
    vec, _, tree = N.chain_logmmexp(ms)
    N.chain_logmmexp_backward(ms, tree, out_vec=vec, grad_vec=gv)
t.cuda.synchronize()
print("done")
