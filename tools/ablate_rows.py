#!/usr/bin/env python3
"""Ablation of the rows kernel (library built with `make -C alan_amd/csrc ABLATE=1`; env ALAN_ROWS_ABLATE:
1 = loads + LDS staging only, 2 = LDS reduction only)
across launch geometries; kernel time from library-recorded HIP events."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.profiling import KernelTimer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cases = {}
for M in (300, 300 * 64 if K <= 30 else 600):
    g = t.Generator(device="cuda").manual_seed(1234)
    F = -0.5 * t.randn(M, K, K, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    gz = -0.5 * t.randn(M, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    cases[M] = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
for blocks in (2048, 4096):
    for ab in (0,):
        os.environ["ALAN_ROWS_BLOCKS"] = str(blocks)
        os.environ["ALAN_ROWS_ABLATE"] = str(ab)
        line = f"BLOCKS={blocks:5d} ABLATE={ab}"
        for M, fac in cases.items():
            for _ in range(3):
                E.reduce_factors(fac, reduce=("z",), plate=("m",))
            t.cuda.synchronize()
            with KernelTimer() as kt:
                for _ in range(20 if M == 300 else 6):
                    E.reduce_factors(fac, reduce=("z",), plate=("m",))
                t.cuda.synchronize()
            ms = sorted(m for _, _, m in kt.results())
            med = ms[len(ms) // 2]
            nbytes = 4 * (M * K ** 3 + M * K + K * K)
            line += f" | M={M}: {med*1e3:8.1f} us {nbytes/med/1e9:6.2f} TB/s"
        print(line, flush=True)
