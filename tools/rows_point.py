#!/usr/bin/env python3
"""One point of the rows-kernel knob sweep (the knobs are read once per process: one process per setting):
    ALAN_ROWS_BLOCKS=600 python3 tools/rows_point.py [K] [M ...]
Kernel time from library-recorded HIP events (median of 30 launches), eager and inside a graph replay loop."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from alan_amd import engine as E
from alan_amd.profiling import KernelTimer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Ms = [int(x) for x in sys.argv[2:]] or [300]
knobs = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ALAN_ROWS"))
for M in Ms:
    g = t.Generator(device="cuda").manual_seed(1234)
    F = -0.5 * t.randn(M, K, K, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    gz = -0.5 * t.randn(M, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    fac = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
    for _ in range(3):
        E.reduce_factors(fac, reduce=("z",), plate=("m",))
    t.cuda.synchronize()
    with KernelTimer(min_bytes=1 << 16) as kt:
        for _ in range(30):
            E.reduce_factors(fac, reduce=("z",), plate=("m",))
        t.cuda.synchronize()
    ms = sorted(m for _, _, m in kt.results())
    nbytes = 4 * (M * K ** 3 + M * K + K * K)
    print(f"{knobs or 'default':40s} K={K} M={M}: median {ms[len(ms)//2]*1e3:7.2f} us  min {ms[0]*1e3:7.2f} us  "
          f"{nbytes/ms[len(ms)//2]/1e9:5.2f} TB/s  ({nbytes/ms[len(ms)//2]/8e9*100:4.1f} % of 8 TB/s)", flush=True)
