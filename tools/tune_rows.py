#!/usr/bin/env python3
"""Sweeps the rows kernel's launch-geometry knobs (ALAN_ROWS_RBMAX / ALAN_ROWS_BLOCKS / ALAN_ROWS_LOGG) at the literal and
scaled S-ML sizes.  The library reads its knobs once per process, so every setting runs in a process of its own
(tools/rows_point.py)."""
import os, subprocess, sys
K = sys.argv[1] if len(sys.argv) > 1 else "30"
sizes = ["300", str(300 * 64 if int(K) <= 30 else 600)]
here = os.path.dirname(os.path.abspath(__file__))
grid = [(256, b, 0) for b in (600, 1200, 4096)] + [(128, b, 1) for b in (2048, 8192)] + [(64, b, 2) for b in (4096, 16384)]
for rb, blocks, logg in grid:
    env = dict(os.environ, ALAN_ROWS_RBMAX=str(rb), ALAN_ROWS_BLOCKS=str(blocks), ALAN_ROWS_LOGG=str(logg))
    out = subprocess.run([sys.executable, os.path.join(here, "rows_point.py"), K, *sizes], env=env, capture_output=True, text=True)
    print("\n".join(l for l in out.stdout.splitlines() if "K=" in l), flush=True)
