#!/usr/bin/env python3
"""Per-evaluation kernel budget from a rocprofv3 --kernel-trace --stats CSV of `bench.py --no-extras`."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
evals = float(sys.argv[2]) if len(sys.argv) > 2 else 107
tot = 0
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"])
    per = calls / evals
    if per < 0.3:
        continue
    print(f"{r['Name'][:96]:96s} {per:4.1f}/eval  avg {avg/1e3:6.2f} us  per-eval {per*avg/1e3:6.1f} us")
    tot += per * avg / 1e3
print(f"sum per eval {tot:.1f} us")
