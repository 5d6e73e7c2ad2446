#!/usr/bin/env python3
"""Markdown per-evaluation kernel budget from a rocprofv3 --kernel-trace --stats kernel_stats.csv:
    python3 tools/kstats2.py <kernel_stats.csv> <evaluations in the trace> [title]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
evals = float(sys.argv[2])
print(f"### {sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]}\n")
print("| kernel | launches / eval | avg us | us / eval |\n|---|---|---|---|")
tot = 0
for r in sorted(rows, key=lambda r: -int(r["Calls"]) * float(r["AverageNs"])):
    calls, avg = int(r["Calls"]), float(r["AverageNs"])
    per = calls / evals
    if per < 0.3:
        continue
    print(f"| `{r['Name'][:110]}` | {per:.1f} | {avg/1e3:.2f} | {per*avg/1e3:.1f} |")
    tot += per * avg / 1e3
print(f"\nsum of kernel time per evaluation: **{tot:.1f} us**\n")
