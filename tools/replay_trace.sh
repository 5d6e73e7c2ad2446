#!/bin/bash
# The GPU's timeline of replayed evaluations (rocprofv3 --kernel-trace): per kernel its duration and the gap since the
# previous kernel ended, for the graph replays of a tools/prof_case.py workload:  bash tools/replay_trace.sh ml 30 60
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/replay_trace_$1_$2
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import sys, csv, collections
rows = sorted(({"name": r["Kernel_Name"].replace("void ", "")[:44], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
tail = rows[-16:]
prev = rows[-17]["e"]
for r in tail:
    print(f"gap {(r['s'] - prev) / 1e3:7.2f} us   runs {(r['e'] - r['s']) / 1e3:7.2f} us   {r['name']}")
    prev = r["e"]
# medians over the second half of the trace, per kernel name: gap before it, its duration
half = rows[len(rows) // 2:]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(half[:-1], half[1:]):
    gaps[b["name"]].append((b["s"] - a["e"]) / 1e3)
    durs[b["name"]].append((b["e"] - b["s"]) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
print("medians over the second half of the trace:")
for n in gaps:
    print(f"  gap before {med(gaps[n]):6.2f} us   runs {med(durs[n]):6.2f} us   x{len(gaps[n])}   {n}")
PY
rm -rf $O/t
