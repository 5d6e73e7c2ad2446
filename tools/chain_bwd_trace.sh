#!/bin/bash
# per-launch durations of the chain backward's rounds (rocprofv3 --kernel-trace):  bash tools/chain_bwd_trace.sh K
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-30}
O=gpurun_out/cbt$K
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/chain_bwd_prof.py 1000 $K 6 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import sys, csv
rows = sorted(({"name": r["Kernel_Name"][:60], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "g": r.get("Grid_Size_X", r.get("Grid_Size", "?"))}
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
tail = rows[-13:]
for r in tail:
    print(f"{(r['e'] - r['s']) / 1e3:8.2f} us  grid {r['g']:>8}  {r['name']}")
PY
rm -rf $O/t
