#!/bin/bash
# Kernel start / end timestamps of a replayed evaluation (rocprofv3 --kernel-trace): per replay the kernels' durations
# and the gaps between them.   gpurun -- 'bash tools/trace_gaps.sh ml 30 200'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_$1_$2
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" $3 <<'PY' | tee $O/summary.txt
import sys, csv, statistics as st
rows = sorted(({"name": r["Kernel_Name"][:90], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
n = int(sys.argv[2])
# the last n replays: find the period (kernels per replay) from the tail
names = [r["name"] for r in rows]
for per in range(1, 80):
    if len(names) >= 3 * per and names[-per:] == names[-2 * per:-per] == names[-3 * per:-2 * per]:
        break
tail = rows[-per * (n - 5):]
print(f"{per} kernels per replay; over the last {len(tail) // per} replays (ns, medians):")
for k in range(per):
    durs = [tail[i * per + k]["e"] - tail[i * per + k]["s"] for i in range(len(tail) // per)]
    gaps = [tail[i * per + k]["s"] - tail[i * per + k - 1]["e"] for i in range(1 if k == 0 else 0, len(tail) // per)]
    print(f"  gap before {st.median(gaps):7.0f}   {tail[k]['name']:50s} duration {st.median(durs):7.0f}")
period = [tail[(i + 1) * per]["s"] - tail[i * per]["s"] for i in range(len(tail) // per - 1)]
print(f"replay period {st.median(period):.0f} ns")
PY
rm -rf $O/t
