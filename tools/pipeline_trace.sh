#!/bin/bash
# The GPU's timeline of PIPELINED evaluations (rocprofv3 --kernel-trace of tools/pipeline_run.py): per kernel its duration
# when evaluations overlap, how many kernels run at the same time, and what the chip spends its time on.
#   bash tools/pipeline_trace.sh K M lanes threads n
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pipeline_trace_$1_$2_$3
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/pipeline_run.py $1 $2 $3 $4 $5 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
grep "K=" $O/log
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import sys, csv, collections
rows = sorted(({"name": r["Kernel_Name"].replace("void ", "").replace("alan::", "")[:40], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]),
                "q": r.get("Queue_Id", "?")} for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
rows = rows[len(rows) // 2:]                        # the steady state of the last run
t0, t1 = rows[0]["s"], rows[-1]["e"]
durs = collections.defaultdict(list)
for r in rows:
    durs[r["name"]].append((r["e"] - r["s"]) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
span = (t1 - t0) / 1e3
print(f"second half of the trace: {len(rows)} kernels over {span:.1f} us, queues {sorted(set(r['q'] for r in rows))}")
for n, v in durs.items():
    print(f"  x{len(v):5d}   median {med(v):7.2f} us   mean {sum(v) / len(v):7.2f}   busy {sum(v) / span:5.2f} of the span   {n}")
# concurrency: time-weighted number of kernels in flight, and per kernel name the fraction of the span it has >= 1 running
ev = []
for r in rows:
    ev.append((r["s"], 1, r["name"])); ev.append((r["e"], -1, r["name"]))
ev.sort()
cur, last, hist = 0, t0, collections.Counter()
per, since = collections.Counter(), {}
run = collections.Counter()
for ts, d, n in ev:
    hist[cur] += ts - last
    for k in run:
        if run[k] > 0:
            per[k] += ts - last
    last = ts
    cur += d
    run[n] += d
tot = sum(hist.values())
print("kernels in flight (fraction of the span): " + "  ".join(f"{k}: {v / tot:.2f}" for k, v in sorted(hist.items())))
for k, v in per.items():
    print(f"  at least one running {v / tot:5.2f} of the span   {k}")
# a stretch of the timeline
for r in rows[:24]:
    print(f"  start {(r['s'] - t0) / 1e3:8.2f}  end {(r['e'] - t0) / 1e3:8.2f}  q{r['q']}  {r['name']}")
PY
rm -rf $O/t
