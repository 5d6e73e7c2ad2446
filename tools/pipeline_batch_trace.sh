#!/bin/bash
# The GPU's timeline of ONE short pipelined batch (what the driver's bench line times: 20 evaluations between two
# synchronisations): rocprofv3 --kernel-trace of tools/pipeline_batch_probe.py, the last batch's kernels in start order.
#   bash tools/pipeline_batch_trace.sh [n] [lanes]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-20}; L=${2:-4}
O=gpurun_out/pipeline_batch_trace
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 tools/pipeline_batch_probe.py $N $L > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" $N <<'PY'
import sys, csv
n = int(sys.argv[2])
rows = sorted(({"name": r["Kernel_Name"].replace("void ", "").replace("alan::", "")[:22], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]),
                "q": r.get("Queue_Id", "?")} for r in csv.DictReader(open(sys.argv[1])) if "alan::" in r["Kernel_Name"]), key=lambda r: r["s"])
rows = rows[-3 * n:]                                # the last batch (the probe's last pipe.run(n))
t0 = rows[0]["s"]
print(f"last batch: {len(rows)} kernels, first start -> last end {(rows[-1]['e'] - t0) / 1e3:.1f} us (the profiler slows the host's launches)")
qs = sorted(set(r["q"] for r in rows))
print("start(us)  " + "  ".join(f"queue {q:>3s}          " for q in qs))
for r in rows:
    col = qs.index(r["q"])
    print(f"{(r['s'] - t0) / 1e3:8.1f}  " + " " * (20 * col) + f"{r['name'][:10]:10s} {(r['e'] - r['s']) / 1e3:5.1f}")
PY
