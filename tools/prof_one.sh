#!/bin/bash
# rocprofv3 per-kernel budget of one tools/prof_case.py workload:   gpurun -- 'bash tools/prof_one.sh bus 30 200'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_one_$1_$2
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
python3 tools/kstats2.py "$(find $O/t -name '*kernel_stats.csv' | head -1)" $3 "$1 K=$2" | tee $O/summary.md
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*kernel_trace.csv" -delete
