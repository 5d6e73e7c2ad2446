#!/bin/bash
# rocprofv3 per-kernel budget of one rank's share of C4 (tools/rank_share.py K N)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_rank_$1_$2
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python3 tools/rank_share.py $1 $2 > $O/log 2>&1 || { echo failed; tail -5 $O/log; exit 1; }
grep "K=" $O/log
python3 tools/kstats2.py "$(find $O/t -name '*kernel_stats.csv' | head -1)" 54 "rank share K=$1 N=$2" | tee $O/summary.md
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete; find $O -name "*kernel_trace.csv" -delete
