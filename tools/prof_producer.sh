#!/bin/bash
# rocprofv3 kernel durations of the outer-product Normal producer at the movielens shape, K = $1 (default 30), with
# and without the transposed-store path.   gpurun -- 'bash tools/prof_producer.sh 30'
set -o pipefail
K=${1:-30}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_producer
rm -rf $O && mkdir -p $O
for ts in 1 0; do
  export ALAN_NORMAL_TS=$ts _CHILD=1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/ts$ts --output-format csv -- python3 tools/ablate_normal.py $K 300 > $O/ts$ts.log 2>&1 || { echo failed; tail -5 $O/ts$ts.log; exit 1; }
  echo "TS=$ts: $(tail -1 $O/ts$ts.log)"
  python3 tools/kstats2.py "$(find $O/ts$ts -name "*kernel_stats.csv" | head -1)" 23 | grep -i "normal_mfma"
done
