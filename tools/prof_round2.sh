#!/bin/bash
# rocprofv3 kernel stats of the round-2 workloads (on the GPU box):  gpurun --timeout 900 -- 'bash tools/prof_round2.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r2
rm -rf $O && mkdir -p $O
for spec in "ml 30 200" "ml 100 50" "vi 30 100" "rws 30 100" "bus 30 200" "ts 30 200" "ts 100 50"; do
  set -- $spec
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/$1_$2 --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/$1_$2.log 2>&1 || { echo "FAILED $spec"; tail -5 $O/$1_$2.log; exit 1; }
  f=$(find $O/$1_$2 -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/$1_$2_kernel_stats.csv
  echo "done $spec"
done
