#!/bin/bash
# rocprofv3 kernel durations of chain_logmmexp alone (tools/chain_bench.py T K), wave kernel vs tree kernel.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_chain
rm -rf $O && mkdir -p $O
for wv in 1 0; do
  export ALAN_CHAIN_WAVE=$wv
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/w$wv --output-format csv -- python3 tools/chain_bench.py ${1:-1000} ${2:-30} > $O/w$wv.log 2>&1 || { echo failed; tail -5 $O/w$wv.log; exit 1; }
  echo "WAVE=$wv: $(grep 'us per chain' $O/w$wv.log)"
  python3 tools/kstats2.py "$(find $O/w$wv -name '*kernel_stats.csv' | head -1)" 27 | grep -i "chain"
done
