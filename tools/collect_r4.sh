#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line plus the rocprofv3 evidence that profiles/r4_* is built from.
#   gpurun --timeout 1100 -- 'bash tools/collect_r4.sh'
# then, back in the container:  python tools/summarise_r4.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_raw
rm -rf $O && mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err || { echo "bench failed"; tail -n 5 $O/bench_full.err; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --no-extras > $O/stats.log 2>&1 || { echo "stats failed"; exit 1; }
echo "stats done"
for spec in "ml 30 200" "ml 100 50" "vi 30 100" "rws 30 100" "vi 100 20" "bus 30 200" "bus 100 50" "ts 30 200" "ts 100 50"; do
  set -- $spec
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/case_$1_$2 --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/case_$1_$2.log 2>&1 || { echo "FAILED $spec"; tail -n 5 $O/case_$1_$2.log; exit 1; }
  cp "$(find $O/case_$1_$2 -name '*kernel_stats.csv' | head -n 1)" $O/case_$1_$2_kernel_stats.csv
  echo "$3" > $O/case_$1_$2.n
  echo "case $spec done"
done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 tools/profile_fused.py > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 tools/profile_fused.py > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
echo "pmc done"
# the pipelined evaluations' timeline (what overlaps with what) and throughput by lanes / threads / plate size
bash tools/pipeline_trace.sh 30 300 4 4 600 > $O/pipeline_trace_30.txt 2>&1 || echo "pipeline trace failed"
timeout -k 10 300 python3 tools/pipeline_probe.py 30 300 3000 2>&1 | grep "K=" > $O/pipeline_30.txt || echo "pipeline probe failed"
timeout -k 10 300 python3 tools/pipeline_configs_probe.py 2>&1 | grep "lane" > $O/pipeline_configs.txt || echo "pipeline configs probe failed"
# the per-wave timeline of the fused plate step (diagnostic build: make -C alan_amd/csrc TIMELINE=1)
if [ -f tools/_build/timeline/libalan_mi355.so ]; then
  timeout -k 10 100 python3 tools/nlse_timeline.py 300 30 18 table > $O/timeline_k30.txt 2>&1 || echo "timeline K=30 failed"
  timeout -k 10 100 python3 tools/nlse_timeline.py 300 30 18 > $O/timeline_k30_own_table.txt 2>&1 || echo "timeline K=30 (own table) failed"
fi
timeout -k 10 300 python3 tools/chain_bwd_probe.py 1000 30 100 2>&1 | grep "T=" > $O/chain_bwd.txt || echo "chain backward probe failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/producer_parts --output-format csv -- python3 tools/producer_parts_probe.py > $O/producer_parts.log 2>&1 || echo "producer parts probe failed"
f=$(find $O/producer_parts -name '*kernel_stats.csv' | head -n 1); if [ -n "$f" ]; then cp "$f" $O/producer_parts_kernel_stats.csv; fi
find $O/producer_parts -name "*kernel_trace.csv" -delete
timeout -k 10 300 python3 tools/nlse_bwd_precision.py > $O/bwd_prec_x2.md 2>/dev/null || echo "backward precision probe failed"
ALAN_NLB_X2=0 timeout -k 10 300 python3 tools/nlse_bwd_precision.py > $O/bwd_prec_f32.md 2>/dev/null || echo "backward precision probe (fp32) failed"
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
find $O -path "*case_*" -name "*kernel_trace.csv" -delete
echo collected; ls $O | head -n 60
