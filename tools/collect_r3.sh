#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line plus the rocprofv3 evidence that profiles/r3_* is built from.
#   gpurun --timeout 1100 -- 'bash tools/collect_r3.sh'
# then, back in the container:  python tools/summarise_r3.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_raw
rm -rf $O && mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { echo "bench failed"; tail -5 $O/bench_full.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --no-extras > $O/stats.log 2>&1 || { echo "stats failed"; exit 1; }
for spec in "ml 30 200" "ml 100 50" "vi 30 100" "rws 30 100" "vi 100 20" "bus 30 200" "bus 100 50" "ts 30 200" "ts 100 50"; do
  set -- $spec
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/case_$1_$2 --output-format csv -- python3 tools/prof_case.py $1 $2 $3 > $O/case_$1_$2.log 2>&1 || { echo "FAILED $spec"; tail -5 $O/case_$1_$2.log; exit 1; }
  cp "$(find $O/case_$1_$2 -name '*kernel_stats.csv' | head -1)" $O/case_$1_$2_kernel_stats.csv
  echo "$3" > $O/case_$1_$2.n
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/fused_stats --output-format csv -- python3 tools/profile_fused.py > $O/fused_stats.log 2>&1 || { echo "fused stats failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 tools/profile_fused.py > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 tools/profile_fused.py > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_rows_fetch --output-format csv -- python3 tools/profile_rows.py > $O/pmc_rows_fetch.log 2>&1 || { echo "pmc rows fetch failed"; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_rows_write --output-format csv -- python3 tools/profile_rows.py > $O/pmc_rows_write.log 2>&1 || { echo "pmc rows write failed"; exit 1; }
# the per-wave timeline of the fused plate step (diagnostic build: make -C alan_amd/csrc TIMELINE=1) and its SQ counters
timeout -k 10 100 python3 tools/nlse_timeline.py 300 30 18 > $O/timeline_k30.txt 2>&1 || echo "timeline K=30 failed"
timeout -k 10 100 python3 tools/nlse_timeline.py 300 100 18 > $O/timeline_k100.txt 2>&1 || echo "timeline K=100 failed"
bash tools/pmc_nlse.sh 300 100 18 6 > $O/pmc_sq_k100.txt 2>&1 || echo "SQ counters K=100 failed"
bash tools/pmc_nlse.sh 300 30 18 20 > $O/pmc_sq_k30.txt 2>&1 || echo "SQ counters K=30 failed"
./tools/_build/mfma_bf16x3_probe > $O/mfma_bf16x3_probe.txt 2>&1 || echo "probe failed"
# ---- later in round 3: bus_breakdown K=100 (pair.hip) HBM traffic, the chain backward's launches, what a launch costs
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_bus_fetch --output-format csv -- python3 tools/prof_case.py bus 100 30 > $O/pmc_bus_fetch.log 2>&1 || echo "pmc bus fetch failed"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_bus_write --output-format csv -- python3 tools/prof_case.py bus 100 30 > $O/pmc_bus_write.log 2>&1 || echo "pmc bus write failed"
for K in 30 100; do
  bash tools/chain_bwd_trace.sh $K > $O/chain_bwd_trace_K$K.txt 2>&1 || echo "chain bwd trace K=$K failed"
  ALAN_CHAIN_BWD_MFMA=0 bash tools/chain_bwd_trace.sh $K > $O/chain_bwd_trace_K${K}_vector.txt 2>&1 || echo "chain bwd (vector) trace K=$K failed"
done
timeout -k 10 100 python3 tools/replay_floor_probe.py 2000 > $O/replay_floor.txt 2>&1 || echo "replay floor probe failed"
timeout -k 10 150 python3 tools/chain_parts.py 30 > $O/chain_parts_K30.txt 2>&1 || echo "chain parts K=30 failed"
timeout -k 10 150 python3 tools/chain_parts.py 100 > $O/chain_parts_K100.txt 2>&1 || echo "chain parts K=100 failed"
timeout -k 10 300 python3 tools/chain_check.py > $O/chain_check.txt 2>&1 || echo "chain check failed"
for b in 0 1; do ALAN_AMD_BATCH_DRAWS=$b timeout -k 10 200 python3 tools/train_step_bench.py 2>/dev/null | grep "graph replay" | sed "s/^/BATCH_DRAWS=$b /"; done > $O/batched_draws_ab.txt 2>&1 || echo "train A/B failed"
# ---- last part of round 3: the one-shot exchange between processes sharing this GPU, the dispatcher's ramp, training
# iterations of the timeseries model
timeout -k 10 120 python3 tools/exchange_probe.py 2 > $O/exchange_probe.txt 2>&1 || echo "exchange probe failed"
[ -x tools/_build/dispatch_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/_build/dispatch_probe > /dev/null 2>&1
timeout -k 10 60 ./tools/_build/dispatch_probe > $O/dispatch_probe.txt 2>&1 || echo "dispatch probe failed"
timeout -k 10 200 python3 tools/ts_train_probe.py 30 30 > $O/ts_train.txt 2>&1 || echo "ts train probe failed"
for b in 0 1; do ALAN_AMD_DEVICE_NOISE=$b timeout -k 10 200 python3 tools/train_step_bench.py 2>/dev/null | grep "graph replay" | sed "s/^/DEVICE_NOISE=$b /"; done > $O/device_noise_ab.txt 2>&1 || echo "noise A/B failed"
# ---- replays: the GPU's timeline as a graph replay and through the recorded launch list, the probes behind sample.DIRECT_REPLAY
ALAN_AMD_DIRECT_REPLAY=0 bash tools/replay_trace.sh ml 30 100 > $O/replay_trace_graph.txt 2>&1 || echo "replay trace (graph) failed"
bash tools/replay_trace.sh ml 30 100 > $O/replay_trace_direct.txt 2>&1 || echo "replay trace (direct) failed"
timeout -k 10 200 python3 tools/replay_alternate_probe.py 2000 > $O/replay_alternate.txt 2>&1 || echo "alternate probe failed"
ALAN_AMD_DIRECT_REPLAY=0 timeout -k 10 200 python3 tools/direct_replay_probe.py 2000 > $O/direct_replay_probe.txt 2>&1 || echo "direct replay probe failed"
# keep what the summariser reads, drop the bulky traces
find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
find $O -path "*case_*" -name "*kernel_trace.csv" -delete
echo collected; ls $O | head -40
