#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line plus the rocprofv3 evidence that profiles/ is built from.
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/summarise_profiles.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_raw
rm -rf $O && mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --no-extras > $O/stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 tools/profile_rows.py > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 tools/profile_rows.py > $O/pmc_write.log 2>&1 &&
find $O -name "*.csv" | head -20 && tail -c 600 $O/bench_full.json
