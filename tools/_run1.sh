set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_tbl; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/ml30 --output-format csv -- python3 tools/prof_case.py ml 30 200 > $O/ml30.log 2>&1 && f=$(find $O/ml30 -name '*kernel_stats.csv' | head -n 1) && if [ -n "$f" ]; then head -n 4 "$f" | cut -c1-160; fi
timeout -k 10 200 python3 tools/pipeline_probe_short.py > $O/pipeline_short.txt 2>&1; echo "pp rc=$?"; tail -n 1 $O/pipeline_short.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log
