set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_chain; rm -rf $O; mkdir -p $O
timeout -k 10 120 python3 tools/chain_bwd_probe.py 37 30 > $O/probe_small.txt 2>&1; echo "small rc=$?"; grep "T=" $O/probe_small.txt
timeout -k 10 300 python3 tools/chain_bwd_probe.py 1000 30 100 > $O/probe.txt 2>&1; echo "probe rc=$?"; grep "T=" $O/probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_chain_batched.py tests/test_gpu_reduce.py tests/test_timeseries.py -m gpu -x -q -k "chain or timeseries" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest.log
