set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_pipe; rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 tools/pipeline_batch_probe.py 20 4 2>&1 | grep -v Warn | tail -n 8
timeout -k 10 300 python3 bench.py --no-extras --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipelined']['steady_state_us_per_eval'], d['single_stream']['us_per_eval'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
timeout -k 10 600 python -m pytest tests/test_e2e_host.py tests/test_device_noise.py tests/test_exchange.py -m gpu -x -q -k "pipeline or Pipeline or pipelined" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
