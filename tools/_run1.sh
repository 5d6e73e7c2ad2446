set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 512 384 256 768; do ALAN_NLSE_BLOCKS=$B timeout -k 10 150 python3 tools/pipeline_probe_short.py 30 300 3000 "blocks=$B" 2>&1 | grep "K="; done
