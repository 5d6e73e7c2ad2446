#!/bin/bash
# kernel-only durations (rocprofv3 kernel trace) of the fused plate step and its backward under tuning knobs:
#   gpurun -- 'bash tools/nlse_sweep.sh "300,30,18" "ALAN_NLSE_BLOCKS=256" "ALAN_NLSE_BLOCKS=1536 ALAN_NLB_GY=1" ...'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
shape=$1; shift
mkdir -p gpurun_out/sweep
i=0
for setting in "default" "$@"; do
  i=$((i+1))
  d=gpurun_out/sweep/run_$i
  rm -rf $d
  if [ "$setting" = "default" ]; then
    timeout -k 10 120 rocprofv3 --kernel-trace -d $d --output-format csv -- python3 tools/nlse_bench.py 20 $shape > $d.log 2>&1
  else
    env $setting timeout -k 10 120 rocprofv3 --kernel-trace -d $d --output-format csv -- python3 tools/nlse_bench.py 20 $shape > $d.log 2>&1
  fi
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  echo "== $shape | $setting"
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "normal_lse" not in n and "nlb_" not in n:
        continue
    key = (n.split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"   {k[0]:42s} grid {k[1]:>8s} x {k[2]:>3s}  n={len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
PY
done
