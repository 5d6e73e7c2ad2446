#!/bin/bash
# SQ counters of the fused plate step's forward kernel at one shape (separate --pmc passes, no tracing domains):
#   gpurun -- 'bash tools/pmc_nlse.sh 300 100 18 6'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_nlse_$1_$2
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $O/p$i --output-format csv -- python3 tools/prof_nlse.py $1 $2 $3 $4 > $O/log$i 2>&1 || { echo "pass $i failed"; tail -5 $O/log$i; }
done
python3 - "$O" <<'PY'
import sys, glob, csv, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "normal_lse" not in k: continue
        k = k.split("(")[0][-60:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
with open(O + "/summary.md", "w") as out:
    for k, cs in tot.items():
        out.write(f"## {k}\n")
        for c, v in sorted(cs.items()):
            out.write(f"{c}: {v / n[(k, c)]:.0f} per launch ({n[(k, c)]} launches)\n")
print(open(O + "/summary.md").read())
PY
find $O -name "*agent_info.csv" -delete
