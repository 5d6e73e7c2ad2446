#!/usr/bin/env python3
"""Turns gpurun_out/r4_raw/ (written on the GPU box by tools/collect_r4.sh) into the tracked round-4 summaries under
profiles/:  r4_bench.json (the bench line), r4_bench_default_kernel_stats.{csv,md} (rocprofv3 --kernel-trace --stats of
`python3 bench.py --no-extras`), r4_cases_kernel_stats.md (per-evaluation kernel budgets of the other workloads),
r4_fused_kernel_pmc.{json,md} (FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes: separate --pmc
runs, FETCH_SIZE doubled for wide coalesced reads), the pipeline probes, the timeline, the backward's precision tables."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "r4_raw")
OUT = os.path.join(ROOT, "profiles")
csv.field_size_limit(1 << 30)


def one(pattern):
    hits = glob.glob(os.path.join(RAW, pattern), recursive=True)
    assert hits, pattern
    return max(hits, key=os.path.getmtime)


bench = json.loads(open(os.path.join(RAW, "bench_full.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(OUT, "r4_bench.json"), "w"), indent=1)

src = one("stats/**/*kernel_stats.csv")
shutil.copy(src, os.path.join(OUT, "r4_bench_default_kernel_stats.csv"))
rows = list(csv.DictReader(open(src)))
evals = max(int(r["Calls"]) for r in rows if "normal_lse_x3_kernel" in r["Name"])
with open(os.path.join(OUT, "r4_bench_default_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extras` (round 4)\n\n"
            "Full CSV: `r4_bench_default_kernel_stats.csv`.  Kernel names truncated.  The run evaluates the movielens K=30 "
            f"ELBO {evals} times: warm-up and capture of the single-stream evaluation and of the pipeline's four lanes, 50 "
            "evaluations one after another (recorded launch list), the pipelined batches (4 lanes on streams of their own: the "
            "kernels of different evaluations OVERLAP there, which lengthens each one's own duration), then 50 eager for the "
            "per-kernel HIP events.  The averages below therefore mix kernels that ran alone with kernels that shared the chip; "
            "`r4_cases_kernel_stats.md` (ml K=30) has the one-after-another durations.\n\n"
            "| kernel | calls | per eval | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        calls = int(r["Calls"])
        if calls < evals // 4:
            continue
        f.write(f"| `{r['Name'][:100]}` | {calls} | {calls / evals:.1f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    f.write(f"\nbench.py's live HIP-event measurement of the fused plate step's launches (`r4_bench.json`, hipExtLaunchKernelGGL "
            f"start/stop events, evaluations one by one): {bench['roofline']['us_per_launch']:.1f} us.\n")

with open(os.path.join(OUT, "r4_cases_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python3 tools/prof_case.py <case> <K> <n>` (round 4)\n\n"
            "Evaluations one after another on one stream (recorded launch lists / graph replays; warm-up and capture included "
            "in the launch counts, hence the fractional launches per evaluation).  ml = movielens elbo_nograd (K=100: "
            "Split('plate_1', 38), the rank's chunks as one slice), vi / rws = one training iteration (sample -> elbo -> backward "
            "-> alan_amd.Adam), bus = bus_breakdown, ts = timeseries T=1000.\n\n")
    for path in sorted(glob.glob(os.path.join(RAW, "case_*_kernel_stats.csv"))):
        tag = os.path.basename(path)[len("case_"):-len("_kernel_stats.csv")]
        n = int(open(os.path.join(RAW, f"case_{tag}.n")).read())
        extra = 3 if tag.startswith(("vi", "rws")) else 6
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kstats2.py"), path, str(n + extra), tag.replace("_", " K=")],
                             capture_output=True, text=True).stdout
        f.write(out + "\n")


def pmc(which, counter, match):
    path = one(f"{which}/**/*counter_collection.csv")
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not any(m in r["Kernel_Name"] for m in match):
            continue
        key = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]))
        a = acc[key]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return {k: (v[0] / v[1], v[1], v[2] / v[1]) for k, v in acc.items()}


note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE doubled per "
        "MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced stream -- calibrated for 16-byte-per-lane reads; the "
        "fused kernels read 4 bytes per lane, so their absolute read figure is indicative only); KiB -> bytes")
match = ("normal_lse_x3_kernel", "normal_lse_bwd_kernel")
fetch, write = pmc("pmc_fetch", "FETCH_SIZE", match), pmc("pmc_write", "WRITE_SIZE", match)
res = {"source": note + " -- python3 tools/profile_fused.py", "kernels": {}}
md = ["# HBM traffic of the fused plate step (round 4)\n", note + ".\n",
      "| kernel | grid | launches | avg us | 2 x FETCH_SIZE (MB) | WRITE_SIZE (MB) | what it replaces |\n|---|---|---|---|---|---|---|"]
sizes = {30: 4 * 300 * 30 ** 3, 100: 4 * 300 * 100 ** 3}
for k in sorted(fetch, key=lambda k: (k[0], k[1])):
    fb, n, us = fetch[k][0] * 1024 * 2, fetch[k][1], fetch[k][2]
    wb = write.get(k, (0, 0, 0))[0] * 1024
    K = 100 if us > 100 else 30
    res["kernels"][f"{k[0]} grid {k[1]}"] = {"fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb, "launches": n,
                                             "mean_duration_us": us, "factor_bytes_never_materialised": sizes[K]}
    md.append(f"| `{k[0]}` | {k[1]} | {n} | {us:.1f} | {fb / 1e6:.2f} | {wb / 1e6:.2f} | a {sizes[K] / 1e6:.0f} MB factor written once and read once |")
json.dump(res, open(os.path.join(OUT, "r4_fused_kernel_pmc.json"), "w"), indent=1)
open(os.path.join(OUT, "r4_fused_kernel_pmc.md"), "w").write("\n".join(md) + "\n")

for src, dst, head in (("timeline_k30.txt", "r4_timeline_K30.txt", "# python3 tools/nlse_timeline.py 300 30 18 table (diagnostic build, make TIMELINE=1): where a wave of the K=30 launch spends its life -- the kernel an evaluation runs since ABI 14: its scale table built ahead of the launch, slices without 64-bit divisions, longer slices first\n"),
                       ("timeline_k30_own_table.txt", "r4_timeline_K30_own_table.txt", "# python3 tools/nlse_timeline.py 300 30 18: the same launch building its scale table itself (what a training forward runs)\n"),
                       ("chain_bwd.txt", "r4_chain_backward.txt", "# python3 tools/chain_bwd_probe.py 1000 30 100: the timeseries chain's backward, one launch for the tree (tree=1, default) against a launch per round (tree=0), HIP events around 50 back-to-back calls\n"),
                       ("producer_parts_kernel_stats.csv", "r4_producer_parts_kernel_stats.csv", ""),
                       ("pipeline_trace_30.txt", "r4_pipeline_timeline_K30.txt", "# bash tools/pipeline_trace.sh 30 300 4 4 600: the GPU's timeline of pipelined movielens K=30 evaluations (rocprofv3 kernel trace; the profiler slows the host's launches, so the period is longer than unprofiled)\n"),
                       ("pipeline_30.txt", "r4_pipeline_probe_K30.txt", "# python3 tools/pipeline_probe.py 30 300 3000: throughput of sample.EvalPipeline by lanes and issuing threads\n"),
                       ("pipeline_configs.txt", "r4_pipeline_other_configs.txt", "# python3 tools/pipeline_configs_probe.py: sample() + elbo with fresh particles (SamplingPipeline), bus_breakdown and timeseries evaluations, pipelined\n")):
    p = os.path.join(RAW, src)
    if os.path.exists(p):
        open(os.path.join(OUT, dst), "w").write(head + "".join(l for l in open(p) if "amdgpu.ids" not in l))
if os.path.exists(os.path.join(RAW, "bwd_prec_x2.md")) and os.path.exists(os.path.join(RAW, "bwd_prec_f32.md")):
    cur = open(os.path.join(OUT, "r4_fused_backward_precision.md")).read() if os.path.exists(os.path.join(OUT, "r4_fused_backward_precision.md")) else ""
    head = cur.split("# fused backward, V / U products on")[0] if "# fused backward, V / U products on" in cur else ""
    open(os.path.join(OUT, "r4_fused_backward_precision.md"), "w").write(
        head + "".join(l for l in open(os.path.join(RAW, "bwd_prec_x2.md")) if "amdgpu.ids" not in l) + "\n" +
        "".join(l for l in open(os.path.join(RAW, "bwd_prec_f32.md")) if "amdgpu.ids" not in l))
print("profiles/ written:", sorted(x for x in os.listdir(OUT) if x.startswith("r4_")))
