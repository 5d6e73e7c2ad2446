#!/usr/bin/env python3
"""Turns gpurun_out/r3_raw/ (written on the GPU box by tools/collect_r3.sh) into the tracked round-3 summaries under
profiles/:  r3_bench.json (the bench line), r3_bench_default_kernel_stats.{csv,md} (rocprofv3 --kernel-trace --stats of
`python3 bench.py --no-extras`), r3_cases_kernel_stats.md (per-evaluation kernel budgets of the other workloads),
r3_fused_kernel_pmc.{json,md} and r3_rows_kernel_pmc.{json,md} (FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes: separate --pmc runs, FETCH_SIZE doubled for wide coalesced reads)."""
import csv, glob, json, os, shutil, io, contextlib, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "r3_raw")
OUT = os.path.join(ROOT, "profiles")
csv.field_size_limit(1 << 30)


def one(pattern):
    hits = glob.glob(os.path.join(RAW, pattern), recursive=True)
    assert hits, pattern
    return max(hits, key=os.path.getmtime)        # (gpurun merges into the local copy: older collections may linger)


bench = json.load(open(os.path.join(RAW, "bench_full.json")))
json.dump(bench, open(os.path.join(OUT, "r3_bench.json"), "w"), indent=1)

# ---- kernel stats of the default bench run
src = one("stats/**/*kernel_stats.csv")
shutil.copy(src, os.path.join(OUT, "r3_bench_default_kernel_stats.csv"))
rows = list(csv.DictReader(open(src)))
evals = max(int(r["Calls"]) for r in rows if "normal_lse_x3_kernel" in r["Name"])
with open(os.path.join(OUT, "r3_bench_default_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extras` (round 3)\n\n"
            "Full CSV: `r3_bench_default_kernel_stats.csv`.  Kernel names truncated.  The run evaluates the movielens K=30 "
            f"ELBO {evals} times (warm-up, capture, 5 + 50 replays of the captured evaluation -- through its recorded launch list, sample.DIRECT_REPLAY --, then 50 eager for the per-kernel HIP events).\n\n"
            "| kernel | calls | per eval | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        calls = int(r["Calls"])
        if calls < evals // 4:
            continue
        f.write(f"| `{r['Name'][:100]}` | {calls} | {calls / evals:.1f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    fk = [r for r in rows if "normal_lse_x3_kernel" in r["Name"]][0]
    f.write(f"\n`alan::normal_lse_x3_kernel` (the dominant kernel: the fused plate step) averages "
            f"{float(fk['AverageNs']) / 1e3:.1f} us in this trace; bench.py's live HIP-event measurement of the same launches "
            f"(`r3_bench.json`, a separate process on the same box, hipExtLaunchKernelGGL start/stop events) gives "
            f"{bench['roofline']['us_per_launch']:.1f} us.\n")

# ---- the other workloads
with open(os.path.join(OUT, "r3_cases_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python3 tools/prof_case.py <case> <K> <n>` (round 3)\n\n"
            "Graph replays (warm-up and capture included in the launch counts, hence the fractional launches per "
            "evaluation).  ml = movielens elbo_nograd (K=100: Split('plate_1', 38), the rank's chunks as one slice), vi / rws "
            "= one training iteration (sample -> elbo -> backward -> Adam), bus = bus_breakdown, ts = timeseries T=1000.\n\n")
    for path in sorted(glob.glob(os.path.join(RAW, "case_*_kernel_stats.csv"))):
        tag = os.path.basename(path)[len("case_"):-len("_kernel_stats.csv")]
        n = int(open(os.path.join(RAW, f"case_{tag}.n")).read())
        extra = 3 if tag.startswith(("vi", "rws")) else 6           # warm-up + capture passes
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kstats2.py"), path, str(n + extra), tag.replace("_", " K=")],
                             capture_output=True, text=True).stdout
        f.write(out + "\n")


# ---- PMC passes
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
md = ["# HBM traffic of the fused plate step (round 3)\n", note + ".\n",
      "| kernel | grid | launches | avg us | 2 x FETCH_SIZE (MB) | WRITE_SIZE (MB) | what it replaces |\n|---|---|---|---|---|---|---|"]
sizes = {30: 4 * 300 * 30 ** 3, 100: 4 * 300 * 100 ** 3}
for k in sorted(fetch, key=lambda k: (k[0], k[1])):
    fb, n, us = fetch[k][0] * 1024 * 2, fetch[k][1], fetch[k][2]
    wb = write.get(k, (0, 0, 0))[0] * 1024
    K = 100 if us > 100 else 30                 # (profile_fused.py runs exactly two sizes: K=30 and K=100, M=300)
    res["kernels"][f"{k[0]} grid {k[1]}"] = {"fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb, "launches": n,
                                             "mean_duration_us": us, "factor_bytes_never_materialised": sizes[K]}
    md.append(f"| `{k[0]}` | {k[1]} | {n} | {us:.1f} | {fb / 1e6:.2f} | {wb / 1e6:.2f} | a {sizes[K] / 1e6:.0f} MB factor written once and read once |")
json.dump(res, open(os.path.join(OUT, "r3_fused_kernel_pmc.json"), "w"), indent=1)
open(os.path.join(OUT, "r3_fused_kernel_pmc.md"), "w").write("\n".join(md) + "\n")

fetch, write = pmc("pmc_rows_fetch", "FETCH_SIZE", ("rows_kernel",)), pmc("pmc_rows_write", "WRITE_SIZE", ("rows_kernel",))
K, lit_M, big_M = 30, 300, 19200
algo = {m: 4 * (m * K ** 3 + m * K + K * K) for m in (lit_M, big_M)}
keys = sorted(fetch, key=lambda k: k[1])
assert len(keys) == 2, keys
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 "
                 "tools/profile_rows.py; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide "
                 "coalesced stream); KiB -> bytes", "kernel": keys[0][0]}
for name, k, m in (("literal_K30_M300", keys[0], lit_M), ("scaled_K30_M19200", keys[1], big_M)):
    fb, wb = fetch[k][0] * 1024 * 2, write[k][0] * 1024
    res[name] = {"fetch_bytes": fb, "write_bytes": wb, "algorithmic_bytes": algo[m], "traffic_bytes": fb + wb,
                 "traffic_over_algorithmic": (fb + wb) / algo[m], "mean_duration_us": fetch[k][2]}
json.dump(res, open(os.path.join(OUT, "r3_rows_kernel_pmc.json"), "w"), indent=1)
with open(os.path.join(OUT, "r3_rows_kernel_pmc.md"), "w") as f:
    f.write("# HBM traffic of the reduce_Ks kernel (rows.hip) on a materialised S-ML factor (round 3)\n\n" + res["source"] + ".\n\n"
            "| case | algorithmic MB | 2 x FETCH_SIZE MB | WRITE_SIZE MB | traffic / algorithmic | avg us (profiled) |\n|---|---|---|---|---|---|\n")
    for name in ("literal_K30_M300", "scaled_K30_M19200"):
        r = res[name]
        f.write(f"| {name} | {r['algorithmic_bytes'] / 1e6:.1f} | {r['fetch_bytes'] / 1e6:.1f} | {r['write_bytes'] / 1e6:.2f} | "
                f"{r['traffic_over_algorithmic']:.3f} | {r['mean_duration_us']:.1f} |\n")
# ---- the fused forward kernel's per-wave timeline, SQ counters, the bf16x3 probe
for src, dst, head in (("timeline_k30.txt", "r3_timeline_K30.txt", "# python3 tools/nlse_timeline.py 300 30 18 (diagnostic build, make TIMELINE=1): where a wave of the K=30 launch spends its life\n"),
                       ("timeline_k100.txt", "r3_timeline_K100.txt", "# python3 tools/nlse_timeline.py 300 100 18\n"),
                       ("mfma_bf16x3_probe.txt", "r3_mfma_bf16x3_probe.txt", "# tools/mfma_bf16x3_probe.hip: numerics and cycles of the bf16x3 tile (f32 path: 442 ns per tile per SIMD)\n")):
    p = os.path.join(RAW, src)
    if os.path.exists(p):
        body = "".join(l for l in open(p) if "amdgpu.ids" not in l)
        open(os.path.join(OUT, dst), "w").write(head + body)
for k in (30, 100):
    p = os.path.join(ROOT, "gpurun_out", f"pmc_nlse_300_{k}", "summary.md")
    if os.path.exists(p):
        open(os.path.join(OUT, f"r3_fused_forward_sq_counters_K{k}.md"), "w").write(
            f"# SQ counters of alan::normal_lse_x3_kernel at M=300, K={k}, E=18 (rocprofv3 --pmc, three passes, no tracing domains; tools/pmc_nlse.sh)\n\n"
            "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES and "
            "SQ_VALU_MFMA_BUSY_CYCLES count cycles.\n\n" + open(p).read())
# ---- bus_breakdown K=100: HBM traffic of the tile kernels (pair.hip) and of the rest of the evaluation
try:
    match = ("pair_lse_kernel", "pair_sum_kernel", "reduce_small", "reduce_group", "bernoulli_linear", "Cijk")
    fetch, write = pmc("pmc_bus_fetch", "FETCH_SIZE", match), pmc("pmc_bus_write", "WRITE_SIZE", match)
    Kb, Y, B, I = 100, 2, 3, 150
    algo = {"pair_lse_kernel<true>": 4 * (Y * B * Kb + 2 * Y * B * I * Kb + Y * B * I + Y * B * Kb * Kb),
            "pair_lse_kernel<false>": 4 * (2 * Y * B * Kb * Kb + Y * B * Kb + B * Y * Kb * Kb)}
    md = ["# HBM traffic of a bus_breakdown K=100 evaluation (round 3; python3 tools/prof_case.py bus 100 30)\n", note + ".\n",
          "Algorithmic bytes: every distinct input element once + the output once (the Bernoulli tile kernel reads alpha, the two "
          "evaluated dot terms [Y,B,I,K] and the observations and writes [Y,B,K,K]; the log-sum-exp tile kernel reads two "
          "[Y,B,K,K] factors and one [Y,B,K] and writes [3 plate elements, Y, K, K] partial results).\n",
          "| kernel | grid | launches | avg us | 2 x FETCH_SIZE (MB) | WRITE_SIZE (MB) | algorithmic (MB) | traffic / algorithmic | GB/s of algorithmic |\n|---|---|---|---|---|---|---|---|---|"]
    res = {"source": note + " -- python3 tools/prof_case.py bus 100 30", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k][2]):
        fb, n, us = fetch[k][0] * 1024 * 2, fetch[k][1], fetch[k][2]
        wb = write.get(k, (0, 0, 0))[0] * 1024
        a = next((v for kk, v in algo.items() if kk in k[0]), None)
        res["kernels"][f"{k[0]} grid {k[1]}"] = {"fetch_bytes": fb, "write_bytes": wb, "launches": n, "mean_duration_us": us, "algorithmic_bytes": a}
        md.append(f"| `{k[0][:70]}` | {k[1]} | {n} | {us:.1f} | {fb / 1e6:.2f} | {wb / 1e6:.2f} | " +
                  (f"{a / 1e6:.2f} | {(fb + wb) / a:.2f} | {a / us / 1e3:.0f} |" if a else "| | |"))
    json.dump(res, open(os.path.join(OUT, "r3_bus_K100_pmc.json"), "w"), indent=1)
    open(os.path.join(OUT, "r3_bus_K100_pmc.md"), "w").write("\n".join(md) + "\n")
except AssertionError as e:
    print("no bus K=100 PMC passes:", e)
# ---- plain-text evidence collected later in the round
for src, dst, head in (("chain_bwd_trace_K30.txt", "r3_chain_backward_launches_K30.txt", "# bash tools/chain_bwd_trace.sh 30: the last forward + backward of chain_logmmexp at T=1000, K=30, launch by launch (matrix-core backward)\n"),
                       ("chain_bwd_trace_K30_vector.txt", "r3_chain_backward_launches_K30_vector.txt", "# ALAN_CHAIN_BWD_MFMA=0 bash tools/chain_bwd_trace.sh 30: the vector-unit backward of rounds 1-2\n"),
                       ("chain_bwd_trace_K100.txt", "r3_chain_backward_launches_K100.txt", "# bash tools/chain_bwd_trace.sh 100 (matrix-core backward)\n"),
                       ("chain_bwd_trace_K100_vector.txt", "r3_chain_backward_launches_K100_vector.txt", "# ALAN_CHAIN_BWD_MFMA=0 bash tools/chain_bwd_trace.sh 100: the vector-unit backward of rounds 1-2\n"),
                       ("replay_floor.txt", "r3_replay_floor_probe.txt", "# python3 tools/replay_floor_probe.py 2000: what a replayed evaluation costs as a function of its launch count\n"),
                       ("chain_parts_K30.txt", "r3_chained_launch_parts_K30.txt", "# python3 tools/chain_parts.py 30: the chained launch (alan_normal_lse_chained) replayed with subsets of its parts\n"),
                       ("chain_parts_K100.txt", "r3_chained_launch_parts_K100.txt", "# python3 tools/chain_parts.py 100\n"),
                       ("chain_check.txt", "r3_chained_launch_ab.txt", "# python3 tools/chain_check.py: movielens evaluations as separate launches / sync-free chained / chained with hand-offs / one launch\n"),
                       ("batched_draws_ab.txt", "r3_batched_draws_ab.txt", "# tools/train_step_bench.py with dist.BATCH_DRAWS off / on (one process each, same box)\n"),
                       ("device_noise_ab.txt", "r3_device_noise_ab.txt", "# tools/train_step_bench.py with dist.DEVICE_NOISE off / on (one process each, same box)\n"),
                       ("exchange_probe.txt", "r3_exchange_probe.txt", "# python3 tools/exchange_probe.py 2: alan_exchange_sum between two processes SHARING this GPU (40 KB partials, 500 exchanges launched one by one): protocol and arithmetic, not the xGMI fabric\n"),
                       ("dispatch_probe.txt", "r3_dispatch_probe.txt", "# tools/dispatch_probe.hip: how long the dispatcher takes to start the 2048 waves of a launch, by workgroup size and register budget\n"),
                       ("replay_trace_graph.txt", "r3_replay_timeline_graph.txt", "# ALAN_AMD_DIRECT_REPLAY=0 bash tools/replay_trace.sh ml 30 100: the GPU's timeline of movielens K=30 evaluations replayed as a HIP graph (rocprofv3 kernel trace: gap since the previous kernel ended, duration)\n"),
                       ("replay_trace_direct.txt", "r3_replay_timeline_direct.txt", "# bash tools/replay_trace.sh ml 30 100: the same evaluations replayed through the recorded launch list (alan_calls_replay)\n"),
                       ("replay_alternate.txt", "r3_replay_alternate_probe.txt", "# python3 tools/replay_alternate_probe.py 2000: 1 / 2 / 4 graphs of the same evaluation replayed in turn\n"),
                       ("direct_replay_probe.txt", "r3_direct_replay_probe.txt", "# ALAN_AMD_DIRECT_REPLAY=0 python3 tools/direct_replay_probe.py 2000: an evaluation's three library calls issued again from the host (ctypes) against its HIP-graph replay\n"),
                       ("ts_train.txt", "r3_timeseries_training_iteration.txt", "# python3 tools/ts_train_probe.py 30 30: VI / RWS iteration of the Kalman timeseries model (T=1000, K=30) as one replayed graph\n")):
    p = os.path.join(RAW, src)
    if os.path.exists(p):
        body = "".join(l for l in open(p) if "amdgpu.ids" not in l and "UserWarning" not in l and "refine_names" not in l
                       and "[Gloo]" not in l and "socket.cpp" not in l)
        open(os.path.join(OUT, dst), "w").write(head + body)
print("profiles/ written:", sorted(x for x in os.listdir(OUT) if x.startswith("r3_")))
