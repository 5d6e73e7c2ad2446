#!/usr/bin/env python3
"""
bench.py -- ELBO evals/sec of the movielens model at K=30 on MI355X (BASELINE.json metric), plus the
HBM-roofline fraction of the dominant reduce_Ks kernel and the CPU baseline timed on the same box.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE ELBO evaluation (``sample.elbo_nograd``) of the movielens-shaped model
(examples/models/movielens/movielens.py:39-82 of the reference: M=300 users, N=5 films, d_z=18) on a
fixed, pre-drawn K-particle sample -- per-factor log-probs on PyTorch-ROCm, reduce_Ks / plate sums
in libalan_mi355.so.  Data are synthetic (same shapes; no network).  With --gpus N > 1 the SAME ELBO
is sharded: ``Split('plate_1', ceil(300/N), shard=True)`` gives each rank one slice of the user plate and
the per-rank [K_mu_z, K_psi_z] partials meet in one RCCL all-reduce(SUM)  => "strong" scaling.
"""
import argparse
import json
import math
import os
import sys
import time

import torch as t

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
M_USERS, N_FILMS, D_Z = 300, 5, 18


def build_problem(device, M=M_USERS, seed=0):
    import alan_amd as alan
    from alan_amd import Normal, Bernoulli, Plate, BoundPlate, Problem, Data, OptParam
    g = t.Generator().manual_seed(seed)
    x = t.randn(M, N_FILMS, D_Z, generator=g).refine_names("plate_1", "plate_2", None)
    obs = (t.rand(M, N_FILMS, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
    P = Plate(
        mu_z=Normal(t.zeros((D_Z,)), t.ones((D_Z,))),
        psi_z=Normal(t.zeros((D_Z,)), t.ones((D_Z,))),
        plate_1=Plate(
            z=Normal("mu_z", lambda psi_z: psi_z.exp()),
            plate_2=Plate(obs=Bernoulli(logits=lambda z, x: z @ x)),
        ),
    )
    Q = Plate(
        mu_z=Normal(OptParam(t.zeros((D_Z,))), OptParam(t.zeros((D_Z,)), transformation=t.exp)),
        psi_z=Normal(OptParam(t.zeros((D_Z,))), OptParam(t.zeros((D_Z,)), transformation=t.exp)),
        plate_1=Plate(
            z=Normal(OptParam(t.zeros((D_Z,))), OptParam(t.zeros((D_Z,)), transformation=t.exp)),
            plate_2=Plate(obs=Data()),
        ),
    )
    sizes = {"plate_1": M, "plate_2": N_FILMS}
    prob = Problem(BoundPlate(P, sizes, inputs={"x": x}), BoundPlate(Q, sizes, inputs={"x": x}), {"obs": obs})
    return prob.to(device)


def build_bus_problem(device, seed=0):
    """bus_breakdown-shaped model (examples/models/bus_breakdown/bus_breakdown.py:38-100 of the reference):
    3 nested plates Year=2 / Borough=3 / ID=150, covariates of width 8 and 59, Bernoulli observations."""
    from alan_amd import Normal, Bernoulli, Plate, BoundPlate, Group, Problem, Data, OptParam
    g = t.Generator().manual_seed(seed)
    Y, B, I, nr, nb = 2, 3, 150, 8, 59
    names = ("plate_Year", "plate_Borough", "plate_ID")
    inp = {"run_type": (t.rand(Y, B, I, nr, generator=g) < 0.2).float().refine_names(*names, None),
           "bus_company_name": (t.rand(Y, B, I, nb, generator=g) < 0.05).float().refine_names(*names, None)}
    obs = (t.rand(Y, B, I, generator=g) < 0.66).float().refine_names(*names)
    P = Plate(
        psi=Normal(t.zeros((nr,)), t.ones((nr,))), phi=Normal(t.zeros((nb,)), t.ones((nb,))),
        sigma_beta=Normal(0, 1), mu_beta=Normal(0, 1),
        plate_Year=Plate(
            beta=Normal("mu_beta", lambda sigma_beta: sigma_beta.exp()), sigma_alpha=Normal(0, 1),
            plate_Borough=Plate(
                alpha=Normal("beta", lambda sigma_alpha: sigma_alpha.exp()),
                plate_ID=Plate(obs=Bernoulli(logits=lambda alpha, phi, psi, run_type, bus_company_name:
                                             (alpha + phi @ bus_company_name + psi @ run_type))))))
    Q = Plate(
        global_latents=Group(
            psi=Normal(OptParam(t.zeros(nr)), OptParam(t.zeros(nr), transformation=t.exp)),
            phi=Normal(OptParam(t.zeros(nb)), OptParam(t.zeros(nb), transformation=t.exp)),
            sigma_beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
            mu_beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp))),
        plate_Year=Plate(
            year_latents=Group(beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
                               sigma_alpha=Normal(OptParam(0.), OptParam(0., transformation=t.exp))),
            plate_Borough=Plate(alpha=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
                                plate_ID=Plate(obs=Data()))))
    sizes = {"plate_Year": Y, "plate_Borough": B, "plate_ID": I}
    prob = Problem(BoundPlate(P, sizes, inputs=inp), BoundPlate(Q, sizes, inputs=inp), {"obs": obs})
    return prob.to(device)


def build_timeseries_problem(device, T=1000, seed=0):
    """Kalman-filter test model of the reference (tests/timeseries.py:5-50) at T=1000."""
    from alan_amd import Normal, Timeseries, Plate, BoundPlate, Problem, Data
    g = t.Generator().manual_seed(seed)
    y = t.randn(T, generator=g).refine_names("T")
    P = Plate(init=Normal(0, 1.0),
              T=Plate(ts=Timeseries("init", Normal(lambda prev: 0.9 * prev, 0.1)), obs=Normal("ts", 1.0)))
    Q = Plate(init=Normal(0, 1), T=Plate(ts=Normal(0, 1), obs=Data()))
    prob = Problem(BoundPlate(P, {"T": T}), BoundPlate(Q, {"T": T}), {"obs": y})
    return prob.to(device)


def build_timeseries_train_problem(device, T=1000, seed=0):
    """The same model with a learned approximate posterior: a Normal per timestep (OptParam location and log-scale of
    shape [T], Param.py:18-25) -- what a VI / RWS iteration on it trains."""
    from alan_amd import Normal, Timeseries, Plate, BoundPlate, Problem, Data, OptParam
    g = t.Generator().manual_seed(seed)
    y = t.randn(T, generator=g).refine_names("T")
    P = Plate(init=Normal(0, 1.0),
              T=Plate(ts=Timeseries("init", Normal(lambda prev: 0.9 * prev, 0.1)), obs=Normal("ts", 1.0)))
    Q = Plate(init=Normal(OptParam(0.0), OptParam(0.0, transformation=t.exp)),
              T=Plate(ts=Normal(OptParam(0.0), OptParam(0.0, transformation=t.exp)), obs=Data()))
    prob = Problem(BoundPlate(P, {"T": T}), BoundPlate(Q, {"T": T}), {"obs": y})
    return prob.to(device)


def draw(prob, K, seed=1):
    t.manual_seed(seed)
    if t.cuda.is_available():
        t.cuda.manual_seed_all(seed)
    return prob.sample(K, reparam=False)


def shard_split_size(M, world):
    """Largest split_size whose chunking of the plate (Split.py:84-95) has at least one chunk per rank."""
    from alan_amd.split import chunk_sizes
    size = max(1, math.ceil(M / world))
    while size > 1 and (size >= M or len(chunk_sizes(M, size)) < world):
        size -= 1
    return size


def strategy_for(world, K, M=M_USERS):
    import alan_amd as alan
    if world > 1:
        return alan.Split("plate_1", shard_split_size(M, world), shard=True)
    if K >= 100:
        return alan.Split("plate_1", 38)        # reference launch line: examples/run_movielens.sh
    return alan.no_checkpoint


def timed_evals(sample, strat, steps, warmup, world, timer=None, graph=False):
    import torch.distributed as dist
    val = None
    for _ in range(warmup):
        val = sample.elbo_nograd(strat, graph=graph)
    if world > 1:
        dist.barrier()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    if timer is not None:
        with timer:
            for _ in range(steps):
                val = sample.elbo_nograd(strat, graph=graph)
    else:
        for _ in range(steps):
            val = sample.elbo_nograd(strat, graph=graph)
    t.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = t.tensor([dt], device="cuda", dtype=t.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    return dt, float(val)


def pipelined_figure(sample, strat, n, lanes, single_us):
    """A secondary configuration through sample.EvalPipeline: n evaluations, every result checked against the eager value."""
    try:
        with t.no_grad():
            ev = float(sample.elbo_nograd(strat, graph=False))
        pipe = sample.pipeline(strat, lanes=lanes, results=max(64, -(-n // lanes) + 8))
        pipe.run(2 * lanes)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        vals = pipe.run(n)
        t.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        worst = float((vals - ev).abs().max()) / abs(ev)
        pipe.close()
        sample.__dict__.get("_pipelines", {}).clear()
        if not worst <= 2e-6:
            return {"error": f"pipelined evaluations differ from the eager value by {worst:.3g} (relative)"}
        return {"n_streams": lanes, "evaluations": n, "us_per_eval": dt * 1e6, "evals_per_s": 1 / dt,
                "max_rel_diff_vs_eager": worst, "against_one_after_another": single_us / (dt * 1e6)}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def timed_pipeline(sample, strat, steps, warmup, lanes, eager_value):
    """`steps` INDEPENDENT evaluations through sample.EvalPipeline (alan_pipeline_*: `lanes` recorded copies of the
    evaluation, each on a stream of its own, issued round-robin by the library's threads), timed as the contract says
    (synchronize on both sides), EVERY result checked against the eager evaluation's value."""
    pipe = sample.pipeline(strat, lanes=lanes)
    pipe.run(max(warmup, 1))
    pipe.run(max(warmup, 1))
    t.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.submit(steps)
    vals = pipe.results(copy=False)         # (the ELBOs where the last launch of each evaluation left them: a view, no kernel)
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = vals.clone()
    worst = float((vals - eager_value).abs().max()) / abs(eager_value)
    if not worst <= 2e-6:
        raise RuntimeError(f"pipelined evaluations differ from the eager value by {worst:.3g} (relative)")
    # the steady state beside it: the same pipeline over 150 x as many evaluations (what a long loop sees)
    n_long = min(150 * steps, lanes * pipe.capacity)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(n_long)
    t.cuda.synchronize()
    dt_long = (time.perf_counter() - t0) / n_long
    return dt, float(vals[-1]), worst, dt_long


def sample_on_cpu(sample, cpu_prob):
    """The same particles as a Sample of the CPU problem (fresh Dim objects, matched by name)."""
    import alan_amd as alan
    from alan_amd.dims import PT, Dim
    Kd = {g: Dim(str(d), d.size) for g, d in sample.groupvarname2Kdim.items()}
    by_name = {str(d): d for d in Kd.values()}
    by_name.update(cpu_prob.all_platedims)

    def conv(tree):
        return {k: (conv(v) if isinstance(v, dict) else PT(v.x.detach().cpu(), [by_name[str(d)] for d in v.dims]))
                for k, v in tree.items()}

    return alan.Sample(problem=cpu_prob, sample=conv(sample._pt_detached), groupvarname2Kdim=Kd,
                       sampler=sample.sampler, reparam=False)


def cpu_baseline(K, gpu_sample, gpu_elbo, budget_s=20.0):
    """The same ELBO -- same model, data, parameters AND particles as the GPU run -- evaluated on the host cores
    through the CPU oracle (kind "port"): torch-CPU log-probs + oracle/alan_oracle.py contractions, all cores.
    Bounded to ~budget_s of CPU work."""
    from oracle import backend
    # torch-CPU elementwise ops stop scaling (and then regress) beyond a few dozen threads
    ncores = min(os.cpu_count() or 1, 32)
    t.set_num_threads(ncores)
    prob = build_problem("cpu")
    import alan_amd as alan
    sample = sample_on_cpu(gpu_sample, prob)
    with backend.installed():
        t0 = time.perf_counter()
        v = sample.elbo_nograd(alan.no_checkpoint)       # warm
        one = time.perf_counter() - t0
        n = max(2, min(30, int(budget_s / max(one, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            v = sample.elbo_nograd(alan.no_checkpoint)
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "ELBO evals/s", "cores": ncores, "kind": "port",
            "sample": f"{n} evals of the same movielens K={K} ELBO (same particles as the GPU run) on CPU "
                      "(torch-CPU log-probs + oracle reduce_Ks)",
            "elbo": float(v), "elbo_rel_diff_vs_gpu": abs(float(v) - gpu_elbo) / abs(gpu_elbo)}


def pmc_traffic(name, key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, collected in separate
    --pmc runs as MI355X_MICROARCH.md prescribes: profiles/<name>).  Counters cannot be read from inside this process;
    null when the committed profile does not cover this configuration."""
    try:
        for cand in (name.replace("r2_", "r4_"), name.replace("r2_", "r3_"), name):
            p = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(p):
                return json.load(open(p))[key]["traffic_bytes"]
    except Exception:
        pass
    return None


def fused_pmc_traffic(K, backward=False):
    """The same for the fused plate step at the S-ML sizes profiles/r3_fused_kernel_pmc.json covers (M=300, K=30 / 100)."""
    for name in ("r4_fused_kernel_pmc.json", "r3_fused_kernel_pmc.json", "r2_fused_kernel_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
            want = 4 * M_USERS * K ** 3
            hits = []
            for kname, rec in d.items():
                if ("bwd" in kname) == backward and rec["factor_bytes_never_materialised"] == want:
                    hits.append(rec)
            if hits:      # (the forward appears twice at K <= 32: the gradient-free launch -- the one an evaluation runs, its scale
                return min(hits, key=lambda r: r["write_bytes"])["traffic_bytes"]     # table ready-made -- writes no lse_out)
        except Exception:
            pass
    return None


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32) = the fp32 vector rate


def fused_roofline(records, tag, what, traffic=None):
    """roofline object of the fused plate step from KernelTimer records (native.MODE_FUSED_*): algorithmic FLOPs of
    the tile GEMM (2 * M * K^3 * (E + 1)) over the MFMA kernel's own duration."""
    sel = [(f, ms) for mode, f, ms in records if mode == tag]
    if not sel:
        return None
    flops = max(f for f, _ in sel)
    ts = [ms for f, ms in sel if f == flops]
    ms = sum(ts) / len(ts)
    tf = flops / ms / 1e9
    return {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
            "traffic": traffic, "kernel": what, "us_per_launch": ms * 1e3, "algorithmic_flops": flops,
            "launches_timed": len(ts),
            "note": "algorithmic fp32 flops of the tile GEMM against the chip's fp32 matrix peak (what an fp32 "
                    "implementation has).  The forward executes each fp32 product as six exact bf16 x bf16 products of 3-way "
                    "split operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error against fp64 equal to the fp32 "
                    "fma chain's, tools/mfma_bf16x3_probe.hip); it is bound by the vector unit (A-operand split, one exp "
                    "per element), not by the matrix cores.  The backward recomputes the tile on fp32 MFMA (whose results "
                    "the vector unit consumes without overlap, tools/mfma_f32_probe.hip) and takes its two 32-deep products "
                    "-- 32 of its 42 matrix steps -- on v_mfma_f32_32x32x16_bf16 with 2-way split operands (16-bit "
                    "mantissas: hi hi + hi lo + lo hi, fp32 accumulate; ALAN_NLB_X2=0 is the all-fp32 form)"}


def literal_hbm_reading(us_per_launch, K, M=None):
    """SURVEY 8(d) read literally: the S-ML algorithmic bytes -- the factor a materialising implementation streams,
    4 (M K^3 + M K + 2 K) -- over the fused launch's duration, against 8 TB/s."""
    M = M_USERS if M is None else M
    b = 4 * (M * K ** 3 + M * K + 2 * K)
    gbs = b / (us_per_launch * 1e-6) / 1e9
    return {"bound": "hbm", "algorithmic_bytes": b, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "note": "the fused kernel never reads these bytes (the factor is not materialised): the figure is the rate at "
                    "which it replaces a stream of them"}


def rows_roofline(K, M, iters=20, traffic_key=None):
    """The HBM-bound reduce_Ks kernel (rows.hip) on the S-ML plate step with the factor MATERIALISED --
    F[M,K,K,K] + g[M,K] -> lse K_z -> sum M -- the route every shape other than the fused one takes."""
    from alan_amd import engine as E
    from alan_amd.profiling import KernelTimer
    g = t.Generator(device="cuda").manual_seed(1234)
    F = -0.5 * t.randn(M, K, K, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    gz = -0.5 * t.randn(M, K, device="cuda", generator=g) ** 2 - 0.9189 - math.log(K)
    fac = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
    for _ in range(3):
        E.reduce_factors(fac, reduce=("z",), plate=("m",))
    t.cuda.synchronize()
    with KernelTimer(min_bytes=1 << 16) as kt:
        for _ in range(iters):
            E.reduce_factors(fac, reduce=("z",), plate=("m",))
        t.cuda.synchronize()
    res = kt.results()
    algo = 4 * (M * K ** 3 + M * K + K * K)
    ms = sum(m for _, _, m in res) / len(res)
    gbs = algo / ms / 1e6
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": pmc_traffic("r2_rows_kernel_pmc.json", traffic_key) if traffic_key else None,
            "kernel": "alan::rows_kernel", "us_per_launch": ms * 1e3, "algorithmic_bytes": algo,
            "workload": f"S-ML plate step on a materialised factor, K={K}, M={M}"}


def launch_mode(sample, use_graph):
    """How the timed evaluations were issued: what the captured evaluation of this sample actually replays through."""
    if not use_graph:
        return "eager (one Python-driven launch per kernel)"
    graphs = list(sample.__dict__.get("_graphs", {}).values())
    if graphs and all(getattr(g, "calls", None) is not None for g in graphs):
        return ("one captured ELBO evaluation, replayed by issuing its recorded library launches again from one C call "
                "(alan_calls_replay; the captured HIP graph owns the memory and is what proves the evaluation holds nothing else)")
    return "HIP graph replay of one captured ELBO evaluation"


def cpu_elbo_of(builder, gpu_sample):
    """The same ELBO (same particles) through the CPU oracle -- the parity check of a bench configuration."""
    from oracle import backend
    import alan_amd as alan
    prob = builder("cpu")
    sample = sample_on_cpu(gpu_sample, prob)
    with backend.installed():
        return float(sample.elbo_nograd(alan.no_checkpoint))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as CHILD processes -- one per GPU, RCCL
    rendezvous on 127.0.0.1 -- before this process has touched the GPU, relay rank 0's JSON line, exit with the
    launcher's code.  (Nothing is re-exec'ed: this parent never initialises HIP.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--no-extras", action="store_true", help="skip everything but the headline figure and its roofline")
    ap.add_argument("--c4-only", action="store_true", help="of the extras, only the C4 (movielens K=100) figure")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel from Python each step instead of replaying a HIP graph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))         # (before any torch.cuda call in this process)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not t.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # (rehearsal hook for a one-GPU box: ALAN_BENCH_REHEARSAL=1 puts every rank on cuda:0 and moves the partials with
    # gloo -- the control flow of an N-rank run without N GPUs; never set by the driver)
    rehearsal = os.environ.get("ALAN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    t.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=t.device("cuda", local))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr "
                         f"127.0.0.1 bench.py --gpus {args.gpus} ...` or plain `python bench.py --gpus {args.gpus}`")

    import alan_amd as alan
    from alan_amd import native
    from alan_amd import dist as adist
    from alan_amd.profiling import KernelTimer
    native.lib()                                   # fail loudly if the HIP library is missing

    K = args.K
    prob = build_problem("cuda")
    sample = draw(prob, K)
    strat = strategy_for(world, K)
    use_graph = not args.eager and os.environ.get("ALAN_BENCH_GRAPH", "1") != "0" and not (rehearsal and world > 1)
    if use_graph:
        try:
            dt, elbo = timed_evals(sample, strat, args.steps, args.warmup, world, graph=True)
        except Exception as e:                           # capture unsupported: eager
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
    # ---- the headline: independent evaluations OVERLAPPED (one GPU, library launches alone); the one-after-another figure
    # stays beside it as `single_stream`
    PIPE_LANES = int(os.environ.get("ALAN_BENCH_LANES", "4"))
    single = None
    pipelined = None
    if use_graph and world == 1 and PIPE_LANES > 1:
        try:
            with t.no_grad():
                eager_val = float(sample.elbo_nograd(strat, graph=False))
            dt_p, elbo_p, worst, dt_long = timed_pipeline(sample, strat, args.steps, args.warmup, PIPE_LANES, eager_val)
            single = {"evals_per_s": args.steps / dt, "us_per_eval": dt / args.steps * 1e6, "launch": launch_mode(sample, use_graph)}
            pipelined = {"n_streams": PIPE_LANES, "issuing_threads": PIPE_LANES,
                         "max_rel_diff_vs_eager": worst, "steady_state_us_per_eval": dt_long * 1e6,
                         "steady_state_evals_per_s": 1.0 / dt_long}
            dt, elbo = dt_p, elbo_p
        except Exception as e:
            print(f"[bench] pipelined evaluations unavailable ({type(e).__name__}: {e}); one after another", file=sys.stderr)
            single = pipelined = None
    kt = KernelTimer(min_bytes=1 << 20)
    if use_graph:
        # per-kernel HIP events cannot be recorded inside a replayed graph: time the same launches
        # eagerly, same process, same buffers, right after the timed region
        with kt:
            for _ in range(args.steps):
                sample.elbo_nograd(strat)
            t.cuda.synchronize()
    else:
        dt, elbo = timed_evals(sample, strat, args.steps, args.warmup, world, timer=kt)
    res = kt.results()

    out = {
        "metric": "ELBO evals/sec at K=30 (movielens)" if K == 30 else f"ELBO evals/sec at K={K} (movielens)",
        "value": args.steps / dt, "unit": "ELBO evals/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"movielens M={M_USERS} N={N_FILMS} d_z={D_Z}, K={K}, elbo_nograd on a fixed sample",
                   "launch": (f"{pipelined['n_streams']} independent evaluations in flight (sample.EvalPipeline / alan_pipeline_*: "
                              f"{pipelined['n_streams']} recorded copies of the evaluation, each re-issued on a stream of its "
                              "own by a library thread; every result checked against the eager value); one evaluation after "
                              "another: `single_stream`") if pipelined else launch_mode(sample, use_graph),
                   "n_streams": pipelined["n_streams"] if pipelined else 1,
                   "plate_step": "fused (alan_normal_lse: producer + log-sum-exp + plate sum in one launch)"
                   if adist.FUSE_PLATE_STEP else "materialised factor (producer kernel + rows kernel)",
                   "computation_strategy": type(strat).__name__ +
                   (f"('plate_1', {strat.split_size}, shard=True)" if world > 1 else ""),
                   "parallelism": f"plate_1 sharded over {world} rank(s), one all-reduce(SUM) of [K,K] per eval"
                   if world > 1 else "single GPU"},
        "elbo": elbo,
    }
    if pipelined:
        out["pipelined"] = pipelined
        out["single_stream"] = single
    # ---- roofline of the DOMINANT kernel of the evaluation, timed live with HIP events handed to its launch
    rf = fused_roofline(res, native.MODE_FUSED_FWD,
                        "alan::normal_lse_x3_kernel (plate_1 step, the factor F[M,K,K,K] never materialised: per "
                        "(m, K_mu) a 32x32 tile of log-probs on v_mfma_f32_32x32x16_bf16 with 3-way split fp32 operands, "
                        "log-sum-exp over K_z down the accumulator registers, plate sum in a register)",
                        traffic=fused_pmc_traffic(K) if world == 1 else None)
    if rf is not None:
        rf["literal_hbm"] = literal_hbm_reading(rf["us_per_launch"], K)
        out["roofline"] = rf
    else:                                                # FUSE_PLATE_STEP off: the HBM-bound rows kernel dominates
        res_lse = [(b, m) for mode, b, m in res if mode == native.MODE_LSE]
        if res_lse:
            big = max(b for b, _ in res_lse)
            sel = [m for b, m in res_lse if b == big]
            ms = sum(sel) / len(sel)
            out["roofline"] = {"bound": "hbm", "achieved": big / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": big / ms / 1e6 / HBM_PEAK_GBS,
                               "traffic": pmc_traffic("r2_rows_kernel_pmc.json", "literal_K30_M300"),
                               "kernel": "alan::rows_kernel", "us_per_launch": ms * 1e3, "algorithmic_bytes": big,
                               "launches_timed": len(sel)}
    if not args.no_extras:
        # BASELINE config C4 in the same run, under the same key at every N: movielens K=100,
        # Split('plate_1', 38) -- one GPU evaluates its chunks as one slice; sharded over the ranks (one all-reduce) at N > 1
        s100 = draw(prob, 100)
        st100 = strategy_for(world, 100)
        try:
            d100, v100 = timed_evals(s100, st100, 10, 2, world, graph=use_graph)
            out["c4_movielens_K100"] = {"evals_per_s": 10 / d100, "ms_per_eval": d100 / 10 * 1e3, "elbo": v100,
                                        "n_gpus": world,
                                        "strategy": f"Split('plate_1', {st100.split_size}" +
                                                    (", shard=True)" if world > 1 else ")"),
                                        # (what this rank actually evaluated: split.MERGE_CHUNKS makes a rank's block of
                                        # chunks one slice unless that would break Split's memory bound)
                                        "merged": st100.merging() and len(getattr(st100, "last_sizes", [])) == max(1, world),
                                        "effective_chunk_sizes": getattr(st100, "last_sizes", None)}
            if world == 1 and use_graph:
                # independent evaluations overlapped (alan_pipeline_*), as the headline
                out["c4_movielens_K100"]["pipelined"] = pipelined_figure(s100, st100, 200, 3, d100 / 10 * 1e6)
            with KernelTimer() as kt100:
                for _ in range(5):
                    s100.elbo_nograd(st100)
                t.cuda.synchronize()
            rf100 = fused_roofline(kt100.results(), native.MODE_FUSED_FWD, "alan::normal_lse_x3_kernel at K=100",
                                   traffic=fused_pmc_traffic(100) if world == 1 else None)
            if rf100 is not None:
                rf100["literal_hbm"] = literal_hbm_reading(rf100["us_per_launch"], 100)
                out["c4_movielens_K100"]["roofline"] = rf100
            if world > 1:
                # the same evaluation unsharded, on this very GPU, in this very run (every rank does it: no collective
                # inside): the N = 1 figure the sharded one is to be read against
                st1 = alan.Split("plate_1", st100.split_size)
                d1, v1 = timed_evals(s100, st1, 10, 2, 1, graph=use_graph)
                out["c4_movielens_K100"]["n1_same_run"] = {"evals_per_s": 10 / d1, "ms_per_eval": d1 / 10 * 1e3, "elbo": v1}
                out["c4_movielens_K100"]["speedup_vs_n1_same_run"] = d1 / d100
                out["c4_movielens_K100"]["collective"] = ("gloo all_reduce (REHEARSAL: every rank on one GPU)" if rehearsal else
                                                           "RCCL all_reduce(SUM) of the [K, K] partial (the default)")
                # the same sharded evaluation with the library's own one-shot exchange as its collective
                # (split.ONE_SHOT_EXCHANGE: every rank writes its partial into every peer's inbox over xGMI and adds what
                # arrived -- one library launch, so the evaluation is re-issued from its launch list instead of replayed as a
                # graph), in this very run: both collectives read against the same n1_same_run
                if True:
                    from alan_amd import split as asplit
                    saved = asplit.ONE_SHOT_EXCHANGE
                    asplit.ONE_SHOT_EXCHANGE = True
                    try:
                        s100x = draw(prob, 100)
                        s100x.elbo_nograd(st100, graph=False)                 # (sets the exchange up: outside any capture)
                        dx, vx = timed_evals(s100x, st100, 10, 2, world, graph=use_graph)
                        import torch.distributed as dist_
                        done, bad = asplit.exchange_for(st100.group).status()
                        out["c4_movielens_K100"]["one_shot_exchange"] = {
                            "evals_per_s": 10 / dx, "ms_per_eval": dx / 10 * 1e3, "elbo": vx,
                            "speedup_vs_n1_same_run": d1 / dx, "against_rccl": d100 / dx,
                            "exchanges_completed": done, "first_failed_exchange": bad,
                            "launch": launch_mode(s100x, use_graph),
                            "elbo_rel_diff_vs_rccl": abs(vx - v100) / abs(v100)}
                        if use_graph:
                            # and the rank's sharded evaluations overlapped (an exchange per lane): every rank issues the same batch
                            pfx = pipelined_figure(s100x, st100, 120, 3, dx / 10 * 1e6)
                            if "error" not in pfx:
                                pfx["speedup_vs_n1_same_run"] = d1 / 10 * 1e6 / pfx["us_per_eval"]
                            out["c4_movielens_K100"]["one_shot_exchange"]["pipelined"] = pfx
                        del s100x
                    except Exception as e:
                        out["c4_movielens_K100"]["one_shot_exchange"] = {"error": f"{type(e).__name__}: {e}"}
                    finally:
                        asplit.ONE_SHOT_EXCHANGE = saved
        except Exception as e:
            out["c4_movielens_K100"] = {"error": f"{type(e).__name__}: {e}"}
        del s100
        t.cuda.empty_cache()
        if world == 1 and not args.c4_only and "error" not in out["c4_movielens_K100"]:
            # what ONE rank of an N-GPU C4 run computes, timed on this GPU without the collective: the same K=100 ELBO over
            # the rank's ceil(300 / N) users (tools/rank_share.py) -- the compute side of the 1 -> 8 GPU scaling figure
            share = {}
            try:
                for n in (2, 4, 8):
                    pn = build_problem("cuda", M=-(-M_USERS // n))
                    sn = draw(pn, 100)
                    dn, _ = timed_evals(sn, alan.no_checkpoint, 10, 2, 1, graph=use_graph)
                    share[f"N{n}"] = {"users": -(-M_USERS // n), "us_per_eval": dn / 10 * 1e6,
                                      # (what the rank's compute alone allows from 1 to N GPUs: the all-reduce of the
                                      # [K, K] partial, 40 KB, comes on top)
                                      "projected_speedup_before_collective":
                                          out["c4_movielens_K100"]["ms_per_eval"] * 1e3 / (dn / 10 * 1e6)}
                    if use_graph:
                        # the rank's evaluations overlapped: THROUGHPUT per evaluation of one rank's share, and what it
                        # allows against the one-GPU evaluation -- one after another, and overlapped as well
                        pf = pipelined_figure(sn, alan.no_checkpoint, 400, 4, dn / 10 * 1e6)
                        if "error" not in pf:
                            pf["projected_speedup_before_collective"] = out["c4_movielens_K100"]["ms_per_eval"] * 1e3 / pf["us_per_eval"]
                            p1 = out["c4_movielens_K100"].get("pipelined", {})
                            if "us_per_eval" in p1:
                                pf["projected_speedup_before_collective_vs_pipelined_n1"] = p1["us_per_eval"] / pf["us_per_eval"]
                        share[f"N{n}"]["pipelined"] = pf
                    del pn, sn
                out["c4_movielens_K100"]["rank_share_no_collective"] = share
            except Exception as e:
                out["c4_movielens_K100"]["rank_share_no_collective"] = {"error": f"{type(e).__name__}: {e}"}
            t.cuda.empty_cache()
        if world > 1:
            # the same plate step at a fixed 300 users PER GPU (M = 300 * N): what sharding buys when the plate grows
            # with the machine -- per-GPU work constant, still one all-reduce of [K,K] per evaluation
            try:
                pw = build_problem("cuda", M=M_USERS * world)
                sw = draw(pw, K)
                stw = alan.Split("plate_1", M_USERS, shard=True)
                dw, vw = timed_evals(sw, stw, args.steps, args.warmup, world, graph=use_graph)
                out["weak_scaling_300_users_per_gpu"] = {"evals_per_s": args.steps / dw, "us_per_eval": dw / args.steps * 1e6,
                                                         "users": M_USERS * world, "elbo": vw, "n_gpus": world}
                del sw, pw
            except Exception as e:
                out["weak_scaling_300_users_per_gpu"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_extras and not args.c4_only:
        # ---- the reduce_Ks kernel proper (rows.hip): what streams a MATERIALISED factor -- literal size and bandwidth regime
        out["roofline_rows"] = {"literal": rows_roofline(K, M_USERS, traffic_key="literal_K30_M300" if K == 30 else None),
                                "scaled": rows_roofline(K, M_USERS * 64, traffic_key="scaled_K30_M19200" if K == 30 else None)}
        # ---- the same evaluation with the factor materialised (dist.FUSE_PLATE_STEP = False)
        adist.FUSE_PLATE_STEP = False
        try:
            sf = draw(prob, K)
            d_f, v_f = timed_evals(sf, strat, args.steps, args.warmup, world, graph=use_graph)
            out["materialised_route"] = {"evals_per_s": args.steps / d_f, "us_per_eval": d_f / args.steps * 1e6,
                                         "elbo": v_f, "default": False,
                                         "what": "dist.FUSE_PLATE_STEP = False: producer kernel writes F, rows kernel reads it"}
        finally:
            adist.FUSE_PLATE_STEP = True
        # ---- the reference-signature call: sample.elbo_nograd() with no graph argument
        for _ in range(5):
            sample.elbo_nograd(strat)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            sample.elbo_nograd(strat)
        t.cuda.synchronize()
        out["plain_call"] = {"us_per_eval": (time.perf_counter() - t0) / 50 * 1e6,
                             "what": "sample.elbo_nograd(strategy), as the reference spells it"}
        # sample() + elbo per iteration, as examples/basic_runner.py:86-97 of the reference counts it (eager)
        for _ in range(3):
            prob.sample(K, reparam=False).elbo_nograd(strat)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            prob.sample(K, reparam=False).elbo_nograd(strat)
        t.cuda.synchronize()
        dt_se = (time.perf_counter() - t0) / 20
        out["sample_plus_elbo"] = {"ms_per_iter": dt_se * 1e3, "iters_per_s": 1 / dt_se,
                                   "launch": "eager (a fresh sample every iteration)"}
        try:                                      # the same iteration captured once (alan_amd.GraphedEval)
            ev = alan.GraphedEval(prob, K, strat)
            for _ in range(3):
                ev()
            t.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                v_se = ev()
            t.cuda.synchronize()
            dt_g = (time.perf_counter() - t0) / 100
            out["sample_plus_elbo"].update({"graph_ms_per_iter": dt_g * 1e3, "graph_iters_per_s": 1 / dt_g,
                                            "graph_last_elbo": float(v_se)})
        except Exception as e:
            out["sample_plus_elbo"]["graph_error"] = f"{type(e).__name__}: {e}"
        try:                                      # eight such iterations per captured graph (each with its own fresh particles)
            ev8 = alan.GraphedEval(prob, K, strat, unroll=8)
            for _ in range(3):
                ev8()
            t.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(25):
                v8 = ev8()
            t.cuda.synchronize()
            dt_8 = (time.perf_counter() - t0) / (25 * 8)
            out["sample_plus_elbo"].update({"graph_unroll8_ms_per_iter": dt_8 * 1e3, "graph_unroll8_iters_per_s": 1 / dt_8,
                                            "graph_unroll8_distinct_elbos": len(set(v8.tolist()))})
            del ev8
        except Exception as e:
            out["sample_plus_elbo"]["graph_unroll8_error"] = f"{type(e).__name__}: {e}"
        try:                                      # the same iterations overlapped (alan_amd.SamplingPipeline: a generator state per lane)
            sp_ = alan.SamplingPipeline(prob, K, strat, lanes=4, results=1024)
            sp_.run(64)
            t.cuda.synchronize()
            t0 = time.perf_counter()
            v_sp = sp_.run(2000)
            t.cuda.synchronize()
            dt_sp = (time.perf_counter() - t0) / 2000
            out["sample_plus_elbo"].update({"pipelined_ms_per_iter": dt_sp * 1e3, "pipelined_iters_per_s": 1 / dt_sp, "pipelined_n_streams": 4,
                                            "pipelined_distinct_elbos_of_2000": len(set(v_sp.tolist())),
                                            "pipelined_mean_elbo": float(v_sp.mean())})
            sp_.close()
            del sp_
        except Exception as e:
            out["sample_plus_elbo"]["pipelined_error"] = f"{type(e).__name__}: {e}"
        # row a10 (the path's backward) in production use: a whole training iteration -- sample -> elbo_vi | elbo_rws ->
        # backward -> Adam, the loop of examples/basic_runner.py:81-112 -- captured once and replayed
        try:
            tr = {}
            for mode in ("vi", "rws"):
                # the optimiser step as the library's own launch (alan_amd.Adam, bitwise torch's fused capturable Adam):
                # the iteration is then library launches alone and is re-issued from its recorded launch list
                p_tr = build_problem("cuda")
                params = list(p_tr.parameters()) if mode == "vi" else list(p_tr.Q.parameters())
                opt = alan.Adam(params, lr=1e-2, maximize=(mode == "rws"))
                step = alan.GraphedStep(p_tr, K, opt, method=mode)
                for _ in range(5):
                    step()
                t.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    v_tr = step()
                t.cuda.synchronize()
                tr[mode] = {"ms_per_iter": (time.perf_counter() - t0) / 50 * 1e3, "last_elbo": float(v_tr),
                            "optimizer": "alan_amd.Adam (alan_adam_step: one launch, step count on the device)",
                            "launch": "recorded launch list (library launches alone)" if step.calls is not None
                            else "HIP graph replay (the iteration holds kernels that are not the library's)",
                            "library_launches_per_iteration": step.calls.launches() if step.calls is not None else None}
                del step, opt, p_tr
                # the same with torch's optimiser (fused=True: one multi-tensor kernel per Adam step + the counters' increment):
                # a HIP graph replay, as until round 3
                p_tr = build_problem("cuda")
                params = list(p_tr.parameters()) if mode == "vi" else list(p_tr.Q.parameters())
                opt = t.optim.Adam(params, lr=1e-2, capturable=True, fused=True, maximize=(mode == "rws"))
                step = alan.GraphedStep(p_tr, K, opt, method=mode)
                for _ in range(5):
                    step()
                t.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    v_tr = step()
                t.cuda.synchronize()
                tr[mode]["torch_adam_graph_replay_ms_per_iter"] = (time.perf_counter() - t0) / 50 * 1e3
                del step, opt, p_tr
                # four consecutive iterations per captured graph (GraphedStep(unroll=4))
                p_tr = build_problem("cuda")
                params = list(p_tr.parameters()) if mode == "vi" else list(p_tr.Q.parameters())
                opt = alan.Adam(params, lr=1e-2, maximize=(mode == "rws"))
                step = alan.GraphedStep(p_tr, K, opt, method=mode, unroll=4)
                for _ in range(2):
                    step()
                t.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(13):
                    v_tr = step()
                t.cuda.synchronize()
                tr[mode]["unroll4_ms_per_iter"] = (time.perf_counter() - t0) / (13 * 4) * 1e3
                tr[mode]["unroll4_last_elbos"] = v_tr.tolist()
                del step, opt, p_tr
            # the fused plate step's backward alone (elbo_vi, eager, HIP events on its launch)
            p_tr = build_problem("cuda")
            with KernelTimer() as ktb:
                for _ in range(5):
                    for q in p_tr.parameters():
                        q.grad = None
                    p_tr.sample(K, reparam=True).elbo_vi(alan.no_checkpoint).backward()
                t.cuda.synchronize()
            rfb = fused_roofline(ktb.results(), native.MODE_FUSED_BWD, "alan::normal_lse_bwd_kernel (every gradient of the "
                                 "plate step in one pass: D recomputed in fp32, V and U products as bf16x2 on the matrix cores)",
                                 traffic=fused_pmc_traffic(K, backward=True))
            if rfb is not None:                          # (flops from the shape: native.run_normal_lse_backward)
                tr["fused_backward_kernel"] = rfb
            del p_tr
            # the same at K = 100 under C4's strategy (Split('plate_1', 38): the rank's chunks as one slice)
            try:
                p_tr = build_problem("cuda")
                opt = alan.Adam(list(p_tr.parameters()), lr=1e-2)
                step = alan.GraphedStep(p_tr, 100, opt, method="vi", computation_strategy=alan.Split("plate_1", 38))
                for _ in range(3):
                    step()
                t.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    v_tr = step()
                t.cuda.synchronize()
                tr["vi_K100_split38"] = {"ms_per_iter": (time.perf_counter() - t0) / 10 * 1e3, "last_elbo": float(v_tr)}
                del step, opt
                with KernelTimer() as ktb100:
                    for _ in range(3):
                        for q in p_tr.parameters():
                            q.grad = None
                        p_tr.sample(100, reparam=True).elbo_vi(alan.Split("plate_1", 38)).backward()
                    t.cuda.synchronize()
                rfb100 = fused_roofline(ktb100.results(), native.MODE_FUSED_BWD, "alan::normal_lse_bwd_kernel at K=100 "
                                        "(flat row tiling)", traffic=fused_pmc_traffic(100, backward=True))
                if rfb100 is not None:
                    tr["vi_K100_split38"]["fused_backward_kernel"] = rfb100
                del p_tr
            except Exception as e:
                tr["vi_K100_split38"] = {"error": f"{type(e).__name__}: {e}"}
            out["training_iteration"] = tr
        except Exception as e:
            out["training_iteration"] = {"error": f"{type(e).__name__}: {e}"}
        out["cpu_baseline"] = cpu_baseline(K, sample, elbo)
        sweep = {}
        for k2 in (3, 10, 100):
            s2 = draw(prob, k2)
            st2 = strategy_for(world, k2)
            n2 = 5 if k2 >= 100 else 20
            d2, _ = timed_evals(s2, st2, n2, 2, world, graph=use_graph)
            sweep[f"K{k2}"] = {"evals_per_s": n2 / d2, "ms_per_eval": d2 / n2 * 1e3,
                               "strategy": type(st2).__name__}
            del s2
            t.cuda.empty_cache()
        out["sweep"] = sweep
        # ---- S-BUS (C3) and S-TS (C5) at K in {3, 10, 30, 100}; K=30 checked against the CPU oracle
        others = {}
        for name, builder, algo in (
                ("bus_breakdown Y=2 B=3 I=150 (3 nested plates)", build_bus_problem,
                 lambda k: 4 * (k * k * 900 + 6 * k * k + 4 * k * k + 2 * k)),
                ("timeseries Kalman T=1000", build_timeseries_problem, lambda k: 4 * 1000 * k * k)):
            p2 = builder("cuda")
            per_k = {}
            for k2 in (3, 10, 30, 100):
                s2 = draw(p2, k2)
                n_g = 50 if k2 <= 30 else 10
                d_g, v_g = timed_evals(s2, alan.no_checkpoint, n_g, 3, world, graph=True)
                us = d_g / n_g * 1e6
                rec = {"evals_per_s": n_g / d_g, "us_per_eval": us, "elbo": v_g, "algorithmic_bytes": algo(k2),
                       "hbm_floor_us": algo(k2) / HBM_PEAK_GBS / 1e3}
                if k2 in (30, 100) and world == 1:
                    # independent evaluations overlapped (alan_pipeline_*), every result checked against the eager value
                    rec["pipelined"] = pipelined_figure(s2, alan.no_checkpoint, 400 if k2 == 30 else 100, 4, us)
                if k2 == 30:
                    d_e, _ = timed_evals(s2, alan.no_checkpoint, 20, 3, world)          # graph=False: kernel by kernel
                    rec["evals_per_s_eager"] = 20 / d_e
                    try:
                        v_cpu = cpu_elbo_of(builder, s2)
                        rec["elbo_cpu_oracle"] = v_cpu
                        rec["elbo_rel_diff_vs_cpu"] = abs(v_cpu - v_g) / abs(v_cpu)
                    except Exception as e:
                        rec["elbo_cpu_oracle_error"] = f"{type(e).__name__}: {e}"
                per_k[f"K{k2}"] = rec
                del s2
            others[name] = per_k
        out["other_configs"] = others
        # ---- the timeseries chain with its backward (row a10 for C5's shape): forward alone and forward + backward of
        # logsumexp(chain_logmmexp(ms), -1) on a [T=1000, K, K] factor, each as a replayed graph
        chain = {}
        try:
            from alan_amd.contract import chain_logmmexp_lse
            for k2 in (30, 100):
                ms = (-0.5 * t.randn(1000, k2, k2, device="cuda") ** 2 - 0.92 - math.log(k2)).requires_grad_(True)
                gv = t.rand(k2, device="cuda")
                rec = {}
                for what, fn in (("forward_us", lambda: chain_logmmexp_lse(ms.detach())),
                                 ("forward_backward_us", lambda: t.autograd.grad(chain_logmmexp_lse(ms), ms, gv))):
                    side = t.cuda.Stream()
                    side.wait_stream(t.cuda.current_stream())
                    with t.cuda.stream(side):
                        for _ in range(3):
                            fn()
                    t.cuda.current_stream().wait_stream(side)
                    t.cuda.synchronize()
                    gr = t.cuda.CUDAGraph()
                    with t.cuda.graph(gr, stream=side):
                        keep_ = fn()
                    for _ in range(3):
                        gr.replay()
                    t.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        gr.replay()
                    t.cuda.synchronize()
                    rec[what] = (time.perf_counter() - t0) / 20 * 1e6
                    del gr, keep_
                rec["backward_us"] = rec["forward_backward_us"] - rec["forward_us"]
                chain[f"K{k2}"] = rec
                del ms
        except Exception as e:
            chain["error"] = f"{type(e).__name__}: {e}"
        out["timeseries_chain_T1000"] = chain
        # ---- a training iteration of the same timeseries model (learned Normal posterior per timestep), one replayed graph
        ts_tr = {}
        try:
            for mode in ("vi", "rws"):
                p_ts = build_timeseries_train_problem("cuda")
                opt = alan.Adam(list(p_ts.parameters()), lr=1e-2, maximize=(mode == "rws"))
                step = alan.GraphedStep(p_ts, 30, opt, method=mode)
                for _ in range(5):
                    step()
                t.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    v_ts = step()
                t.cuda.synchronize()
                ts_tr[mode] = {"ms_per_iter": (time.perf_counter() - t0) / 30 * 1e3, "last_elbo": float(v_ts)}
                del step, opt, p_ts
            ts_tr["config"] = "Kalman T=1000, K=30, OptParam location and log-scale per timestep, alan_amd.Adam"
        except Exception as e:
            ts_tr["error"] = f"{type(e).__name__}: {e}"
        out["timeseries_training_iteration"] = ts_tr
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
