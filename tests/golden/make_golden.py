#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by RUNNING THE REFERENCE
(/root/reference/src, imported read-only) in the build container.  Inert elsewhere:
it exits with a message when /root/reference is absent (e.g. on the GPU box).

    python tests/golden/make_golden.py

What is captured:
  lse_dims.pt        logsumexp_dims / logmeanexp_dims      (reference utils.py:207-225)
  seam_synthetic.pt  reduce_Ks on seeded synthetic factors (reduce_Ks.py:236-244) + its
                     autograd grads + an order-free fp64 brute force; edge cases
  seam_recorded.pt   every reduce_Ks call made by elbo_nograd on real reference models
                     (linear_gaussian, linear_gaussian_latents, model1, movielens, bus_breakdown)
  chain.pt           logmmexp / chain_logmmexp            (utils.py:478-510)
  chain_peaked.pt    the same on sharply peaked transition matrices, where the +eps inside the log (utils.py:506)
                     floors most entries: results then depend on the tree's bracketing, and autograd differentiates
                     through the floor.  Includes batches of chains (torchdim batch dims, logpq.py:133-139).
  e2e_*.pt           sample tree + data + params + ELBO under no_checkpoint / checkpoint / Split
  posterior.pt       the tables the reference's posterior K sampling hands to t.multinomial: per step of sample_Ks
                     (reduce_Ks.py:35-83) and per timestep of sample_Ks_timeseries (reduce_Ks.py:85-232)
  e2e_wide_group.pt  a Group of seven Normal latents (more factors on one K than one launch takes)
  e2e_movielens_K30.pt, e2e_bus_breakdown_K30.pt, e2e_movielens_K100_split38.pt
                     the same at the BASELINE.json sizes (C2, C3, C4): `make_golden.py baseline_sizes`

This-container-only accommodations (none touches the hot path's arithmetic):
  * ``opt_einsum`` stand-in (see _planner_standin/): elimination ORDER only.
  * torch-2.10 drift on the *sampling* side: td.Uniform.arg_constraints became an instance
    property (breaks TorchDimDist.py:47 via Sampler.py:147) and zero-arg Tensor.expand()
    raises for plate-free scalar params (BoundPlate.py:30).  Both patched at run time here.
No reference source is copied: fixtures are tensors, dim names and scalars only.
"""
import os
import sys
import importlib
import warnings

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

if not os.path.isdir(os.path.join(REF, "src", "alan")):
    print("reference not present; nothing to do")
    sys.exit(0)

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.join(HERE, "_planner_standin"))
sys.path.insert(0, os.path.join(REF, "src"))

import torch as t                                  # noqa: E402
import torch.distributions as td                   # noqa: E402
from torch.distributions import constraints        # noqa: E402
from functorch.dim import Dim                      # noqa: E402

td.Uniform.arg_constraints = {
    "low": constraints.dependent(is_discrete=False, event_dim=0),
    "high": constraints.dependent(is_discrete=False, event_dim=0),
}

import alan                                        # noqa: E402
import alan.logpq as ref_logpq                     # noqa: E402
from alan import utils as ref_utils                # noqa: E402
from alan.reduce_Ks import reduce_Ks as ref_reduce_Ks  # noqa: E402

_bp = sys.modules["alan.BoundPlate"]
_orig_expand_named = _bp.expand_named


def _expand_named(x, names, all_platesizes):
    if len(names) == 0 and x.ndim == 0:
        return x
    return _orig_expand_named(x, names, all_platesizes)


_bp.expand_named = _expand_named


# ------------------------------------------------------------------ conversions
def undim(x):
    """functorch.dim tensor -> (contiguous positional tensor, names) with the dims listed in
    STORAGE order (outermost first), so the fixture keeps the layout the reference produced."""
    if not ref_utils.is_dimtensor(x) and not isinstance(x, t.Tensor):
        x = t.as_tensor(x)
    dims = list(ref_utils.generic_dims(x))
    if not dims:
        return x.detach().clone(), ()
    pos = x.order(*dims)
    strides = pos.stride()[: len(dims)]
    order = sorted(range(len(dims)), key=lambda i: (-strides[i], i))
    dims = [dims[i] for i in order]
    pos = x.order(*dims).detach().contiguous().clone()
    return pos, tuple(str(d) for d in dims)


def todim(x, names, dimmap):
    ds = []
    for n, s in zip(names, x.shape[: len(names)]):
        if n not in dimmap:
            dimmap[n] = Dim(n, s)
        ds.append(dimmap[n])
    return x[ds] if ds else x


def tree_undim(tree):
    out = {}
    for k, v in tree.items():
        out[k] = tree_undim(v) if isinstance(v, dict) else undim(v)
    return out


def save(name, obj):
    path = os.path.join(HERE, name)
    t.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def loglike(gen, *shape, dtype=t.float32, K=1):
    """values distributed like unit-Gaussian log densities."""
    return (-0.5 * t.randn(*shape, generator=gen, dtype=dtype) ** 2 - 0.9189 - float(t.log(t.tensor(float(K)))))


# ------------------------------------------------------------------ 1. logsumexp_dims
def gen_lse_dims():
    g = t.Generator().manual_seed(20241)
    cases = []
    specs = [
        ("last", (7, 5), ("a", "b"), ("b",), t.float32),
        ("first", (7, 5), ("a", "b"), ("a",), t.float32),
        ("both", (7, 5), ("a", "b"), ("a", "b"), t.float32),
        ("mid3", (4, 30, 6), ("p", "k", "q"), ("k",), t.float32),
        ("two_of_four", (3, 10, 4, 10), ("p", "k1", "q", "k2"), ("k2", "k1"), t.float32),
        ("f64", (5, 9), ("a", "b"), ("b",), t.float64),
        ("K100", (6, 100), ("p", "k"), ("k",), t.float32),
        ("empty_dims", (4, 3), ("a", "b"), (), t.float32),
        ("wide_range", (8, 33), ("a", "k"), ("k",), t.float32),
    ]
    for name, shape, names, red, dtype in specs:
        x = loglike(g, *shape, dtype=dtype)
        if name == "wide_range":
            x = x * 40.0
        dm = {}
        xd = todim(x, names, dm)
        out = ref_utils.logsumexp_dims(xd, tuple(dm[r] for r in red))
        o, on = undim(out)
        case = dict(name=name, x=x, names=names, reduce=red, out=o, out_names=on)
        if red:
            m, mn = undim(ref_utils.logmeanexp_dims(xd, tuple(dm[r] for r in red)))
            case.update(mean_out=m, mean_out_names=mn)
        cases.append(case)
    # -inf handling (reference: single -inf fine, all -inf slice -> NaN)
    x = loglike(g, 4, 6)
    x[1, 2] = float("-inf")
    x[3, :] = float("-inf")
    dm = {}
    out = ref_utils.logsumexp_dims(todim(x, ("a", "k"), dm), (dm["k"],))
    o, on = undim(out)
    cases.append(dict(name="neg_inf", x=x, names=("a", "k"), reduce=("k",), out=o, out_names=on))
    save("lse_dims.pt", cases)


# ------------------------------------------------------------------ 2. synthetic seam
def run_seam(factors, Ks, want_grad=True):
    dm = {}
    leaves = [x.clone().requires_grad_(want_grad and x.is_floating_point()) for x, _ in factors]
    lps = [todim(x, n, dm) for x, (_, n) in zip(leaves, factors)]
    out = ref_reduce_Ks(lps, [dm[k] for k in Ks])
    o, on = undim(out)
    res = dict(factors=[(x, tuple(n)) for x, n in factors], Ks=tuple(Ks), out=o, out_names=on)
    if want_grad:
        pos = out.order(*ref_utils.generic_dims(out)) if ref_utils.generic_dims(out) else out
        gs = t.autograd.grad(pos.sum(), leaves, allow_unused=True, retain_graph=True)
        res["grads_of_sum"] = [None if g is None else g.detach() for g in gs]
        gw = t.Generator().manual_seed(99)
        # weighted upstream grad, aligned with (out, out_names)
        w = t.randn(o.shape, generator=gw, dtype=o.dtype)
        pos_named = out.order(*[dm[n] for n in on]) if on else out
        gs = t.autograd.grad((pos_named * w).sum(), leaves, allow_unused=True)
        res["grad_out"] = w
        res["grads_weighted"] = [None if g is None else g.detach() for g in gs]
    # order-free brute force in fp64
    names = []
    for _, n in factors:
        for d in n:
            if d not in names:
                names.append(d)
    tot = 0
    for x, n in factors:
        perm = [n.index(d) for d in names if d in n]
        tot = tot + x.double().permute(perm)[tuple(slice(None) if d in n else None for d in names)]
    axes = [names.index(k) for k in Ks]
    if axes:
        tot = t.logsumexp(tot, axes)
    res["brute_f64"] = tot
    res["brute_names"] = tuple(d for d in names if d not in Ks)
    return res


def gen_seam_synthetic():
    cases = []

    def add(name, factors, Ks, **kw):
        c = run_seam(factors, Ks, **kw)
        c["name"] = name
        cases.append(c)

    # the shapes of the reference's (stale) test_ad_hoc/test_reduce_Ks.py:29-31, seed 127
    g = t.Generator().manual_seed(127)
    add("adhoc_2345",
        [(t.randn(2, 3, 4, 5, generator=g), ("p", "K1", "K2", "K3")),
         (t.randn(2, 3, 5, generator=g), ("p", "K1", "K3")),
         (t.randn(2, 4, 5, generator=g), ("p", "K2", "K3"))],
        ("K1", "K2", "K3"))
    add("adhoc_partial",
        [(t.randn(2, 3, 4, 5, generator=g), ("p", "K1", "K2", "K3")),
         (t.randn(2, 3, 5, generator=g), ("p", "K1", "K3")),
         (t.randn(2, 4, 5, generator=g), ("p", "K2", "K3"))],
        ("K2",))

    for cfg, K in enumerate((3, 10)):
        g = t.Generator().manual_seed(1234 + cfg)
        M = 12
        # S-ML (movielens-shaped): F[M,Ka,Kb,Kz] + g[M,Kz] -> lse Kz
        add(f"S-ML_plate_K{K}",
            [(loglike(g, M, K, K, K, K=K), ("plate_1", "K_mu", "K_psi", "K_z")),
             (loglike(g, M, K, K=K), ("plate_1", "K_z"))],
            ("K_z",))
        add(f"S-ML_top_K{K}",
            [(loglike(g, K, K=K), ("K_mu",)), (loglike(g, K, K=K), ("K_psi",)),
             (loglike(g, K, K) * M, ("K_mu", "K_psi"))],
            ("K_mu", "K_psi"))
        # S-BUS (bus_breakdown-shaped)
        Y, B = 2, 3
        add(f"S-BUS_borough_K{K}",
            [(loglike(g, K, B, K, Y, K=K), ("K_alpha", "B", "K_year", "Y")),
             (loglike(g, K, Y, B, K) * 50, ("K_alpha", "Y", "B", "K_glob"))],
            ("K_alpha",))
        add(f"S-BUS_year_K{K}",
            [(loglike(g, K, Y, K, K=K), ("K_year", "Y", "K_glob")),
             (loglike(g, K, Y, K) * 3, ("K_year", "Y", "K_glob"))],
            ("K_year",))
        add(f"S-BUS_top_K{K}",
            [(loglike(g, K, K=K), ("K_glob",)), (loglike(g, K) * 6, ("K_glob",))],
            ("K_glob",))
    # chain a-b-c (forces a multi-step elimination) and a 4-factor star
    g = t.Generator().manual_seed(77)
    add("chain3",
        [(loglike(g, 4, 5), ("K1", "K2")), (loglike(g, 5, 6), ("K2", "K3")), (loglike(g, 6, 3), ("K3", "K4"))],
        ("K1", "K2", "K3", "K4"))
    add("star4_keep_plate",
        [(loglike(g, 7, 4), ("p", "Kc")), (loglike(g, 4, 3), ("Kc", "K1")),
         (loglike(g, 7, 4, 5), ("p", "Kc", "K2")), (loglike(g, 5), ("K2",))],
        ("Kc", "K1", "K2"))
    # edge cases
    add("empty_Ks_single", [(loglike(g, 10, 3), ("T", "K_a"))], ())
    add("empty_Ks_multi", [(loglike(g, 10, 3), ("T", "K_a")), (loglike(g, 3), ("K_a",))], ())
    add("single_factor", [(loglike(g, 6, 5), ("p", "K"))], ("K",))
    add("f64", [(loglike(g, 5, 4, dtype=t.float64), ("p", "K")), (loglike(g, 4, dtype=t.float64), ("K",))], ("K",))
    add("mixed_f32_f64", [(loglike(g, 5, 4), ("p", "K")), (loglike(g, 5, 4, dtype=t.float64), ("p", "K"))], ("K",))
    x = loglike(g, 5, 6)
    x[2, 3] = float("-inf")
    add("single_neg_inf", [(x, ("p", "K")), (loglike(g, 6), ("K",))], ("K",), want_grad=False)
    x = loglike(g, 5, 6)
    x[4, :] = float("-inf")
    add("all_neg_inf_row", [(x, ("p", "K")), (loglike(g, 6), ("K",))], ("K",), want_grad=False)
    save("seam_synthetic.pt", cases)


# ------------------------------------------------------------------ 3. chain
def gen_chain():
    cases = []
    for (T, K, dtype, seed) in [(1, 3, t.float32, 1), (2, 3, t.float32, 2), (3, 4, t.float32, 3),
                                (4, 3, t.float32, 4), (7, 3, t.float32, 5), (16, 10, t.float32, 6),
                                (37, 5, t.float64, 7), (1000, 30, t.float32, 8), (129, 100, t.float32, 9)]:
        g = t.Generator().manual_seed(4000 + seed)
        ms = loglike(g, T, K, K, dtype=dtype, K=K)
        kk = ref_utils.chain_logmmexp(ms)
        out = t.logsumexp(kk, -1)                         # logpq.py:135-139
        x = ms.clone().requires_grad_(True)
        gw = t.randn(K, generator=t.Generator().manual_seed(5), dtype=dtype)
        (grad,) = t.autograd.grad((t.logsumexp(ref_utils.chain_logmmexp(x), -1) * gw).sum(), x)
        big = ms.numel() > 20000
        cases.append(dict(T=T, K=K, dtype=str(dtype), seed=4000 + seed,
                          ms=None if big else ms, chain=kk, out=out, grad_out=gw,
                          grad=None if big else grad,
                          grad_checksum=(float(grad.double().sum()), float(grad.double().abs().sum()))))
    g = t.Generator().manual_seed(31)
    a, b = loglike(g, 5, 4, 4), loglike(g, 5, 4, 4)
    cases.append(dict(name="logmmexp", prev=a, curr=b, out=ref_utils.logmmexp(a, b)))
    save("chain.pt", cases)


def gen_chain_peaked():
    """Transition factors of a random-walk timeseries with a small noise scale: log N(x_t[j]; 0.9 x_{t-1}[i], 0.1).
    Most pairs are > 16 nats below the row / column maxima, so log(. + eps) floors them."""
    cases = []
    for (B, T, K, dtype, seed) in [(1, 4, 5, t.float32, 1), (1, 25, 30, t.float32, 2), (1, 11, 8, t.float64, 3),
                                   (6, 13, 10, t.float32, 4), (3, 64, 30, t.float32, 5), (1, 7, 100, t.float32, 6)]:
        g = t.Generator().manual_seed(7000 + seed)
        x = t.randn(B, T + 1, K, generator=g, dtype=dtype)
        ms = td.Normal(0.9 * x[:, :-1, :, None], 0.1).log_prob(x[:, 1:, None, :])          # [B, T, K_prev, K_curr]
        ms = ms - float(t.log(t.tensor(float(K))))
        leaves = ms.clone().requires_grad_(True)
        outs = [t.logsumexp(ref_utils.chain_logmmexp(leaves[b]), -1) for b in range(B)]     # logpq.py:135-139
        out = t.stack(outs, 0)
        chain = t.stack([ref_utils.chain_logmmexp(ms[b]) for b in range(B)], 0)
        gw = t.softmax(out.detach() + t.randn(B, K, generator=g, dtype=dtype), -1)          # ELBO-like upstream weights
        (grad,) = t.autograd.grad((out * gw).sum(), leaves)
        gc = t.randn(B, K, K, generator=g, dtype=dtype)
        l2 = ms.clone().requires_grad_(True)
        (grad_c,) = t.autograd.grad(sum((ref_utils.chain_logmmexp(l2[b]) * gc[b]).sum() for b in range(B)), l2)
        cases.append(dict(B=B, T=T, K=K, dtype=str(dtype), ms=ms, chain=chain, out=out, grad_out=gw, grad=grad,
                          grad_chain_out=gc, grad_chain=grad_c,
                          floored=float((chain < chain.amax(-1, keepdim=True) - 15.5).double().mean())))
    save("chain_peaked.pt", cases)


# ------------------------------------------------------------------ 4. recorded + e2e
class Recorder:
    def __init__(self):
        self.calls = []
        self._orig = ref_logpq.reduce_Ks

    def __enter__(self):
        def wrapped(lps, Ks):
            out = self._orig(lps, Ks)
            self.calls.append(dict(
                factors=[undim(lp) for lp in lps],
                Ks=tuple(str(k) for k in Ks),
                out=undim(out)[0], out_names=undim(out)[1]))
            return out
        ref_logpq.reduce_Ks = wrapped
        return self

    def __exit__(self, *a):
        ref_logpq.reduce_Ks = self._orig


def named_flat(d):
    return {k: (v.rename(None).detach().clone(), tuple(v.names)) for k, v in d.items()}


def capture_problem(tag, problem, K, strategies, seed, keep_calls=True, record_strategy=None, rws_grads=True):
    t.manual_seed(seed)
    sample = problem.sample(K, True)
    rec = dict(
        tag=tag, K=K, seed=seed,
        sample=tree_undim(sample.detached_sample),
        Kdims={k: str(v) for k, v in sample.groupvarname2Kdim.items()},
        data=named_flat(problem._data.to_dict()),
        platesizes=dict(problem.P.all_platesizes),
        P_inputs_params=named_flat(problem.P.inputs_params_flat_named()),
        Q_inputs_params=named_flat(problem.Q.inputs_params_flat_named()),
        elbo={},
    )
    calls = []
    for sname, strat in strategies.items():
        if sname == (record_strategy or "no_checkpoint"):
            with Recorder() as r:
                e = sample.elbo_nograd(strat)
            calls = r.calls
        else:
            e = sample.elbo_nograd(strat)
        rec["elbo"][sname] = e.detach().clone()
    # gradients of the ELBO (detached sample, i.e. elbo_rws) wrt Q's raw optimisable parameters
    opt = problem.Q._opt_params
    if rws_grads and len(opt.keys):
        for p in opt.parameters():
            p.grad = None
        sample.elbo_rws(alan.no_checkpoint).backward()
        rec["rws_grads"] = {k: getattr(opt, k).grad.detach().clone() for k in opt.keys
                            if getattr(opt, k).grad is not None}
        for p in opt.parameters():
            p.grad = None
    print(f"  {tag} K={K}: " + ", ".join(f"{k}={float(v):.6f}" for k, v in rec["elbo"].items()),
          f"[{len(calls)} reduce_Ks calls]", f"[{len(rec.get('rws_grads', {}))} param grads]")
    return rec, [dict(model=tag, K=K, call=i, **c) for i, c in enumerate(calls)] if keep_calls else []


def gen_models():
    recorded = []
    sys.path.insert(0, os.path.join(REF, "tests"))

    def strategies(split):
        return {"no_checkpoint": alan.no_checkpoint, "checkpoint": alan.checkpoint,
                "split": split}

    t.manual_seed(11)
    lg = importlib.import_module("linear_gaussian")
    rec, calls = capture_problem("linear_gaussian", lg.tp.problem, 3, strategies(alan.Split("T", 4)), 101)
    rec["known_elbo"] = lg.known_elbo.clone()
    recorded += calls
    save("e2e_linear_gaussian.pt", rec)

    t.manual_seed(12)
    lgl = importlib.import_module("linear_gaussian_latents")
    rec, calls = capture_problem("linear_gaussian_latents", lgl.tp.problem, 3,
                                 strategies(alan.Split("T", 3)), 102)
    rec["known_elbo"] = lgl.known_elbo.clone()
    recorded += calls
    # also the Split run's calls: pins chunk sizes [3,3,2,2]
    _, calls_split = capture_problem("linear_gaussian_latents_split", lgl.tp.problem, 3,
                                     strategies(alan.Split("T", 3)), 102, record_strategy="split")
    recorded += calls_split
    save("e2e_linear_gaussian_latents.pt", rec)

    t.manual_seed(13)
    m1 = importlib.import_module("model1")
    rec, calls = capture_problem("model1", m1.tp.problem, 3, strategies(alan.Split("p1", 3)), 103)
    recorded += calls
    save("e2e_model1.pt", rec)

    # ---- the remaining problems of tests/test_problem_vs_itself.py:15-30, with their constants
    others = {
        "bernoulli_no_plate": [],
        "linear_gaussian_two_params": ["prior_mean", "a_scale", "b_scale", "like_scale"],
        "linear_gaussian_two_params_corr_Q": ["prior_mean", "a_scale", "b_scale", "like_scale"],
        "linear_gaussian_two_params_corr_Q_reversed": ["prior_mean", "a_scale", "b_scale", "like_scale"],
        "linear_gaussian_two_params_dangling": ["prior_mean", "prior_scale", "like_scale", "mult"],
        "linear_gaussian_latents_dangling": ["prior_mean", "prior_scale", "z_scale", "d_scale"],
        "linear_gaussian_latents_batch": ["prior_mean", "prior_scale", "z_scale", "d_scale"],
        "linear_multivariate_gaussian": ["prior_mean", "prior_cov", "ap_mean", "ap_cov", "like_cov"],
        "linear_multivariate_gaussian_batch": ["prior_mean", "prior_cov", "ap_mean", "ap_cov", "like_cov"],
        "linear_multivariate_gaussian_param": ["prior_mean", "prior_cov", "ap_mean", "ap_cov", "like_cov"],
    }
    bundle = {}
    for i, (name, consts) in enumerate(others.items()):
        t.manual_seed(40 + i)
        mod = importlib.import_module(name)
        strat = {"no_checkpoint": alan.no_checkpoint, "checkpoint": alan.checkpoint}
        cs = mod.tp.computation_strategy
        if isinstance(cs, alan.Split):
            strat["split"] = cs
        rec, _ = capture_problem(name, mod.tp.problem, 3, strat, 300 + i, keep_calls=False)
        rec["consts"] = {c: (getattr(mod, c).clone() if isinstance(getattr(mod, c), t.Tensor) else getattr(mod, c))
                         for c in consts}
        if isinstance(cs, alan.Split):
            rec["split"] = (cs.platename, cs.split_size)
        if hasattr(mod, "known_elbo"):
            rec["known_elbo"] = mod.known_elbo.clone()
        bundle[name] = rec
    save("e2e_small_models.pt", bundle)

    # ---- movielens / bus_breakdown from the shipped example data
    for mod, sub, splits in [("movielens", "movielens", ("plate_1", 38)),
                             ("bus_breakdown", "bus_breakdown", ("plate_ID", 40))]:
        mdir = os.path.join(REF, "examples", "models", sub)
        sys.path.insert(0, mdir)
        m = importlib.import_module(mod)
        prob, *_ = m._load_and_generate_problem("cpu", "opt", run=0, data_dir=os.path.join(mdir, "data") + "/")
        for K in (3, 10):
            rec, calls = capture_problem(mod, prob, K, strategies(alan.Split(*splits)), 200 + K,
                                         keep_calls=True)
            recorded += calls
            save(f"e2e_{mod}_K{K}.pt", rec)
    save("seam_recorded.pt", recorded)


def gen_wide_group():
    """A Group of seven Normal latents over one child plate (a model of ours, written against the reference's API):
    with gradients each variable contributes log P and -log Q as separate factors of ONE K -- more factors than a
    single alan_reduce launch takes, so the planner has to pre-add.  The reference evaluates it like any other."""
    n = 7
    names = [f"g{i}" for i in range(n)]
    g = t.Generator().manual_seed(77)
    data = {"d": t.randn(5, generator=g).refine_names("p")}
    mean = eval("lambda " + ", ".join(names) + ": " + " + ".join(f"{0.5 + 0.25 * i} * {v}" for i, v in enumerate(names)))
    P = alan.Plate(**{v: alan.Normal(0.1 * i, 1.0 + 0.1 * i) for i, v in enumerate(names)},
                   p=alan.Plate(d=alan.Normal(mean, 1.5)))
    Q = alan.Plate(grp=alan.Group(**{v: alan.Normal(alan.OptParam(0.2 * i - 0.5), alan.OptParam(0.05 * i - 0.1, transformation=t.exp))
                                     for i, v in enumerate(names)}),
                   p=alan.Plate(d=alan.Data()))
    sizes = {"p": 5}
    prob = alan.Problem(alan.BoundPlate(P, sizes), alan.BoundPlate(Q, sizes), data)
    strat = {"no_checkpoint": alan.no_checkpoint, "checkpoint": alan.checkpoint, "split": alan.Split("p", 2)}
    rec, _ = capture_problem("wide_group", prob, 4, strat, 707, keep_calls=False)
    save("e2e_wide_group.pt", rec)


def gen_posterior():
    """What the reference's posterior K sampling draws FROM (the tables handed to t.multinomial), so that the draws
    themselves -- random -- need not be compared:
      sample_Ks (reduce_Ks.py:35-83): per elimination step, walked backwards, the table sum(lps) over the step's Ks
        (before already-drawn indices are plugged in: deterministic) and the Ks drawn from it;
      sample_Ks_timeseries (reduce_Ks.py:85-232): called directly on a hand-built [T, K_init, K] factor and given
        initial-state indices (an end-to-end timeseries model cannot be built with this torch: Timeseries.py:123), with
        t.multinomial wrapped to record its weights at every timestep."""
    from alan.reduce_Ks import collect_lps, sample_Ks_timeseries as ref_sample_ts
    out = {"sample_Ks": [], "timeseries": []}
    g = t.Generator().manual_seed(515)
    cases = [
        ("chain3", [(loglike(g, 4, 5), ("K1", "K2")), (loglike(g, 5, 6), ("K2", "K3")), (loglike(g, 6, 3), ("K3", "K4"))],
         ("K1", "K2", "K3", "K4")),
        ("star_plate", [(loglike(g, 7, 4), ("p", "Kc")), (loglike(g, 4, 3), ("Kc", "K1")),
                        (loglike(g, 7, 4, 5), ("p", "Kc", "K2")), (loglike(g, 5), ("K2",))], ("Kc", "K1", "K2")),
        ("top_level", [(loglike(g, 5), ("Ka",)), (loglike(g, 4), ("Kb",)), (loglike(g, 5, 4) * 3, ("Ka", "Kb"))], ("Ka", "Kb")),
        ("single", [(loglike(g, 6, 5), ("p", "K"))], ("K",)),
    ]
    for name, factors, Ks in cases:
        dm = {}
        lps = [todim(x, n, dm) for x, n in factors]
        _, lps_for_sampling, Ks_to_sample = collect_lps(lps, [dm[k] for k in Ks])
        steps = []
        for lps_s, kd in zip(lps_for_sampling[::-1], Ks_to_sample[::-1]):
            tab, names = undim(sum(lps_s))
            steps.append(dict(Ks=tuple(str(k) for k in kd), table=tab, names=names))
        out["sample_Ks"].append(dict(name=name, factors=[(x, tuple(n)) for x, n in factors], Ks=Ks, steps=steps))
    for (T, K, N, seed) in [(4, 3, 6, 1), (7, 5, 9, 2), (12, 10, 20, 3)]:
        gg = t.Generator().manual_seed(900 + seed)
        P = 2          # an enclosing plate: two independent chains (without one the reference's index bookkeeping,
        #                t.zeros(...)[N_dim, []] at reduce_Ks.py:123, breaks under this torch)
        ms = loglike(gg, P, T, K, K, K=K) + 0.7 * t.randn(P, T, 1, K, generator=gg)
        init = t.randint(0, K, (N,), generator=gg)
        dm = {}
        lp = todim(ms, ("p", "T", "K_init", "K_ts"), dm)
        N_dim = Dim("N", N)
        recorded = []
        real = t.multinomial

        def spy(probs, *a, **k):
            recorded.append(undim(probs))
            return real(probs, *a, **k)

        t.multinomial = spy
        try:
            res = ref_sample_ts([lp], [dm["K_ts"]], [dm["K_init"]], N_dim, N, dm["T"], {dm["K_init"]: init[N_dim]})
        finally:
            t.multinomial = real
        assert len(recorded) == T and all(n == ("p",) for _, n in recorded), [n for _, n in recorded]
        # recorded[0] belongs to t = T-1 (the reference walks backwards); store in time order [p, T, K], normalised
        probs = t.stack([x for x, _ in recorded[::-1]], 1)
        out["timeseries"].append(dict(T=T, K=K, N=N, ms=ms, init=init, probs=probs / probs.sum(-1, keepdim=True),
                                      drawn=undim(res[dm["K_ts"]])))
    save("posterior.pt", out)


def gen_baseline_sizes():
    """The BASELINE.json configurations at their literal sizes (SURVEY section 6 timed the reference on exactly these):
    C2 movielens K=30, C4 movielens K=100 under Split('plate_1', 38) (the reference's unsplit K=100 evaluation would
    materialise a 21.6 GB [K,K,K,300,18] broadcast: only the Split run is recorded), C3 bus_breakdown K=30.
    reduce_Ks calls are not recorded here (the K=30 factor alone is 32 MB)."""
    for mod, sub, splits in [("movielens", "movielens", ("plate_1", 38)),
                             ("bus_breakdown", "bus_breakdown", ("plate_ID", 40))]:
        mdir = os.path.join(REF, "examples", "models", sub)
        sys.path.insert(0, mdir)
        m = importlib.import_module(mod)
        prob, *_ = m._load_and_generate_problem("cpu", "opt", run=0, data_dir=os.path.join(mdir, "data") + "/")
        strat = {"no_checkpoint": alan.no_checkpoint, "checkpoint": alan.checkpoint, "split": alan.Split(*splits)}
        rec, _ = capture_problem(mod, prob, 30, strat, 230, keep_calls=False)
        save(f"e2e_{mod}_K30.pt", rec)
        if mod == "movielens":
            rec, _ = capture_problem(mod, prob, 100, {"split": alan.Split(*splits)}, 2100, keep_calls=False,
                                     record_strategy="none", rws_grads=False)
            save(f"e2e_{mod}_K100_split38.pt", rec)


if __name__ == "__main__":
    if sys.argv[1:] == ["posterior"]:
        gen_posterior()
        sys.exit(0)
    if sys.argv[1:] == ["wide_group"]:
        gen_wide_group()
        sys.exit(0)
    if sys.argv[1:] == ["baseline_sizes"]:        # one group of fixtures only (the others are not regenerated)
        gen_baseline_sizes()
        sys.exit(0)
    if sys.argv[1:] == ["chain_peaked"]:          # one fixture only (the others are not regenerated)
        gen_chain_peaked()
        sys.exit(0)
    gen_lse_dims()
    gen_seam_synthetic()
    gen_chain()
    gen_chain_peaked()
    gen_models()
    gen_wide_group()
    gen_posterior()
    gen_baseline_sizes()
