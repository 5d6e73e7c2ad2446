"""
The plate recursion that turns a sample tree into the ELBO (logpq.py of the reference):

    logPQ_plate  -- owns the Split loop / multi-GPU shard of one plate      (logpq.py:15-60)
    _logPQ_plate -- gathers the plate's log-prob factors, then eliminates the plate's K dims and
                    sums the plate out on the HIP engine                     (logpq.py:68-155)
    logPQ_group  -- per-group factors  log P - log Q - log K                 (logpq.py:157-254)

All trees handed to these functions hold ``PT`` leaves (dims.py): plain positional tensors plus their
first-class dims.  Differences from the reference that matter for the GPU (results agree to rounding):
  * a group contributes SEVERAL factors (log P terms, and -(log Q + log K)) instead of one
    pre-added tensor, so the big [plate, K, K, K] log P tensor is read exactly once, by the kernel;
  * reduce_Ks and the plate sum are one fused contraction (engine.contract with plate=...);
  * factors are produced with the plate dims outermost and the group's own K innermost.
"""
import contextlib
import math

import torch as t
import torch.utils.checkpoint

from . import engine as E
from .dims import PT, ExpPT, LazyNormalPT, PartialSumPT, ShiftPT, pt_add, pt_align
from .model import Plate, tree_tensors, update_scope
from .split import all_reduce_sum, no_checkpoint
from .dist import TorchDimDist
from . import native as N
from .timeseries import Timeseries


def logPQ_plate(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                dimcache=None, merge=None):
    split_here = name is not None and name == getattr(computation_strategy, "platename", None)
    if split_here and merge is None and computation_strategy.merging():
        # the merged slice under Split's memory bound: any engine allocation beyond split.MERGE_MAX_BYTES sends the plate
        # back to the reference's per-chunk loop
        from . import split as S
        saved, E._ALLOC_LIMIT[0] = E._ALLOC_LIMIT[0], S.MERGE_MAX_BYTES
        try:
            return logPQ_plate(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                               all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                               dimcache, merge=True)
        except E.TooLargeForMergedSplit:
            merge = False
        finally:
            E._ALLOC_LIMIT[0] = saved
    kw = {"merge": merge} if split_here else {}
    chunks = computation_strategy.split_args(
        name=name, sample=sample, inputs_params=inputs_params, extra_log_factors=extra_log_factors,
        data=data, all_platedims=all_platedims, **kw)
    run = _logPQ_plate if computation_strategy is no_checkpoint else _logPQ_plate_checkpointed

    sharded = len(chunks) > 1 and getattr(computation_strategy, "sharded", lambda: False)()
    mine = computation_strategy.my_chunks(len(chunks)) if sharded else range(len(chunks))

    parts = []
    for i in mine:
        parts.append(run(name=name, P=P, Q=Q, scope=scope, active_platedims=active_platedims,
                         groupvarname2Kdim=groupvarname2Kdim, varname2groupvarname=varname2groupvarname,
                         sampler=sampler, computation_strategy=computation_strategy, prev_lpq=None,
                         dimcache=dimcache, **chunks[i]))
    # the chunks' results are summed (logpq.py:151-153) -- in one stacked reduction rather than a chain of adds
    lpq = parts[0]
    if len(parts) > 1 or sharded:
        N.flush()                  # a data-only plate's chunk results may still be queued producer launches
    if len(parts) > 1:
        for p in parts[1:]:
            assert set(p.ids) == set(lpq.ids)
        lpq = PT(t.stack([pt_align(p, lpq.ids) for p in parts]).sum(0), lpq.dims)
    assert isinstance(lpq, PT)
    if sharded:
        lpq = all_reduce_sum(lpq, computation_strategy.group)
    return lpq, (), (), ()


def _logPQ_plate_checkpointed(**kwargs):
    if not t.is_grad_enabled():
        return _logPQ_plate(**kwargs)
    dims = []

    def fresh(tree):
        # lazily-transformed parameters cache their value on first use: give the forward run and the
        # recomputation identical (un-materialised) starting points
        return {k: (fresh(v) if isinstance(v, dict) else ExpPT(v.raw, v.dims) if isinstance(v, ExpPT) else v)
                for k, v in tree.items()}

    def body(kw):
        kw = {**kw, "inputs_params": fresh(kw["inputs_params"]), "scope": fresh(kw["scope"])}
        out = _logPQ_plate(**kw)
        dims.append(out.dims)
        return out.x

    x = torch.utils.checkpoint.checkpoint(body, kwargs, use_reentrant=False)
    return PT(x, dims[0])


def _fused_plate_step(lps, Ks, plate):
    """The whole plate step as ONE launch (engine.normal_lse) when it has the hierarchical-Normal shape:
    exactly one unevaluated Normal factor N(value[plate, K]; loc[l], scale[s]), every other factor on dims within
    {plate, K}, log-sum-exp over K, sum over the plate.  None otherwise (the caller contracts as usual)."""
    lazy = [lp for lp in lps if isinstance(lp, LazyNormalPT) and not lp.materialised]
    if len(lazy) != 1 or len(plate) != 1 or len(Ks) != 1:
        return None
    z, pl, K = lazy[0], plate[0], Ks[0]
    if set(z.value.ids) != {id(pl), id(K)} or len(z.loc.dims) != 1 or len(z.scale.dims) != 1 or z.loc_mul != 1.0:
        return None
    smalls = []
    for lp in lps:
        if lp is z:
            continue
        if lp.n_pos or not set(lp.ids) <= {id(pl), id(K)}:
            return None
        smalls.append((lp.x, lp.dims))
    res = E.normal_lse((z.value.x, z.value.dims), (z.loc.x, z.loc.dims), (z.scale.x, z.scale.dims), smalls, pl, K,
                       log_scale=z.log_scale, partials=PARTIAL_PLATE_SUMS)
    if res is None:
        return None
    out, dims = res
    return PartialSumPT(out, dims) if out.ndim == len(dims) + 1 else PT(out, dims)


def _chain_of_terms(lps, Ks, core):
    """A timeseries plate with no K of its own to sum: its factors are only ADDED into the chain's [T, K_init, K]
    input (reduce_Ks with Ks = [], logpq.py:128) -- the chain's first round can add them on load (no launch, no
    3.6 MB tensor written and read back at T=1000, K=30).  Gradient-free, <= 3 factors, at most one batch dim."""
    if Ks or t.is_grad_enabled() or not 1 <= len(lps) <= 3 or N._TIMER[0] is not None:
        return None
    # at most one factor may be an unevaluated Normal with scalar events (the transition, timeseries.py): the chain's
    # first round computes it on load (alan_chain_logmmexp_terms_normal) and the [T, K_init, K] tensor is never written
    lazy = [lp for lp in lps if isinstance(lp, LazyNormalPT) and not lp.materialised]
    if len(lazy) > 1 or (lazy and (len(lps) < 2 or lazy[0].grad or
                                   any(p.n_pos for p in (lazy[0].value, lazy[0].loc, lazy[0].scale)))):
        lazy = []                                         # (reading .x below evaluates them the usual way)
    trans = lazy[0] if lazy else None
    lps = [lp for lp in lps if lp is not trans]
    if any(lp.n_pos for lp in lps):
        return None
    xs = [lp.x for lp in lps]
    # (the transition's location may be the un-concatenated previous state, dims.ShiftPT: looked at through its series)
    shift = trans.loc if trans is not None and isinstance(trans.loc, ShiftPT) and not trans.loc.materialised else None
    peek = lambda p: p.rest if p is shift else p.x
    if trans is not None and not all(peek(p).is_cuda and peek(p).dtype == xs[0].dtype for p in (trans.value, trans.loc, trans.scale)):
        lps, xs, trans, shift = [*lps, trans], [*xs, trans.x], None, None
    if not all(x.is_cuda and x.dtype == xs[0].dtype for x in xs) or xs[0].dtype not in (t.float32, t.float64):
        return None
    K = core[1].size
    from .contract import CHAIN_KERNEL_MAX_K
    if core[2].size != K or K > CHAIN_KERNEL_MAX_K.get(xs[0].dtype, 0):
        return None
    core_ids = [id(d) for d in core]
    batch = []
    every = [*lps, *([trans] if trans is not None else [])]
    for lp in every:
        for d in lp.dims:
            if id(d) not in core_ids and all(d is not b for b in batch):
                batch.append(d)
    if len(batch) > 1 or not all(any(lp.has(c) for lp in every) for c in core):
        if trans is not None:
            trans.x                                       # (evaluated: the caller contracts the usual way)
        return None
    ids = [id(d) for d in (*batch, *core)]
    shape = [d.size for d in (*batch, *core)]

    def view(lp):
        x = pt_align(lp, ids).expand(shape)                     # stride 0 where the factor lacks a dim
        return x if batch else x.unsqueeze(0)

    normal = None
    if trans is not None and shift is not None:
        # location = init at step 0, the series one step behind after it: two views, no concatenation
        rest = view(PT(shift.rest, shift.dims))
        lead = tuple(d for d in shift.dims if d is not core[0])
        first = pt_align(PT(shift.first, lead), ids).expand([1 if i == id(core[0]) else n for i, n in zip(ids, shape)])
        normal = (view(trans.value), rest, view(trans.scale), trans.loc_mul, trans.log_scale,
                  first if batch else first.unsqueeze(0))
    elif trans is not None:
        normal = (view(trans.value), view(trans.loc), view(trans.scale), trans.loc_mul, trans.log_scale)
    vec = N.chain_logmmexp_terms([view(lp) for lp in lps], normal)
    return PT(vec if batch else vec[0], (*batch, core[1]))


PARTIAL_PLATE_SUMS = True
"""The fused plate step of a gradient-free evaluation leaves its per-slice partial sums for the parent's contraction to
add as it loads them (engine.contract, role PRESUM) instead of adding them in a launch of its own."""


def _contract(lps, Ks, plate=(), final=False):
    fused = _fused_plate_step(lps, Ks, plate)
    if fused is not None:
        return fused
    for lp in lps:
        # "There shouldn't be any non-torchdim dimensions" (reduce_Ks.py:13-14)
        assert lp.n_pos == 0, "log-prob factors must have no positional dims"
    facs, part = [], None
    for lp in lps:
        if isinstance(lp, PartialSumPT) and not lp.materialised and part is None:
            part = lp                                  # (one per contraction; a second one is summed the usual way)
            facs.append((lp.parts, (E.presum_dim(lp.parts.shape[0]), *lp.dims)))
        else:
            facs.append((lp.x, lp.dims))
    out, dims, _ = E.contract(facs, tuple(Ks), plate=tuple(plate), final=final)
    return PT(out, dims)


def plate_factors(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                  all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                  dimcache=None):
    """All log-prob factors of one plate (lp_getter, logpq.py:257-332): extra factors, every latent group's
    factors, data likelihoods and the fully-reduced results of the child plates.
    Returns (platedim, active_platedims, scope, lps, Ks, K_currs, K_inits)."""
    assert isinstance(P, Plate) and isinstance(Q, Plate)
    platedim = None
    if name is not None:
        platedim = all_platedims[name]
        active_platedims = [*active_platedims, platedim]
    scope = update_scope(scope, inputs_params)
    scope = update_scope(scope, sample)

    lps = list(tree_tensors(extra_log_factors).values())
    Ks, K_currs, K_inits = [], [], []
    for kind, child, q in Q.entries():
        if kind == "plate":
            lp, *_ = logPQ_plate(
                name=child, P=P.flat_prog[child], Q=q, sample=sample[child],
                inputs_params=inputs_params.get(child, {}), data=data.get(child, {}),
                extra_log_factors=extra_log_factors.get(child, {}), scope=scope,
                active_platedims=active_platedims, all_platedims=all_platedims,
                groupvarname2Kdim=groupvarname2Kdim, varname2groupvarname=varname2groupvarname,
                sampler=sampler, computation_strategy=computation_strategy, dimcache=dimcache)
            lps.append(lp)
        elif kind == "data":
            assert sample.get(child) is None
            with N.may_defer():          # a factor of this plate's contraction and nothing else
                lp, _ = P.flat_prog[child].log_prob(data[child], scope, dim_order=(active_platedims, ()),
                                                    dimcache=dimcache)
            lps.append(lp)
        else:
            facs, k_plain, k_ts, k_init = logPQ_group(
                child, {v: P.flat_prog[v] for v in q}, q, sample, scope, active_platedims,
                groupvarname2Kdim, varname2groupvarname, sampler, dimcache)
            lps.extend(facs)
            Ks.extend(k_plain)
            K_currs.extend(k_ts)
            K_inits.extend(k_init)
    assert len(K_currs) == len(K_inits)
    return platedim, active_platedims, scope, lps, Ks, K_currs, K_inits


def _logPQ_plate(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                 all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                 prev_lpq, dimcache=None):
    if name is not None and not tree_tensors(extra_log_factors) and \
            all(kind == "data" for kind, _, _ in Q.entries()) and len(Q.grouped_prog):
        # a plate of observations only (movielens plate_2, bus_breakdown's ID plate): no K of its own, so
        # sum_plate commutes with everything and each likelihood's producer sums the plate out itself
        platedim = all_platedims[name]
        inner = update_scope(update_scope(scope, inputs_params), sample)
        lp = None
        only = len(Q.grouped_prog) == 1 and prev_lpq is None      # then the result is a factor of the parent's launch
        for _, child, _ in Q.entries():
            with (N.may_defer() if only else contextlib.nullcontext()):
                f, _ = P.flat_prog[child].log_prob(data[child], inner, dim_order=(active_platedims, ()),
                                                   dimcache=dimcache, sum_dims=(platedim,))
            lp = f if lp is None else pt_add(lp, f)
        if prev_lpq is not None:
            assert set(lp.ids) == set(prev_lpq.ids)
            lp = pt_add(prev_lpq, lp)
        return lp

    platedim, active_platedims, scope, lps, Ks, K_currs, K_inits = plate_factors(
        name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims, all_platedims,
        groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy, dimcache)

    if name is None:
        return _contract(lps, Ks, final=True)

    if K_inits:
        # timeseries plate (logpq.py:131-143): eliminate the ordinary Ks, then the chain over T
        assert len(K_inits) == 1 and len(K_currs) == 1
        assert prev_lpq is None, "a timeseries plate cannot be split"
        core = (platedim, K_inits[0], K_currs[0])
        fast = _chain_of_terms(lps, Ks, core)
        if fast is not None:
            return fast
        lp = _contract(lps, Ks)
        # lp.order(T, K_init, K_curr) (logpq.py:133): every other dim -- enclosing plates, Ks of parents in higher
        # plates -- stays a batch dim of the chain
        assert all(lp.has(d) for d in core)
        batch = tuple(d for d in lp.dims if not any(d is c for c in core))
        ms = pt_align(lp, tuple(id(d) for d in (*batch, *core)))
        from .contract import chain_logmmexp_lse
        if not batch:
            return PT(chain_logmmexp_lse(ms), (K_inits[0],))
        sizes = [d.size for d in batch]
        flat = ms.reshape(-1, *ms.shape[-3:])
        return PT(chain_logmmexp_lse(flat).reshape(*sizes, K_inits[0].size), (*batch, K_inits[0]))

    lp = _contract(lps, Ks, (platedim,))
    if prev_lpq is not None:
        assert set(lp.ids) == set(prev_lpq.ids)
        lp = pt_add(prev_lpq, lp)
    return lp


def _for_own_log_p(x):
    """The sample as its own log P term sees it: the second output of the node that drew it, where there is one and
    gradients are recorded (dims.ReparamPT.alias: the gradient then reaches that node in a slot of its own)."""
    from .dims import ReparamPT
    if t.is_grad_enabled() and isinstance(x, ReparamPT) and x.x2 is not None:
        return x.alias()
    return x


def logPQ_group(name, prog_P, prog_Q, sample, scope, active_platedims, groupvarname2Kdim,
                varname2groupvarname, sampler, dimcache=None):
    """Factors of one latent group on its K dim:  sum_v log P(v) , -(reduce_logQ(sum_v log Q(v)) + log K)."""
    assert set(prog_P) == set(prog_Q) and len(prog_P) >= 1
    Kdim = groupvarname2Kdim[name]
    T_dim = active_platedims[-1] if active_platedims else None
    order = (active_platedims, (Kdim,))

    logPs, total_logQ, Kinits = [], None, []
    init_Ks = [groupvarname2Kdim[varname2groupvarname[d.init]] for d in prog_P.values() if isinstance(d, Timeseries)]
    # A one-variable group whose log Q carries no parent K (a factorised Q: reduce_logQ is then the
    # identity) has its -(log Q + log K) written by the log-prob producer itself -- together with log P
    # when that is a Normal on the same dims (a prior with constant / plate-level parameters).
    own = {id(Kdim), *(id(d) for d in active_platedims)}
    K = Kdim.size
    single = len(prog_P) == 1 and not any(isinstance(d, Timeseries) for d in (*prog_P.values(), *prog_Q.values()))
    if single:
        (var,) = prog_P
        tP, tQ = prog_P[var].tdd(scope, dimcache), prog_Q[var].tdd(scope, dimcache)
        # (these results only ever feed the plate's contraction -- or reduce_logQ, itself a launch: may_defer)
        with N.may_defer():
            pq = TorchDimDist.log_p_minus_q(tP, tQ, sample[var], order, own, math.log(K))
        if pq is not None:                      # log P - log Q - log K in one launch
            return [pq], (Kdim,), (), ()
        with N.may_defer():
            lp = tP.log_prob_pt(_for_own_log_p(sample[var]), dim_order=order)
            neg_q = tQ.log_prob_pt(sample[var], dim_order=order, affine=(-1.0, -math.log(K), own))
        if not set(neg_q.ids) <= own:           # log Q carries a parent K: reduce it first (Sampler.py:118-134)
            lq = sampler.reduce_logQ(neg_q, active_platedims, Kdim)
            neg_q = PT(t.sub(-math.log(K), lq.x), lq.dims)
        return [lp, neg_q], (Kdim,), (), ()
    # Several variables (a Group): a variable whose prior and posterior are both Normal on the group's own dims
    # contributes log P - log Q as ONE launch (the first such one also carries the -log K); the others
    # contribute a log P factor and their log Q to the sum that reduce_logQ sees (Sampler.py:118-134 is
    # separable over variables whose log Q has no parent K, which is exactly when the fusion applies).
    pqs, negqs = [], []          # launches that already carry "- log Q" (and, the first of them, "- log K")
    for var in prog_P:
        x = sample[var]
        assert isinstance(x, PT)
        if not isinstance(prog_P[var], Timeseries) and not isinstance(prog_Q[var], Timeseries):
            tP, tQ = prog_P[var].tdd(scope, dimcache), prog_Q[var].tdd(scope, dimcache)
            logK_here = 0.0 if (pqs or negqs) else math.log(K)
            with N.may_defer():
                pq = TorchDimDist.log_p_minus_q(tP, tQ, x, order, own, logK_here)
            if pq is not None:
                pqs.append(pq)
                continue
            with N.may_defer():                    # a factor of the plate's contraction and nothing else
                lp = tP.log_prob_pt(_for_own_log_p(x), dim_order=order)
            Kinit_p = Kinit_q = None
            if (set(x.ids) | set(tQ.all_arg_ids)) <= own:
                # log Q on the group's own dims: reduce_logQ leaves it alone and is separable over such terms, so
                # -(log Q [+ log K]) is a factor of its own, written by the producer (no adds, no negation pass)
                with N.may_defer():
                    nq = tQ.log_prob_pt(x, dim_order=order, affine=(-1.0, -logK_here, own))
                assert set(nq.ids) <= own
                negqs.append(nq)
                logPs.append(lp)
                continue
            lq = tQ.log_prob_pt(x, dim_order=order)
        else:
            with N.may_defer():                    # (the transition factor only ever feeds the plate's contraction)
                lp, Kinit_p = prog_P[var].log_prob(x, scope=scope, T_dim=T_dim, K_dim=Kdim, dim_order=order,
                                                   dimcache=dimcache)
            tQ = None if isinstance(prog_Q[var], Timeseries) else prog_Q[var].tdd(scope, dimcache)
            if tQ is not None and (set(x.ids) | set(tQ.all_arg_ids)) <= own:
                # an ordinary distribution as the timeseries' approximate posterior, on the group's own dims: its
                # -(log Q [+ log K]) is a producer-written factor, as above
                logK_here = 0.0 if (pqs or negqs) else math.log(K)
                with N.may_defer():
                    nq = tQ.log_prob_pt(x, dim_order=order, affine=(-1.0, -logK_here, own))
                assert set(nq.ids) <= own
                negqs.append(nq)
                logPs.append(lp)
                if Kinit_p is not None:
                    Kinits.append(Kinit_p)
                continue
            lq, Kinit_q = prog_Q[var].log_prob(x, scope=scope, T_dim=T_dim, K_dim=Kdim, dim_order=order,
                                               dimcache=dimcache)
        if Kinit_q is not None:
            assert Kinit_p is Kinit_q
        if Kinit_p is not None:
            Kinits.append(Kinit_p)
        logPs.append(lp)
        total_logQ = lq if total_logQ is None else pt_add(total_logQ, lq)
    neg_q = None
    if total_logQ is not None:
        assert K == total_logQ.size_of(id(Kdim))
        lq = sampler.reduce_logQ(total_logQ, active_platedims, Kdim)
        # -(log Q + log K) in one pass; every log P term stays its own factor (the contraction kernels add
        # factors on the fly, so pre-adding them would only cost extra launches)
        neg_q = PT(t.sub(0.0 if (pqs or negqs) else -math.log(K), lq.x), lq.dims)
    # every term stays its own factor: the contraction kernels add factors on the fly (up to 6 per launch; the
    # planner pre-adds the smallest ones only when a step would exceed that), so pre-adding here would only
    # cost launches
    factors = [*logPs, *pqs, *negqs] + ([neg_q] if neg_q is not None else [])

    if Kinits:
        for k in init_Ks:
            assert k is Kinits[0]
        return factors, (), (Kdim,), (Kinits[0],)
    return factors, (Kdim,), (), ()
