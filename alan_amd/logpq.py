"""
The plate recursion that turns a sample tree into the ELBO (logpq.py of the reference):

    logPQ_plate  -- owns the Split loop / multi-GPU shard of one plate      (logpq.py:15-60)
    _logPQ_plate -- gathers the plate's log-prob factors, then eliminates the plate's K dims and
                    sums the plate out on the HIP engine                     (logpq.py:68-155)
    logPQ_group  -- per-group factors  log P - log Q - log K                 (logpq.py:157-254)

Differences from the reference that matter for the GPU (results agree to rounding):
  * a group contributes SEVERAL factors (log P terms, and -(log Q + log K)) instead of one
    pre-added tensor, so the big [plate, K, K, K] log P tensor is read exactly once, by the kernel;
  * reduce_Ks and the plate sum are one fused contraction (contract.reduce_Ks_plate);
  * factors are produced with the plate dims outermost and the group's own K innermost.
"""
import math

import torch as t
import torch.utils.checkpoint

from .contract import chain_logmmexp_lse, reduce_Ks, reduce_Ks_plate
from .dims import Dim, dims_of, is_tensor
from .model import Plate, tree_tensors, update_scope
from .split import Split, all_reduce_sum, checkpoint, no_checkpoint
from .timeseries import Timeseries


def logPQ_plate(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy):
    chunks = computation_strategy.split_args(
        name=name, sample=sample, inputs_params=inputs_params, extra_log_factors=extra_log_factors,
        data=data, all_platedims=all_platedims)
    run = _logPQ_plate if computation_strategy is no_checkpoint else _logPQ_plate_checkpointed

    sharded = len(chunks) > 1 and getattr(computation_strategy, "sharded", lambda: False)()
    mine = computation_strategy.my_chunks(len(chunks)) if sharded else range(len(chunks))

    lpq = None
    for i in mine:
        lpq = run(name=name, P=P, Q=Q, scope=scope, active_platedims=active_platedims,
                  groupvarname2Kdim=groupvarname2Kdim, varname2groupvarname=varname2groupvarname,
                  sampler=sampler, computation_strategy=computation_strategy, prev_lpq=lpq, **chunks[i])
    assert is_tensor(lpq)
    if sharded:
        lpq = all_reduce_sum(lpq, computation_strategy.group)
    return lpq, (), (), ()


def _logPQ_plate_checkpointed(**kwargs):
    return torch.utils.checkpoint.checkpoint(_call_kwargs, kwargs, use_reentrant=False)


def _call_kwargs(kwargs):
    return _logPQ_plate(**kwargs)


def _logPQ_plate(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                 all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                 prev_lpq):
    assert isinstance(P, Plate) and isinstance(Q, Plate)
    platedim = None
    if name is not None:
        platedim = all_platedims[name]
        active_platedims = [*active_platedims, platedim]
    scope = update_scope(scope, inputs_params)
    scope = update_scope(scope, sample)
    assert set(P.flat_prog) == set(Q.flat_prog)

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
                sampler=sampler, computation_strategy=computation_strategy)
            lps.append(lp)
        elif kind == "data":
            assert sample.get(child) is None
            lp, _ = P.flat_prog[child].log_prob(data[child], scope, dim_order=(active_platedims, ()))
            lps.append(lp)
        else:
            facs, k_plain, k_ts, k_init = logPQ_group(
                child, {v: P.flat_prog[v] for v in q}, q, sample, scope, active_platedims,
                groupvarname2Kdim, varname2groupvarname, sampler)
            lps.extend(facs)
            Ks.extend(k_plain)
            K_currs.extend(k_ts)
            K_inits.extend(k_init)
    assert len(K_currs) == len(K_inits)

    if name is None:
        return reduce_Ks(lps, Ks)

    if K_inits:
        # timeseries plate (logpq.py:131-143): eliminate the ordinary Ks, then the chain over T
        assert len(K_inits) == 1 and len(K_currs) == 1
        assert prev_lpq is None, "a timeseries plate cannot be split"
        lp = reduce_Ks(lps, Ks)
        others = [d for d in dims_of(lp) if d not in {platedim, K_inits[0], K_currs[0]}]
        if others:
            raise NotImplementedError("alan_amd: timeseries plates nested under other K/plate dims are not "
                                      "supported yet")
        ms = lp.order(platedim, K_inits[0], K_currs[0])
        return chain_logmmexp_lse(ms)[K_inits[0]]

    lp = reduce_Ks_plate(lps, Ks, platedim)
    if prev_lpq is not None:
        assert set(dims_of(lp)) == set(dims_of(prev_lpq))
        lp = prev_lpq + lp
    return lp


def logPQ_group(name, prog_P, prog_Q, sample, scope, active_platedims, groupvarname2Kdim,
                varname2groupvarname, sampler):
    """Factors of one latent group on its K dim:  sum_v log P(v) , -(reduce_logQ(sum_v log Q(v)) + log K)."""
    assert set(prog_P) == set(prog_Q) and len(prog_P) >= 1
    Kdim = groupvarname2Kdim[name]
    T_dim = active_platedims[-1] if active_platedims else None
    order = (active_platedims, (Kdim,))

    logPs, total_logQ, Kinits = [], 0.0, []
    init_Ks = [groupvarname2Kdim[varname2groupvarname[d.init]] for d in prog_P.values() if isinstance(d, Timeseries)]
    for var in prog_P:
        x = sample[var]
        assert is_tensor(x)
        lp, Kinit_p = prog_P[var].log_prob(x, scope=scope, T_dim=T_dim, K_dim=Kdim, dim_order=order)
        lq, Kinit_q = prog_Q[var].log_prob(x, scope=scope, T_dim=T_dim, K_dim=Kdim, dim_order=order)
        if Kinit_q is not None:
            assert Kinit_p is Kinit_q
        if Kinit_p is not None:
            Kinits.append(Kinit_p)
        logPs.append(lp)
        total_logQ = total_logQ + lq
    lq = sampler.reduce_logQ(total_logQ, active_platedims, Kdim)
    neg_q = -(lq + math.log(Kdim.size))                     # small: [plates, K]
    # log P terms that are no bigger than Q's are folded into it; the big ones stay separate factors
    n_q = _numel(neg_q)
    small = [lp for lp in logPs if _numel(lp) <= n_q]
    big = [lp for lp in logPs if _numel(lp) > n_q]
    for lp in small:
        neg_q = neg_q + lp
    factors = [*big, neg_q]

    if Kinits:
        for k in init_Ks:
            assert k is Kinits[0]
        return factors, (), (Kdim,), (Kinits[0],)
    return factors, (Kdim,), (), ()


def _numel(x):
    n = 1
    for d in dims_of(x):
        n *= d.size
    return n
