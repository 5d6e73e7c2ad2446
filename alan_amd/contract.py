"""
torchdim-level entry points of the hot path, with the reference's names, argument meaning and
error behaviour -- each one a thin bridge onto the HIP engine (engine.py -> libalan_mi355.so):

    reduce_Ks(lps, Ks_to_sum)                    reduce_Ks.py:236-244
    collect_lps(lps, Ks_to_sum)                  reduce_Ks.py:255-298
    logsumexp_sum(Ks, *lps)                      reduce_Ks.py:249-251
    logsumexp_dims(x, dims, ignore_extra_dims)   utils.py:207-222
    logmeanexp_dims(x, dims)                     utils.py:224-225
    chain_logmmexp(ms)                           utils.py:478-510

plus ``reduce_Ks_plate`` -- reduce_Ks with the plate sum of logpq.py:149 fused into the last launch,
which is what the build's own _logPQ_plate calls.  Nothing here computes on the CPU.
"""
import math

import torch as t

from . import engine as E
from . import native as N
from .dims import Dim, check_dims, dims_of, unwrap, wrap, union_dims


def _factors(lps):
    out = []
    for lp in lps:
        # "There shouldn't be any non-torchdim dimensions" (reduce_Ks.py:13-14)
        assert lp.shape == (), "log-prob factors must have no positional dims"
        out.append(unwrap(lp))
    return out


def logsumexp_dims(x, dims, ignore_extra_dims=False):
    check_dims(dims)
    have = set(dims_of(x))
    if ignore_extra_dims:
        dims = tuple(d for d in dims if d in have)
    if not all(d in have for d in dims):
        raise Exception("dims provided that aren't in x; can ignore them by providing "
                        "ignore_extra_dims=True kwarg")
    if len(dims) == 0:
        return x
    return _positional_lse(x, dims, 0.0)


def _positional_lse(x, dims, add_const):
    """x may carry positional dims too (predictive_ll-style callers): they ride along as extra keys."""
    pos, ds = unwrap(x)
    extra = tuple(("_pos", i) for i in range(pos.ndim - len(ds)))
    out, odims = E.reduce_factors([(pos, (*ds, *extra))], reduce=tuple(dims), add_const=add_const)
    # put first-class dims first, positional ones back in their original order
    fc = [d for d in odims if isinstance(d, Dim)]
    where = {d: i for i, d in enumerate(odims)}
    perm = [where[d] for d in (*fc, *extra)]
    out = out.permute(perm) if len(perm) > 1 else out
    return wrap(out, fc)


def logmeanexp_dims(x, dims):
    check_dims(dims)
    if not all(d in set(dims_of(x)) for d in dims):
        raise Exception("dims provided that aren't in x; can ignore them by providing "
                        "ignore_extra_dims=True kwarg")
    if len(dims) == 0:
        return x
    return _positional_lse(x, tuple(dims), -sum(math.log(d.size) for d in dims))


def logsumexp_sum(_Ks_to_sum, *lps_to_reduce):
    """log-sum-exp over ``_Ks_to_sum`` of the broadcast sum of the factors -- ONE fused launch, the
    sum is never materialised (the reference materialises it: reduce_Ks.py:251)."""
    facs = _factors(lps_to_reduce)
    have = set(union_dims(lps_to_reduce))
    Ks = tuple(k for k in _Ks_to_sum if k in have)          # ignore_extra_dims=True
    out, dims = E.reduce_factors(facs, reduce=Ks)
    return wrap(out, dims)


def collect_lps(lps, Ks_to_sum):
    """Returns (result, per-step factor lists, per-step Ks) like reduce_Ks.py:255-298; steps that sum
    no K are dropped from the two lists, as the reference does (:289-296)."""
    check_dims(list(Ks_to_sum), "dims")
    result, record = _contract(lps, Ks_to_sum, ())
    all_reduced, Ks_steps = [], []
    for group, now in record:
        if len(now):
            all_reduced.append([wrap(x, d) for x, d in group])
            Ks_steps.append(tuple(now))
    return result, all_reduced, Ks_steps


def _contract(lps, Ks_to_sum, plate):
    facs = _factors(lps)
    have = set(union_dims(lps))
    for k in Ks_to_sum:
        if k not in have:
            raise Exception(f"K dimension {k} to sum is on none of the factors")
    out, dims, record = E.contract(facs, tuple(Ks_to_sum), plate=tuple(plate))
    return wrap(out, dims), record


def reduce_Ks(lps, Ks_to_sum):
    """Sum over Ks_to_sum in log space, returning a single torchdim tensor."""
    check_dims(list(Ks_to_sum), "dims")
    result, _ = _contract(lps, Ks_to_sum, ())
    return result


def reduce_Ks_plate(lps, Ks_to_sum, platedim):
    """reduce_Ks followed by ``.sum(platedim)`` (logpq.py:128,149) as one fused contraction."""
    check_dims(list(Ks_to_sum), "dims")
    result, _ = _contract(lps, Ks_to_sum, (platedim,))
    return result


class _Chain(t.autograd.Function):
    """Timeseries plate: [T,K,K] -> [K] = logsumexp(chain_logmmexp(ms), -1)  (or the [K,K] chain); a leading
    batch dim (chains nested under other plate / K dims) rides along.  The forward keeps every round of the
    pairwise tree; the backward walks it (autograd through utils.py:503-507, eps floor included)."""

    @staticmethod
    def forward(ctx, ms, want_chain):
        vec, chain, tree = N.chain_logmmexp(ms.detach(), want_chain=want_chain)
        ctx.want_chain = want_chain
        ctx.tree = tree
        ctx.save_for_backward(ms, vec)
        return (chain if want_chain else vec)

    @staticmethod
    def backward(ctx, g):
        ms, vec = ctx.saved_tensors
        if ctx.want_chain:
            return N.chain_logmmexp_backward(ms.detach(), ctx.tree, grad_chain=g.detach()), None
        return N.chain_logmmexp_backward(ms.detach(), ctx.tree, out_vec=vec, grad_vec=g.detach()), None


def chain_logmmexp(ms):
    """[T,K,K] -> [K,K], the log of the ordered product of the exp'd matrices (utils.py:509-510).  [B,T,K,K] ->
    [B,K,K]: what the reference gets from torchdim batch dims on ``ms``."""
    assert ms.ndim in (3, 4)
    assert ms.shape[-2] == ms.shape[-1]
    return _Chain.apply(ms, True)


CHAIN_KERNEL_MAX_K = {t.float32: 100, t.float64: 80}     # the backward holds three [K,K] operands in LDS (160 KB)


def chain_logmmexp_lse(ms):
    """chain_logmmexp followed by t.logsumexp(., -1) (logpq.py:135-139).

    K <= 100: the LDS segment-product kernel (3 launches at T = 1000).  Larger K (the reference's
    ground-truth tests use K = 1000 at T = 4): the same quantity as a right-to-left log-matvec scan,
    u_t[i] = LSE_j(M_t[i,j] + u_{t+1}[j]) -- O(T K^2) instead of O(T K^3), one alan_reduce per step.
    ``ms`` may carry a leading batch dim [B,T,K,K] -> [B,K]."""
    assert ms.ndim in (3, 4)
    assert ms.shape[-2] == ms.shape[-1]
    if ms.shape[-1] <= CHAIN_KERNEL_MAX_K.get(ms.dtype, 0):
        return _Chain.apply(ms, False)
    if ms.ndim == 3:
        keys, at = ("i", "j"), (lambda s: ms[s])
    else:
        keys, at = ("b", "i", "j"), (lambda s: ms[:, s])
    u, _ = E.reduce_factors([(at(-1), keys)], reduce=("j",))
    for step in range(ms.shape[-3] - 2, -1, -1):
        u, _ = E.reduce_factors([(at(step), keys), (u, (*keys[:-2], "j"))], reduce=("j",))
    return u
