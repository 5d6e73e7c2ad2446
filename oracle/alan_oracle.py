"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain torch-CPU restatement of alan's tensorised marginal-likelihood hot path on
*positional* tensors that carry an explicit tuple of dim names (no functorch.dim).
It is the checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped package ``alan_amd``
never imports this module and has no CPU fallback.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against golden vectors produced by the reference itself (``tests/golden/make_golden.py``,
run in the build container against /root/reference/src):
  * logsumexp_dims / logmeanexp_dims / logmmexp / chain_logmmexp are pinned to the
    reference's own functions, imported unmodified;
  * reduce_Ks is pinned to the reference's reduce_Ks.  The reference delegates the
    *order* of pairwise eliminations to the third-party planner ``opt_einsum``
    (unpinned in the reference's setup.py:13, absent from /root/reference and from this
    image).  Order affects fp re-association only; no reference test asserts an order.
    The fixtures were generated with a greedy stand-in planner, and are additionally
    checked against an order-free fp64 brute force, so "planner order: parity unpinned"
     applies to that one degree of freedom and nothing else.

Every function cites the reference file:line it restates (paths under /root/reference).
A "factor" is a pair ``(tensor, names)`` where ``names`` is a tuple of strings, one per
positional dim of ``tensor``.
"""
import math
from typing import Sequence

import torch as t

Factor = tuple  # (t.Tensor, tuple[str, ...])


# --------------------------------------------------------------------------- helpers
def _unify(factors: Sequence[Factor]):
    """Ordered union of dim names (reference: utils.py:229-233 ``unify_dims``)."""
    seen = {}
    for _, names in factors:
        for n in names:
            seen.setdefault(n, None)
    return tuple(seen)


def _sizes(factors: Sequence[Factor]):
    sz = {}
    for x, names in factors:
        assert x.ndim == len(names), (x.shape, names)
        for n, s in zip(names, x.shape):
            assert sz.setdefault(n, s) == s, f"size mismatch on {n}"
    return sz


def align(factor: Factor, names: Sequence[str]):
    """View ``factor`` as a tensor over ``names`` (size-1 where the factor lacks a dim)."""
    x, own = factor
    assert set(own).issubset(names)
    x = x.permute([own.index(n) for n in names if n in own])
    return x[tuple(slice(None) if n in own else None for n in names)]


def broadcast_sum(factors: Sequence[Factor]):
    """``sum(lps_to_reduce)``: the materialised broadcast sum (reduce_Ks.py:251)."""
    names = _unify(factors)
    total = None
    for f in factors:
        a = align(f, names)
        total = a if total is None else total + a
    if total.shape != tuple(_sizes(factors)[n] for n in names):
        total = total.expand([_sizes(factors)[n] for n in names])
    return total, names


# --------------------------------------------------------------------------- a4
def logsumexp_dims(factor: Factor, dims: Sequence[str], ignore_extra_dims=False):
    """utils.py:207-222:  m = amax; s = exp(x-m).sum; out = log(s + finfo.eps) + m."""
    x, names = factor
    if len(set(dims)) != len(dims):
        raise Exception("Non-unique elements in dims")               # utils.py:162-165
    if ignore_extra_dims:
        dims = tuple(d for d in dims if d in names)                    # utils.py:211-212
    if not all(d in names for d in dims):
        raise Exception("dims provided that aren't in x; can ignore them by providing "
                        "ignore_extra_dims=True kwarg")              # utils.py:214-215
    if len(dims) == 0:
        return x, tuple(names)                                         # utils.py:217
    axes = [names.index(d) for d in dims]
    x_max = x.amax(axes, keepdim=True)                                 # utils.py:218
    s = (x - x_max).exp().sum(axes)                                    # utils.py:219
    out = (s + t.finfo(x.dtype).eps).log() + x_max.squeeze(axes)       # utils.py:220
    return out, tuple(n for n in names if n not in dims)


def logmeanexp_dims(factor: Factor, dims: Sequence[str]):
    """utils.py:224-225."""
    x, names = factor
    sz = dict(zip(names, x.shape))
    out, out_names = logsumexp_dims(factor, dims)
    return out - sum(math.log(sz[d]) for d in dims), out_names


def reduce_logQ(factor: Factor, active_plates: Sequence[str], Kdim: str):
    """Sampler.py:118-134: average Q's log-prob over the parent K dims."""
    _, names = factor
    parents = tuple(n for n in names if n != Kdim and n not in active_plates)
    return logmeanexp_dims(factor, parents)


# --------------------------------------------------------------------------- a3
def logsumexp_sum(Ks: Sequence[str], *factors: Factor):
    """reduce_Ks.py:249-251."""
    return logsumexp_dims(broadcast_sum(factors), tuple(Ks), ignore_extra_dims=True)


# --------------------------------------------------------------------------- planner
def greedy_path(dimsets, out_dims, sizes):
    """Stand-in for ``opt_einsum.contract_path(...)[0]`` (reduce_Ks.py:265): pairwise,
    smallest-intermediate-first.  Only the ORDER of eliminations comes from here."""
    if len(dimsets) == 1:
        return [(0,)]
    cur = [set(d) for d in dimsets]
    path = []
    while len(cur) > 1:
        best = None
        for i in range(len(cur)):
            for j in range(i + 1, len(cur)):
                rest = set()
                for k in range(len(cur)):
                    if k not in (i, j):
                        rest |= cur[k]
                res = {d for d in (cur[i] | cur[j]) if d in out_dims or d in rest}
                cost = math.prod(sizes[d] for d in res)
                if best is None or cost < best[0]:
                    best = (cost, i, j, res)
        _, i, j, res = best
        path.append((i, j))
        cur = [cur[k] for k in range(len(cur)) if k not in (i, j)] + [res]
    return path


# --------------------------------------------------------------------------- a1/a2
def collect_lps(factors: Sequence[Factor], Ks_to_sum: Sequence[str], path=None):
    """reduce_Ks.py:255-298 (the per-step intermediates are returned as well)."""
    if len(set(Ks_to_sum)) != len(Ks_to_sum):
        raise Exception("Non-unique elements in dims")
    lps = [(x, tuple(n)) for x, n in factors]
    sizes = _sizes(lps)
    all_dims = _unify(lps)
    out_dims = [d for d in all_dims if d not in set(Ks_to_sum)]
    if path is None:
        path = greedy_path([n for _, n in lps], set(out_dims), sizes)

    steps = []
    for idxs in path:                                                 # reduce_Ks.py:270
        group = tuple(lps[i] for i in idxs)                           # :272
        lps = [lps[i] for i in range(len(lps)) if i not in idxs]      # :273
        remaining = set(_unify(lps))
        Ks = tuple(k for k in Ks_to_sum
                   if k not in remaining and k in _unify(group))      # :276
        steps.append((group, Ks))
        lps.append(logsumexp_sum(Ks, *group))                         # :280
    assert len(lps) == 1
    return lps[0], steps


def reduce_Ks(factors: Sequence[Factor], Ks_to_sum: Sequence[str], path=None):
    """reduce_Ks.py:236-244."""
    result, _ = collect_lps(factors, Ks_to_sum, path)
    return result


def reduce_Ks_bruteforce(factors: Sequence[Factor], Ks_to_sum: Sequence[str]):
    """Order-free check: fp64 torch.logsumexp over the full cross product."""
    x, names = broadcast_sum([(f.double(), n) for f, n in factors])
    axes = [names.index(k) for k in Ks_to_sum if k in names]
    if axes:
        x = t.logsumexp(x, axes)
    return x, tuple(n for n in names if n not in Ks_to_sum)


# --------------------------------------------------------------------------- a5
def plate_sum(factor: Factor, plate: str, prev: Factor = None):
    """logpq.py:149-153: ``lp.sum(plate)`` then ``prev_lpq + lp`` for Split chunks."""
    x, names = factor
    out = x.sum(names.index(plate))
    out_names = tuple(n for n in names if n != plate)
    if prev is not None:
        px, pn = prev
        assert set(pn) == set(out_names)                              # logpq.py:152
        out = align((px, pn), out_names) + out
    return out, out_names


def split_sizes(orig_size: int, split_size: int):
    """Split.py:84-95: chunk sizes, including the size-1-remainder borrow."""
    assert orig_size > split_size
    sizes = (orig_size // split_size) * [split_size]
    rem = orig_size % split_size
    if rem != 0:
        sizes.append(rem)
    if split_size > 2 and sizes[-1] == 1:
        sizes[-2] -= 1
        sizes[-1] += 1
    return sizes


# --------------------------------------------------------------------------- a8
def logmmexp(prev: t.Tensor, curr: t.Tensor):
    """utils.py:499-507."""
    pm = prev.amax(-1, keepdim=True)
    cm = curr.amax(-2, keepdim=True)
    lin = (prev - pm).exp() @ (curr - cm).exp()
    return (lin + t.finfo(lin.dtype).eps).log() + pm + cm


def chain_logmmexp(ms: t.Tensor):
    """utils.py:478-497,509-510: pairwise tree over the leading (time) axis; an odd
    leftover matrix is carried to the END of the next round's list."""
    assert ms.ndim == 3 and ms.shape[-2] == ms.shape[-1]
    while ms.shape[0] != 1:
        prev, curr = ms[::2], ms[1::2]
        rem = None
        if len(prev) > len(curr):
            rem, prev = prev[-1:], prev[:-1]
        ms = logmmexp(prev, curr)
        if rem is not None:
            ms = t.cat([ms, rem], 0)
    return ms[0]


def timeseries_plate(ms: t.Tensor):
    """logpq.py:132-143: chain over T then ``logsumexp(-1)`` -> [K_init]."""
    return t.logsumexp(chain_logmmexp(ms), -1)


# --------------------------------------------------------------------------- a10
def reduce_Ks_grads(factors: Sequence[Factor], Ks_to_sum, plate=None, grad_out=None):
    """Autograd through the restatement: d(sum(grad_out * result))/d factor_f."""
    leaves = [(x.detach().clone().requires_grad_(True), n) for x, n in factors]
    res = reduce_Ks(leaves, Ks_to_sum)
    if plate is not None:
        res = plate_sum(res, plate)
    out, _ = res
    if grad_out is None:
        grad_out = t.ones_like(out)
    gs = t.autograd.grad(out, [x for x, _ in leaves], grad_out)
    return res, list(gs)
