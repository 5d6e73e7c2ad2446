"""
How a child group picks which parent particles to condition on when Q is not factorised
(Sampler.py of the reference).  ``resample_scope`` re-indexes every parent from its own K dim onto
the child's K dim; ``reduce_logQ`` averages Q's log-prob over the parent K dims (the mixture
proposal) -- that average is a log-mean-exp and runs on the HIP engine (utils.py:224-225).
"""
import math

import torch as t

from . import engine as E
from .contract import logmeanexp_dims
from .dims import PT, Dim, dims_of, pt_align


class Sampler:
    @classmethod
    def resample_scope_pt(cls, scope, active_platedims, Kdim):
        """PT version of ``resample_scope``: every parent variable (a PT carrying its own K dim) is
        re-indexed along that K axis by an index tensor drawn by ``perm_pt`` and relabelled onto ``Kdim``;
        variables of one Group share one index tensor (Sampler.py:86-115)."""
        plates = {id(d) for d in active_platedims}
        by_K, out = {}, {}
        for name, p in scope.items():
            Ks = [d for d in p.dims if id(d) not in plates]
            assert len(Ks) <= 1, f"{name} carries several K dims: {Ks}"
            if not Ks:
                out[name] = p
            else:
                by_K.setdefault(id(Ks[0]), (Ks[0], {}))[1][name] = p
        for K_var, group in by_K.values():
            first = next(iter(group.values()))
            for p in group.values():
                assert set(p.ids) == set(first.ids)
            ax = first.ids.index(id(K_var))
            shape = list(first.x.shape[: len(first.dims)])
            idx = cls.perm_pt(shape, ax, first.x.device)          # [*first-class shape], values in [0, K)
            for name, p in group.items():
                if p.ids != first.ids:                              # same dims, different storage order
                    p = PT(pt_align(p, first.ids), first.dims)
                ix = idx[(...,) + (None,) * p.n_pos].expand(p.x.shape) if p.n_pos else idx
                x = t.gather(p.x, ax, ix)
                out[name] = PT(x, (*p.dims[:ax], Kdim, *p.dims[ax + 1:]))
        return out

    @classmethod
    def resample_scope(cls, scope, active_platedims, Kdim):
        by_K = {}
        for name, x in scope.items():
            Ks = [d for d in dims_of(x) if d not in set(active_platedims)]
            assert len(Ks) <= 1, f"{name} carries several K dims: {Ks}"
            by_K.setdefault(Ks[0] if Ks else None, {})[name] = x
        out = {}
        for K_var, group in by_K.items():
            if K_var is None:
                out.update(group)
                continue
            first = next(iter(group.values()))
            for x in group.values():        # variables of one Group share dims exactly
                assert set(dims_of(x)) == set(dims_of(first))
            perm = cls.perm(dims=set(dims_of(first)), Kdim=K_var)
            for name, x in group.items():
                out[name] = x.order(K_var)[perm, ...][Kdim]
        allowed = {Kdim, *active_platedims}
        for x in out.values():
            assert set(dims_of(x)) <= allowed
        return out


class SamplerMP(Sampler):
    @staticmethod
    def reduce_logQ(lq, active_platedims, Kdim):
        """[plates, parent Ks, K] -> [plates, K]: log of the mean over parent particles
        (Sampler.py:118-134).  Accepts / returns a PT (the plate recursion) or a torchdim tensor."""
        if not isinstance(lq, PT):
            parents = tuple(d for d in dims_of(lq) if d is not Kdim and d not in set(active_platedims))
            return logmeanexp_dims(lq, parents)
        keep = {id(Kdim), *(id(d) for d in active_platedims)}
        parents = tuple(d for d in lq.dims if id(d) not in keep)
        if not parents:
            return lq
        assert lq.n_pos == 0
        c = -sum(math.log(lq.size_of(id(d))) for d in parents)
        out, dims = E.reduce_factors([(lq.x, lq.dims)], reduce=parents, add_const=c)
        return PT(out, dims)


def _like(dims, Kdim):
    others = [d for d in dims if d is not Kdim]
    return others, [d.size for d in others]


class PermutationSampler(SamplerMP):
    """Every parent particle gets exactly one child: an independent random permutation of K for each
    element of the other dims."""

    @staticmethod
    def perm_pt(shape, axis, device):
        return t.rand(shape, device=device).argsort(axis)

    @staticmethod
    def perm(dims, Kdim):
        assert isinstance(dims, set) and isinstance(Kdim, Dim)
        others, sizes = _like(dims, Kdim)
        keys = t.rand(Kdim.size, *sizes, device=_device())
        idx = keys.argsort(0)
        return idx[(slice(None), *others)] if others else idx


class CategoricalSampler(SamplerMP):
    """Each child picks a parent particle uniformly at random (with replacement)."""

    @staticmethod
    def perm_pt(shape, axis, device):
        return t.randint(0, shape[axis], shape, device=device)

    @staticmethod
    def perm(dims, Kdim):
        assert isinstance(dims, set) and isinstance(Kdim, Dim)
        others, sizes = _like(dims, Kdim)
        idx = t.randint(0, Kdim.size, (Kdim.size, *sizes), device=_device())
        return idx[(slice(None), *others)] if others else idx


class IndependentSampler(Sampler):
    @staticmethod
    def perm_pt(shape, axis, device):
        view = [1] * len(shape)
        view[axis] = shape[axis]
        return t.arange(shape[axis], device=device).view(view).expand(shape)

    @staticmethod
    def perm(dims, Kdim):
        return t.arange(Kdim.size, device=_device())


_DEVICE = [t.device("cpu")]


def _device():
    return _DEVICE[0]


class on_device:
    """Context manager: samplers draw their index tensors on this device."""

    def __init__(self, device):
        self.device = t.device(device)

    def __enter__(self):
        self.prev = _DEVICE[0]
        _DEVICE[0] = self.device

    def __exit__(self, *a):
        _DEVICE[0] = self.prev
