"""
Posterior moments and marginals over the K particles -- the production users of the hot path's
BACKWARD (Sample.py:208-346, Marginals.py, moments.py of the reference): a zero "source term" J is
added as an extra log-factor and d ELBO / d J is read off.  With J on a group's K dim the gradient
is the posterior marginal over its particles; with f(x)*J it is E[f(x)].
"""
import torch as t

from .dims import Dim, dims_of, dim_to_named, is_tensor


class Moment:
    pass


class RawMoment(Moment):
    """E[f(x)] for a pointwise function f of one or more variables."""

    def __init__(self, f):
        self.f = f

    def from_marginals(self, samples, weights, all_platedims):
        plates = set(all_platedims.values())
        fx = self.f(*samples)
        Ks = tuple(d for d in dims_of(weights) if d not in plates)
        assert Ks and set(d for d in dims_of(fx) if d not in plates) <= set(Ks)
        return (fx * weights).sum(Ks)

    def from_samples(self, samples, Ndim):
        return self.f(*samples).mean(Ndim)

    def all_raw_moments(self):
        return [self]


class CompoundMoment(Moment):
    """A function of several raw moments (e.g. the variance)."""

    def __init__(self, combiner, raw_moments):
        assert all(isinstance(rm, RawMoment) for rm in raw_moments)
        self.combiner, self.raw_moments = combiner, list(raw_moments)

    def from_marginals(self, samples, weights, all_platedims):
        return self.combiner(*[rm.from_marginals(samples, weights, all_platedims) for rm in self.raw_moments])

    def from_samples(self, samples, Ndim):
        return self.combiner(*[rm.from_samples(samples, Ndim) for rm in self.raw_moments])

    def all_raw_moments(self):
        return self.raw_moments


def var_from_raw_moment(rm):
    assert isinstance(rm, RawMoment)
    rm2 = RawMoment(lambda *x: rm.f(*x) ** 2)
    return CompoundMoment(lambda ex, ex2: (ex2 - ex * ex).clamp(min=t.finfo(ex2.dtype).tiny), [rm, rm2])


mean = RawMoment(lambda x: x)
mean2 = RawMoment(t.square)
mean_log = RawMoment(t.log)
mean_log1m = RawMoment(lambda x: t.log(1 - x))
mean_recip = RawMoment(lambda x: 1 / x)
var = var_from_raw_moment(mean)


def uniformise_moment_args(args):
    """``(varname(s), moment)`` or ``([(varnames, moment), ...],)`` -> [(tuple varnames, moment)]"""
    err = Exception(".moments must be called as .moments(varname, moment) or .moments([(varnames, moment), ...])")
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        pairs = list(args[0])
    elif len(args) == 2:
        pairs = [(args[0], args[1])]
    else:
        raise err
    out = []
    for k, m in pairs:
        if not isinstance(k, (tuple, str)) or not isinstance(m, Moment):
            raise err
        out.append(((k,) if isinstance(k, str) else tuple(k), m))
    return out


class _MomentsAPI:
    """``_moments`` returns torchdim tensors, ``moments`` named tensors; single query -> single value."""

    def _moments(self, *args, **kwargs):
        res = self._moments_uniform_input(uniformise_moment_args(args), **kwargs)
        return res[0] if len(args) == 2 else res

    def moments(self, *args, **kwargs):
        res = [dim_to_named(x) if is_tensor(x) else x
               for x in self._moments_uniform_input(uniformise_moment_args(args), **kwargs)]
        return res[0] if len(args) == 2 else res


class Marginals(_MomentsAPI):
    """Pre-computed posterior marginals over the K particles of every latent group (and of any
    requested joint): moments of any function of those variables follow without another ELBO pass."""

    def __init__(self, samples, weights, all_platedims, varname2groupvarname):
        self.samples, self.weights = samples, weights
        self.all_platedims, self.varname2groupvarname = all_platedims, varname2groupvarname

    def _moments_uniform_input(self, moms):
        out = []
        for varnames, m in moms:
            xs = tuple(self.samples[v] for v in varnames)
            key = frozenset(self.varname2groupvarname[v] for v in varnames)
            out.append(m.from_marginals(xs, self.weights[key], self.all_platedims))
        return out

    def ess(self):
        plates = set(self.all_platedims.values())
        out = {}
        for key, w in self.weights.items():
            Ks = tuple(d for d in dims_of(w) if d not in plates)
            out[key] = 1 / (w ** 2).sum(Ks)
        return out

    def min_ess(self):
        vals = []
        for e in self.ess().values():
            ds = dims_of(e)
            vals.append(float((e.order(*ds) if ds else e).min()))
        return min(vals)
