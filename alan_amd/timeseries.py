"""
``Timeseries(init, trans)``: a Markov chain along the innermost active plate (Timeseries.py of the
reference).  Sampling is sequential over T on the host; ``log_prob`` produces the
[T, K_init, K] transition factor that the timeseries plate reduces with the HIP chain kernel
(logpq.py:131-143 -> alan_chain_logmmexp).
"""
import torch as t
import torch.nn as nn

from .dims import PT, ShiftPT, Dim, dims_of
from .dist import _DistSpec


class Timeseries(nn.Module):
    is_timeseries = True
    qem_dist = False
    opt_dist = False

    def __init__(self, init, trans):
        super().__init__()
        if not isinstance(init, str):
            raise Exception("the first / `init` argument in a Timeseries should be a string, representing a "
                            "variable name in the above plate")
        if not isinstance(trans, _DistSpec):
            raise Exception("the second / `trans` argument in a Timeseries should be a distribution")
        if t.Size(trans.sample_shape) != t.Size([]):
            raise Exception("sample_shape on the transition distribution must not be set; if you want a "
                            "sample_shape, it needs to be on the initial state")
        self.init = init
        self.trans = trans.finalize(None)
        self.all_args = [init, *self.trans.all_args]
        self.sample_shape = t.Size([])

    def finalize(self, varname):
        self.varname = varname
        return self

    @property
    def opt_qem_params(self):
        return self.trans.opt_qem_params

    @staticmethod
    def _at_time(scope, T_dim, time):
        return {k: (v.order(T_dim)[time] if T_dim in set(dims_of(v)) else v) for k, v in scope.items()}

    def sample(self, scope, reparam, active_platedims, K_dim, timeseries_perm, dimcache=None):
        """Roll the chain forward T steps; between steps the particles are re-paired by
        ``timeseries_perm`` (one permutation of K per timestep), as Timeseries.py:89-123 does.
        (Sequential over T on the host, in torchdim: sampling is not on the hot path.)"""
        scope = {k: (v.dim() if isinstance(v, PT) else v) for k, v in scope.items()}
        *other, T_dim = active_platedims
        prev = scope[self.init]
        if set(dims_of(prev)) != {K_dim, *other}:
            raise Exception(f"Initial state, {self.init}, doesn't have the right dimensions for the timeseries; "
                            "the initial state must be defined one step up in the plate hierarchy")
        order = [K_dim, *other]
        steps = []
        for time in range(T_dim.size):
            local = self._at_time(scope, T_dim, time)
            local["prev"] = prev
            x = self.trans.sample(local, reparam, other, K_dim, None).dim()
            steps.append(x.order(*order))
            if timeseries_perm is not None:
                perm = timeseries_perm.order(T_dim)[time]
                x = x.order(K_dim)[perm, ...][K_dim]
            prev = x
        stacked = t.stack(steps, 0)                      # [T, K, *other, ...event]
        return stacked[(T_dim, *order)]

    def log_prob(self, sample, scope, T_dim, K_dim, dim_order=None, dimcache=None):
        """Returns (lp[T, K_init, K, ...] as a PT, K_init): the previous state is indexed by the K dim of
        the initial-state variable (Timeseries.py:205-245)."""
        assert isinstance(T_dim, Dim) and isinstance(K_dim, Dim)
        if isinstance(sample, PT):
            sample = sample.dim()
        scope = {k: (v.dim() if (isinstance(v, PT) and k == self.init) else v) for k, v in scope.items()}
        sdims = set(dims_of(sample))
        assert K_dim in sdims and T_dim in sdims
        init = scope[self.init]
        idims = set(dims_of(init))
        assert T_dim not in idims and len(idims) + 1 == len(sdims)
        (Kinit,) = list(idims - sdims)
        # previous state: x_{t-1}, re-labelled onto K_init; x_0's predecessor is the initial state
        lead = [d for d in dims_of(init)]
        init_pos = init.order(*lead)
        from . import dist as D
        full_pos = sample.order(K_dim)[Kinit].order(*lead, T_dim)          # the series on K_init: [*lead, T, ...]
        scope = dict(scope)
        if D.LAZY_TRANSITION and full_pos.is_cuda and not (t.is_grad_enabled() and (full_pos.requires_grad or init_pos.requires_grad)):
            # not concatenated: the chain's first round reads the two sources (dims.ShiftPT); anyone else gets the cat
            scope["prev"] = ShiftPT(init_pos, full_pos, len(lead), (*lead, T_dim))
        else:
            prev = t.cat([init_pos.unsqueeze(len(lead)), full_pos.narrow(len(lead), 0, full_pos.shape[len(lead)] - 1)], len(lead))
            scope["prev"] = PT(prev, (*lead, T_dim))
        scope[self.init] = PT.of(init)
        order = None
        if dim_order is not None:
            lead, last = dim_order
            order = ([d for d in lead if d not in {Kinit, K_dim}], [Kinit, K_dim])
        # (the factor may come back unevaluated: the chain's first round then computes it on load)
        lp, _ = self.trans.log_prob(PT.of(sample), scope, dim_order=order, dimcache=dimcache, unevaluated_ok=True)
        assert lp.has(Kinit) and lp.has(K_dim) and lp.has(T_dim)
        return lp, Kinit
