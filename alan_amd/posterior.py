"""
Posterior sampling over the K particles (``Sample.importance_sample``): sample_logpq.py:17-107 and
reduce_Ks.py:35-83 (``sample_Ks``) of the reference, on PTs.

Top-down over the plate tree.  At each plate the factors are gathered exactly as for the ELBO (child
plates fully reduced on the HIP engine), indexed with the K indices already drawn higher up, and the
plate's own K dims are then drawn by walking the elimination steps of ``engine.contract`` BACKWARDS:
the factors of a step, conditioned on everything sampled so far, are a categorical over the K dims that
step eliminated.  Timeseries K dims are drawn by a filter / sample pass over T (``sample_Ks_timeseries``).
"""
import torch as t

from . import engine as E
from .dims import PT, Dim, pt_align, pt_order
from .logpq import plate_factors
from .model import Plate


def _take(p, Kdim, idx):
    """p[..., K, ...] indexed along K by idx (a PT of ints over (N, plates...)) -> PT over
    (idx dims) U (p dims - K)."""
    rest = [d for d in p.dims if d is not Kdim]
    lead = list(idx.dims)
    lead_ids = set(idx.ids)
    tail = [d for d in rest if id(d) not in lead_ids]
    # source laid out [K, lead (1 where p lacks it)..., tail...]; index laid out [lead..., 1...]
    order = (id(Kdim), *[id(d) for d in lead], *[id(d) for d in tail])
    src = pt_align(p, order)
    shape = [src.shape[0], *[idx.x.shape[i] for i in range(len(lead))], *src.shape[1 + len(lead):]]
    src = src.expand(shape)
    ix = idx.x[(...,) + (None,) * len(tail)].expand(shape[1:]).unsqueeze(0)
    out = t.gather(src, 0, ix).squeeze(0)
    return PT(out, (*lead, *tail))


def _index_all(lps, indices):
    out = []
    for lp in lps:
        for d in list(lp.dims):
            if id(d) in indices:
                lp = _take(lp, d, indices[id(d)][1])
        out.append(lp)
    return out


def sample_Ks(lps, Ks, N_dim, N):
    """Draw the K dims ``Ks`` of one plate jointly from their posterior given the factors ``lps`` (whose
    already-sampled K dims have been indexed away).  Returns {id(K): (K, PT index over (N, plates...))}."""
    assert all(lp.n_pos == 0 for lp in lps)
    _, _, record = E.contract([(lp.x, lp.dims) for lp in lps], tuple(Ks))
    indices = {}
    for group, now in reversed(record):
        if not len(now):
            continue
        facs = _index_all([PT(x, d) for x, d in group], indices)
        dims, ids = pt_order(facs, last=now)                       # the Ks to draw innermost
        total = None
        for f in facs:
            a = pt_align(f, ids)
            total = a if total is None else total + a
        nk = len(now)
        total = total.expand([max(s) for s in zip(*[pt_align(f, ids).shape for f in facs])])
        flat = total.reshape(*total.shape[: len(ids) - nk], -1)
        probs = (flat - flat.amax(-1, keepdim=True)).exp()
        assert bool(t.isfinite(probs).all()) and bool((probs >= 0).all())
        batch = dims[: len(ids) - nk]
        if any(d is N_dim for d in batch):                          # one draw per (n, plates...) row
            draw = t.multinomial(probs.reshape(-1, probs.shape[-1]), 1, replacement=True)
            draw = draw.reshape(probs.shape[:-1])
            out_dims = list(batch)
        else:                                                       # N draws per (plates...) row
            draw = t.multinomial(probs.reshape(-1, probs.shape[-1]), N, replacement=True)
            draw = draw.reshape(*probs.shape[:-1], N).movedim(-1, 0)
            out_dims = [N_dim, *batch]
        sizes = [total.shape[len(ids) - nk + j] for j in range(nk)]
        for j in range(nk - 1, -1, -1):                             # unravel the joint index
            indices[id(now[j])] = (now[j], PT(draw % sizes[j], out_dims))
            draw = draw // sizes[j]
    return indices


def sample_Ks_timeseries(lps, Ks, K_currs, K_inits, T_dim, indices, N_dim, N):
    """Posterior draw of a timeseries variable's K index at every timestep (role of
    sample_Ks_timeseries, reduce_Ks.py:85-232) by forward-filtering / backward-sampling in the
    reversed direction: backward messages beta_t[a] = LSE_b(M_{t+1}[a,b] + beta_{t+1}[b]) (one alan_reduce
    per step), then k_t ~ softmax_b(M_t[k_{t-1}, b] + beta_t[b]) forwards from the already-sampled
    initial-state index.  O(T K^2) instead of the reference's O(T^2) chain evaluations."""
    if len(Ks) or len(K_currs) != 1:
        raise NotImplementedError("alan_amd: a timeseries plate with further latent groups cannot be "
                                  "posterior-sampled yet")
    K_cur, K_init = K_currs[0], K_inits[0]
    assert id(K_init) in indices, "the initial state must have been sampled in the parent plate"
    out, dims, _ = E.contract([(lp.x, lp.dims) for lp in lps], ())
    lp = PT(out, dims)
    want = (id(T_dim), id(K_init), id(K_cur))
    if set(lp.ids) != set(want):
        raise NotImplementedError("alan_amd: timeseries plates nested under other K/plate dims are not supported")
    ms = pt_align(lp, want).contiguous()                         # [T, K_prev, K]
    T, K = ms.shape[0], ms.shape[2]
    beta = [None] * (T + 1)
    beta[T] = t.zeros(K, dtype=ms.dtype, device=ms.device)
    for step in range(T - 1, 0, -1):
        beta[step], _ = E.reduce_factors([(ms[step], ("a", "b")), (beta[step + 1], ("b",))], reduce=("b",))
    prev = pt_align(indices[id(K_init)][1], (id(N_dim),))        # [N] indices of the initial state
    assert prev.ndim == 1
    draws = []
    for step in range(T):
        logits = ms[step][prev] + beta[step + 1]                 # [N, K]
        probs = (logits - logits.amax(-1, keepdim=True)).exp()
        prev = t.multinomial(probs, 1, replacement=True).squeeze(-1)
        draws.append(prev)
    return {id(K_cur): (K_cur, PT(t.stack(draws, 1), (N_dim, T_dim)))}


def logPQ_sample(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                 all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                 indices, N_dim, N):
    platedim, active, scope2, lps, Ks, K_currs, K_inits = plate_factors(
        name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims, all_platedims,
        groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy, {})
    if K_currs:
        indices = {**indices, **sample_Ks_timeseries(lps, Ks, K_currs, K_inits, platedim, indices, N_dim, N)}
        Ks = ()
    lps = _index_all(lps, indices)
    if Ks:
        indices = {**indices, **sample_Ks(lps, Ks, N_dim, N)}
    for kind, child, q in Q.entries():
        if kind == "plate":
            indices = logPQ_sample(
                name=child, P=P.flat_prog[child], Q=q, sample=sample[child],
                inputs_params=inputs_params.get(child, {}), data=data.get(child, {}),
                extra_log_factors=extra_log_factors.get(child, {}), scope=scope2, active_platedims=active,
                all_platedims=all_platedims, groupvarname2Kdim=groupvarname2Kdim,
                varname2groupvarname=varname2groupvarname, sampler=sampler,
                computation_strategy=computation_strategy, indices=indices, N_dim=N_dim, N=N)
    return indices


def index_into_sample(tree, indices, groupvarname2Kdim, varname2groupvarname):
    """Replace every variable's K dim by N using the drawn indices (Sample.py:358-381)."""
    out = {}
    for name, v in tree.items():
        if isinstance(v, dict):
            out[name] = index_into_sample(v, indices, groupvarname2Kdim, varname2groupvarname)
        else:
            K = groupvarname2Kdim[varname2groupvarname[name]]
            src = v.detach()
            idx = indices[id(K)][1]
            # positional (event) dims ride along as anonymous tail dims
            ev = [Dim(f"_e{i}", s) for i, s in enumerate(src.x.shape[len(src.dims):])]
            taken = _take(PT(src.x, (*src.dims, *ev)), K, idx)
            keep = [d for d in taken.dims if all(d is not e for e in ev)]
            x = pt_align(taken, tuple(id(d) for d in (*keep, *ev)))
            out[name] = PT(x, keep)
    return out


class ImportanceSample:
    """N joint posterior samples of every latent (ImportanceSample.py:25-39): ``samples_flatdict`` maps
    variable name -> torchdim tensor carrying the ``N`` dim."""

    def __init__(self, problem, samples_tree, Ndim):
        from .model import flatten_tree
        self.problem, self.Ndim = problem, Ndim
        self.samples_tree = {k: v for k, v in samples_tree.items()}
        self.samples_flatdict = {k: v.dim() for k, v in flatten_tree(samples_tree).items()}

    def dump(self):
        from .dims import dim_to_named
        return {k: dim_to_named(v) for k, v in self.samples_flatdict.items()}

    def _moments_uniform_input(self, moms):
        return [m.from_samples(tuple(self.samples_flatdict[v] for v in varnames), self.Ndim)
                for varnames, m in moms]

    def _moments(self, *args):
        from .moments import _MomentsAPI
        return _MomentsAPI._moments(self, *args)

    def moments(self, *args):
        from .moments import _MomentsAPI
        return _MomentsAPI.moments(self, *args)
