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
from . import native as N_
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


def _step_table(facs, now):
    """Sum of a step's factors laid out [batch dims..., the Ks drawn at this step (``now``)...]: the (unnormalised)
    log-weights the draw is taken from -- the ``lp`` of reduce_Ks.py:52-64.  -> (PT, number of K dims drawn)"""
    dims, ids = pt_order(facs, last=now)                       # the Ks to draw innermost
    total = None
    for f in facs:
        a = pt_align(f, ids)
        total = a if total is None else total + a
    total = total.expand([max(s) for s in zip(*[pt_align(f, ids).shape for f in facs])])
    return PT(total, dims), len(now)


def step_tables(lps, Ks):
    """The tables ``sample_Ks`` draws from, in drawing order, with NOTHING plugged in yet (every K drawn at an earlier
    step is still a dim of the later tables): what tests compare with the reference's pre-multinomial tables
    (tests/golden/posterior.pt).  -> [(Ks drawn, PT table)]"""
    _, _, record = E.contract([(lp.x, lp.dims) for lp in lps], tuple(Ks))
    out = []
    for group, now in reversed(record):
        if len(now):
            out.append((tuple(now), _step_table([PT(x, d) for x, d in group], now)[0]))
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
        table, nk = _step_table(facs, now)
        total, dims = table.x, table.dims
        flat = total.reshape(*total.shape[: len(dims) - nk], -1)
        probs = (flat - flat.amax(-1, keepdim=True)).exp()
        assert bool(t.isfinite(probs).all()) and bool((probs >= 0).all())
        batch = dims[: len(dims) - nk]
        if any(d is N_dim for d in batch):                          # one draw per (n, plates...) row
            draw = t.multinomial(probs.reshape(-1, probs.shape[-1]), 1, replacement=True)
            draw = draw.reshape(probs.shape[:-1])
            out_dims = list(batch)
        else:                                                       # N draws per (plates...) row
            draw = t.multinomial(probs.reshape(-1, probs.shape[-1]), N, replacement=True)
            draw = draw.reshape(*probs.shape[:-1], N).movedim(-1, 0)
            out_dims = [N_dim, *batch]
        sizes = [total.shape[len(dims) - nk + j] for j in range(nk)]
        for j in range(nk - 1, -1, -1):                             # unravel the joint index
            indices[id(now[j])] = (now[j], PT(draw % sizes[j], out_dims))
            draw = draw // sizes[j]
    return indices


TIMESERIES_POSTERIOR = "reference"
"""How ``importance_sample`` draws a timeseries variable's K indices.

"reference" (default since round 4: a drop-in draws from the reference's distribution): what reduce_Ks.py:85-232
evaluates -- every timestep drawn INDEPENDENTLY from the filtering marginal p(k_t | factors up to t), mixed over the N
sampled initial states.  (Its backward term is added on the K_init torchdim while the filtered term lives on the K
torchdim; after indexing K_init with the sampled initial states and summing over N it is a constant over k, so the
normalised "smoothed" table equals the filtered one -- tests/golden/posterior.pt records exactly that.)  One forward
recursion (alan_chain_filter) instead of the reference's O(T^2) chain evaluations; the trajectories are not joint draws.

"smoothing" (opt-in improvement): an exact joint draw of the whole trajectory from p(k_0..k_{T-1} | everything), by
backward messages + forward sampling (two launches, O(T K^2)); its moments agree with ``marginals()`` and the Kalman
smoother, which the reference's draws do not."""


def _timeseries_factor(lps, Ks, K_cur, K_init, T_dim, indices):
    """The plate's factors with its ordinary Ks summed out and every already-drawn K (but the initial state's) plugged
    in: a PT over (batch dims..., T, K_init, K_cur), batch = enclosing plates (+ N once a drawn K was plugged in)."""
    out, dims, _ = E.contract([(lp.x, lp.dims) for lp in lps], tuple(Ks))
    lp = PT(out, dims)
    core = (T_dim, K_init, K_cur)
    assert all(lp.has(d) for d in core)
    sub = {k: v for k, v in indices.items() if k != id(K_init)}
    (lp,) = _index_all([lp], sub)
    batch = tuple(d for d in lp.dims if not any(d is c for c in core))
    return lp, batch, core


def filtering_marginals(ms, init):
    """[C, T, K]: log softmax_k( LSE_n alpha_t[n, k] ) with alpha the forward recursion from init[n] -- the table the
    reference's sample_Ks_timeseries hands to t.multinomial at every timestep (see TIMESERIES_POSTERIOR)."""
    alpha = N_.chain_filter(ms, init)                            # [C, T, N, K]
    mix = t.logsumexp(alpha, 2)
    return mix - t.logsumexp(mix, -1, keepdim=True)


def sample_Ks_timeseries(lps, Ks, K_currs, K_inits, T_dim, indices, N_dim, N):
    """Posterior draw of a timeseries variable's K index at every timestep (role of sample_Ks_timeseries,
    reduce_Ks.py:85-232).  The plate may hold further latent groups (their Ks are summed out first and drawn afterwards
    by ``sample_Ks`` given the trajectory) and sit under other plates / parents' K dims (a batch of chains)."""
    assert len(K_currs) == 1 and len(K_inits) == 1, "one timeseries group per plate (logpq.py:136,140)"
    K_cur, K_init = K_currs[0], K_inits[0]
    assert id(K_init) in indices, "the initial state must have been sampled in the parent plate"
    lp, batch, core = _timeseries_factor(lps, Ks, K_cur, K_init, T_dim, indices)
    has_N = any(d is N_dim for d in batch)
    plates = tuple(d for d in batch if d is not N_dim)
    # chains laid out [N (if the factor depends on the sample), plates..., T, K_prev, K]
    lead = ((N_dim,) if has_N else ()) + plates
    ms = pt_align(lp, tuple(id(d) for d in (*lead, *core))).contiguous()
    T, K = ms.shape[-3], ms.shape[-1]
    B = 1
    for d in plates:
        B *= d.size
    init = pt_align(indices[id(K_init)][1], tuple(id(d) for d in (N_dim, *plates)))
    init = init.expand(N, *[d.size for d in plates]).reshape(N, B)
    flat = ms.reshape(-1, T, K, K)
    if ms.dtype != t.float32 or K > N_.POSTERIOR_MAX_K or not ms.is_cuda:
        # (fp64 factors, more than 128 particles, or not on the GPU -- the latter only under the test-only CPU backend,
        # whose seam is alan_reduce: the same two distributions, one alan_reduce per timestep)
        if TIMESERIES_POSTERIOR == "reference":
            logp = _filtering_by_steps(flat, init, N, B, has_N)                        # [B, T, K]
            d = t.multinomial(logp.exp().reshape(B * T, K).float(), N, replacement=True)
            draws = d.reshape(B, T, N).permute(2, 0, 1).contiguous()
        else:
            draws = _sample_chain_by_steps(flat, init, N, B, has_N)
    elif TIMESERIES_POSTERIOR == "reference" and has_N:
        # chains that depend on the sample (a drawn K of another group plugged in): the same table, the forward recursion
        # of sample n run on chain n from its own initial state, mixed over n
        fl = flat.reshape(N, B, T, K, K)
        tabs = []
        for b in range(B):
            alpha = t.stack([N_.chain_filter(fl[n, b:b + 1], init[n:n + 1, b].contiguous())[0, :, 0] for n in range(N)], 1)
            mix = t.logsumexp(alpha, 1)                                                  # [T, K]
            tabs.append(mix - t.logsumexp(mix, -1, keepdim=True))
        logp = t.stack(tabs)
        d = t.multinomial(logp.exp().reshape(B * T, K), N, replacement=True)
        draws = d.reshape(B, T, N).permute(2, 0, 1).contiguous()
    elif TIMESERIES_POSTERIOR == "reference":
        if B == 1 or bool((init == init[:, :1]).all()):
            logp = filtering_marginals(flat, init[:, 0].contiguous())                  # [B, T, K]
        else:                                                                           # initial states differ by plate
            logp = t.cat([filtering_marginals(flat[b:b + 1], init[:, b].contiguous()) for b in range(B)], 0)
        # N independent draws per (b, t)
        d = t.multinomial(logp.exp().reshape(B * T, K), N, replacement=True)            # [B*T, N]
        draws = d.reshape(B, T, N).permute(2, 0, 1).contiguous()
    else:
        beta = N_.chain_messages(flat)
        draws = N_.chain_sample(flat, beta, init, N, B, B if has_N else 0, 1)
    draws = draws.reshape(N, *[d.size for d in plates], T)
    return {id(K_cur): (K_cur, PT(draws, (N_dim, *plates, T_dim)))}


def _filtering_by_steps(flat, init, N, B, has_N):
    """The "reference" table (per-timestep filtering marginals mixed over the sampled initial states, normalised) with one
    alan_reduce per timestep: flat [C, T, K, K] (C = N B chains when the factor depends on the sample, else B), init
    [N, B] -> log-probabilities [B, T, K]."""
    C_, T, K, _ = flat.shape
    n_idx = t.arange(N, device=flat.device).unsqueeze(1).expand(N, B)
    b_idx = t.arange(B, device=flat.device).unsqueeze(0).expand(N, B)
    chain = (n_idx * B + b_idx) if has_N else b_idx                                       # [N, B]
    alpha = flat[chain, 0, init]                                                         # [N, B, K]: from each initial state
    tabs = []
    for step in range(T):
        if step:
            trans = flat[chain, step]                                                    # [N, B, K, K]
            alpha, _ = E.reduce_factors([(alpha.contiguous(), ("n", "b", "a")), (trans.contiguous(), ("n", "b", "a", "k"))],
                                        reduce=("a",))
        mix = t.logsumexp(alpha, 0)                                                      # [B, K]
        tabs.append(mix - t.logsumexp(mix, -1, keepdim=True))
    return t.stack(tabs, 1)


def _sample_chain_by_steps(flat, init, N, B, has_N):
    """The same draw with one alan_reduce per backward message and one torch.multinomial per step (fp64 factors, or
    more than 128 particles)."""
    C_, T, K, _ = flat.shape
    beta = [None] * (T + 1)
    beta[T] = t.zeros(C_, K, dtype=flat.dtype, device=flat.device)
    for step in range(T - 1, 0, -1):
        beta[step], _ = E.reduce_factors([(flat[:, step], ("c", "a", "b")), (beta[step + 1], ("c", "b"))], reduce=("b",))
    n_idx = t.arange(N, device=flat.device).unsqueeze(1).expand(N, B)
    b_idx = t.arange(B, device=flat.device).unsqueeze(0).expand(N, B)
    chain = (n_idx * B + b_idx) if has_N else b_idx
    prev = init
    draws = []
    for step in range(T):
        logits = flat[chain, step, prev] + beta[step + 1][chain]                         # [N, B, K]
        probs = (logits - logits.amax(-1, keepdim=True)).exp()
        prev = t.multinomial(probs.reshape(N * B, K), 1, replacement=True).reshape(N, B)
        draws.append(prev)
    return t.stack(draws, -1)


def logPQ_sample(name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims,
                 all_platedims, groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy,
                 indices, N_dim, N):
    platedim, active, scope2, lps, Ks, K_currs, K_inits = plate_factors(
        name, P, Q, sample, inputs_params, data, extra_log_factors, scope, active_platedims, all_platedims,
        groupvarname2Kdim, varname2groupvarname, sampler, computation_strategy, {})
    if K_currs:
        # the trajectory first (the plate's ordinary Ks summed out), then those Ks given the trajectory
        indices = {**indices, **sample_Ks_timeseries(lps, Ks, K_currs, K_inits, platedim, indices, N_dim, N)}
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
