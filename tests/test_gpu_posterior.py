"""Posterior K sampling pinned to the reference: what ``posterior.py`` hands to the random
draw against the tables the reference hands to ``t.multinomial`` (tests/golden/posterior.pt, recorded by running
reduce_Ks.py:35-83 and :85-232), plus the two-launch trajectory sampler against exact smoothing marginals."""
import pytest
import torch as t

from alan_amd import native as N, posterior as PS
from alan_amd.dims import PT, Dim
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _joint(steps, universe):
    """log p(all Ks | plates) implied by a sequence of conditional tables: sum of the per-step tables, each normalised
    over the Ks it draws.  steps = [(Ks names, table [names...], names)]; result laid out like ``universe``."""
    total = 0
    for Ks, tab, names in steps:
        axes = [names.index(k) for k in Ks]
        norm = tab.double() - t.logsumexp(tab.double(), axes, keepdim=True)
        perm = [names.index(n) for n in universe if n in names]
        norm = norm.permute(perm)
        total = total + norm[tuple(slice(None) if n in names else None for n in universe)]
    return total


@pytest.mark.parametrize("case", range(4))
def test_sample_Ks_draws_from_the_references_distribution(case):
    """The elimination ORDER (and so the split into conditionals) is the planner's; the joint distribution the
    conditionals multiply to is not: it must equal the one the reference's recorded tables multiply to."""
    c = load_golden("posterior.pt")["sample_Ks"][case]
    dims = {}
    lps = []
    for x, names in c["factors"]:
        ds = [dims.setdefault(n, Dim(n, s)) for n, s in zip(names, x.shape)]
        lps.append(PT(x.to(DEV), ds))
    mine = PS.step_tables(lps, [dims[k] for k in c["Ks"]])
    universe = []
    for x, names in c["factors"]:
        universe += [n for n in names if n not in universe]
    ref = _joint([(s["Ks"], s["table"], list(s["names"])) for s in c["steps"]], universe)
    got = _joint([(tuple(str(k) for k in now), tab.x.cpu(), [str(d) for d in tab.dims]) for now, tab in mine], universe)
    assert sorted(k for s in c["steps"] for k in s["Ks"]) == sorted(str(k) for now, _ in mine for k in now)
    t.testing.assert_close(got.expand(ref.shape), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("case", range(3))
def test_timeseries_reference_mode_matches_what_the_reference_draws_from(case):
    """alan_chain_filter + mixing over the sampled initial states + normalisation == the probabilities the reference
    passes to t.multinomial at every timestep of sample_Ks_timeseries (two enclosing-plate elements)."""
    c = load_golden("posterior.pt")["timeseries"][case]
    logp = PS.filtering_marginals(c["ms"].to(DEV), c["init"].to(DEV))                # [P, T, K]
    t.testing.assert_close(logp.exp().cpu(), c["probs"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("K,T,B,per_sample", [(3, 4, 1, False), (10, 25, 3, False), (30, 200, 2, False), (7, 12, 2, True),
                                              (100, 9, 1, False)])
def test_two_launch_trajectory_sampler_against_exact_smoothing_marginals(K, T, B, per_sample):
    """alan_chain_messages + alan_chain_sample: the per-timestep frequencies of the drawn trajectories against the exact
    smoothing marginals p(k_t | all factors, k_init) (forward x backward in fp64), within 5 standard errors; batches
    of chains, and chains that differ per sample."""
    g = t.Generator().manual_seed(K * 100 + T)
    Nsamp = 40000
    C = Nsamp * B if per_sample else B
    if per_sample:
        base = -0.5 * t.randn(B, T, K, K, generator=g) ** 2
        ms = (base.unsqueeze(0) + 0.0 * t.zeros(Nsamp, 1, 1, 1, 1)).reshape(C, T, K, K).contiguous()   # same chains, stored per sample
    else:
        ms = -0.5 * t.randn(C, T, K, K, generator=g) ** 2 + 0.5 * t.randn(C, T, 1, K, generator=g)
    init = t.randint(0, K, (1, B), generator=g).expand(Nsamp, B).contiguous()
    dev = ms.to(DEV)
    beta = N.chain_messages(dev)
    gen = t.Generator(device=DEV).manual_seed(1)
    draws = N.chain_sample(dev, beta, init.to(DEV), Nsamp, B, B if per_sample else 0, 1, generator=gen).cpu()
    assert draws.shape == (Nsamp, B, T) and int(draws.min()) >= 0 and int(draws.max()) < K
    m64 = (ms.reshape(Nsamp, B, T, K, K)[0] if per_sample else ms).double()
    for b in range(B):
        M = m64[b]
        alpha = [M[0, int(init[0, b])]]
        for step in range(1, T):
            alpha.append(t.logsumexp(alpha[-1][:, None] + M[step], 0))
        bt = [None] * (T + 1)
        bt[T] = t.zeros(K, dtype=t.float64)
        for step in range(T - 1, 0, -1):
            bt[step] = t.logsumexp(M[step] + bt[step + 1][None, :], 1)
        t.testing.assert_close(beta[b if not per_sample else b].cpu().double()[1:], t.stack(bt[1:]), rtol=1e-4, atol=1e-4)
        for step in (0, T // 2, T - 1):
            p = t.softmax(alpha[step] + bt[step + 1], 0)
            freq = t.bincount(draws[:, b, step], minlength=K).double() / Nsamp
            se = (p * (1 - p) / Nsamp).sqrt() + 1e-4
            assert bool(((freq - p).abs() <= 5 * se).all()), (b, step, (freq - p).abs().max())
    # reproducible from the generator's seed
    gen = t.Generator(device=DEV).manual_seed(1)
    again = N.chain_sample(dev, beta, init.to(DEV), Nsamp, B, B if per_sample else 0, 1, generator=gen).cpu()
    assert t.equal(again, draws)
