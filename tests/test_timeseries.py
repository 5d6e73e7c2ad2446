"""Timeseries plate end to end (BASELINE config C5): a linear-Gaussian state-space model whose log
evidence is known in closed form (the Kalman construction of the reference's tests/timeseries.py:5-56,
restated), evaluated through alan_amd's Timeseries + the HIP chain kernel.  Acceptance band follows
tests/test_problem_vs_itself.py:160-205: the mean of several large-K ELBOs brackets the truth."""
import math

import pytest
import torch as t

import alan_amd as alan
from alan_amd import Normal, Timeseries, Plate, BoundPlate, Problem, Data

A, INIT_SCALE, NOISE, OBS = 0.9, 1.0, 0.1, 1.0


def kalman_problem(T, seed=0):
    g = t.Generator().manual_seed(seed)
    cov = t.zeros(T, T, dtype=t.float64)
    var = INIT_SCALE ** 2
    for i in range(T):
        var = var * A ** 2 + NOISE ** 2
        fut = var * A ** t.arange(T - i, dtype=t.float64)
        cov[i, i:] = fut
        cov[i:, i] = fut
    total = cov + OBS ** 2 * t.eye(T, dtype=t.float64)
    L = t.linalg.cholesky(total)
    y = L @ t.randn(T, generator=g, dtype=t.float64)
    known = t.distributions.MultivariateNormal(t.zeros(T, dtype=t.float64), scale_tril=L).log_prob(y)
    P = Plate(init=Normal(0, INIT_SCALE),
              T=Plate(ts=Timeseries("init", Normal(lambda prev: A * prev, NOISE)), obs=Normal("ts", OBS)))
    Q = Plate(init=Normal(0, 1), T=Plate(ts=Normal(0, 1), obs=Data()))
    sizes = {"T": T}
    prob = Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes), {"obs": y.float().refine_names("T")})
    return prob, float(known)


def _band(prob, known, K, iters, device):
    prob.to(device)
    vals = []
    for i in range(iters):
        t.manual_seed(100 + i)
        vals.append(float(prob.sample(K, reparam=False).elbo_nograd(alan.no_checkpoint)))
    v = t.tensor(vals, dtype=t.float64)
    stderr = float(v.std() / math.sqrt(iters))
    return float(v.mean()), stderr, vals


def test_timeseries_elbo_host_logic(oracle_backend):
    prob, known = kalman_problem(4)
    mean, stderr, vals = _band(prob, known, 300, 8, "cpu")
    assert mean - 6 * stderr - 0.05 < known          # the ELBO is a lower bound in expectation
    assert known < mean + 6 * stderr + 0.5, (known, mean, stderr)


@pytest.mark.gpu
def test_timeseries_elbo_ground_truth_gpu():
    prob, known = kalman_problem(4)
    mean, stderr, vals = _band(prob, known, 1000, 20, "cuda")
    assert mean - 6 * stderr - 0.02 < known
    assert known < mean + 6 * stderr + 0.2, (known, mean, stderr)


@pytest.mark.gpu
def test_timeseries_T1000_K30_runs_and_matches_oracle_chain():
    """C5 size: T=1000, K=30.  The importance-weighted estimate is loose at this T, so the check is
    structural: same sample through the HIP chain and through the CPU oracle chain agree."""
    from oracle import backend
    prob, known = kalman_problem(1000)
    prob.to("cuda")
    t.manual_seed(5)
    sample = prob.sample(30, reparam=False)
    gpu = float(sample.elbo_nograd(alan.no_checkpoint))
    assert math.isfinite(gpu)
    # move the identical sample tree to the CPU and evaluate with the oracle backend
    import models
    from alan_amd.dims import dims_of
    cpu_prob, _ = kalman_problem(1000)
    Kd = {g: alan.dims.Dim(str(d), 30) for g, d in sample.groupvarname2Kdim.items()}
    by = {**{str(d): d for d in Kd.values()}, **{n: d for n, d in cpu_prob.all_platedims.items()}}

    def cpu_tree(tree):
        out = {}
        for k, v in tree.items():
            if isinstance(v, dict):
                out[k] = cpu_tree(v)
            else:
                ds = dims_of(v)
                pos = v.order(*ds).cpu()
                out[k] = pos[tuple(by[str(d)] for d in ds)]
        return out

    cs = alan.Sample(problem=cpu_prob, sample=cpu_tree(sample.detached_sample), groupvarname2Kdim=Kd,
                     sampler=alan.PermutationSampler, reparam=False)
    with backend.installed():
        cpu = float(cs.elbo_nograd(alan.no_checkpoint))
    assert abs(gpu - cpu) <= 1e-4 * abs(cpu) + 1e-3, (gpu, cpu)


def _ts_posterior_check(device, K, N):
    """Timeseries posterior: marginals (chain backward) and importance samples (sample_Ks_timeseries) vs
    each other and vs the closed-form Kalman smoother mean (tests/timeseries.py:52-56 of the reference)."""
    from alan_amd import mean
    from alan_amd.dims import dims_of
    T = 4
    prob, _ = kalman_problem(T)
    prob.to(device)
    y = prob._data.to_dict()["obs"].rename(None).double().cpu()
    cov = t.zeros(T, T, dtype=t.float64)
    var = INIT_SCALE ** 2
    for i in range(T):
        var = var * A ** 2 + NOISE ** 2
        fut = var * A ** t.arange(T - i, dtype=t.float64)
        cov[i, i:] = fut
        cov[i:, i] = fut
    post_cov = t.inverse(t.inverse(cov) + t.eye(T, dtype=t.float64) / OBS ** 2)
    post_mean = post_cov @ (y / OBS ** 2)
    t.manual_seed(3)
    sample = prob.sample(K, reparam=False)
    marg = sample.marginals()
    m_ts = marg._moments("ts", mean)
    v_ts = marg._moments("ts", alan.var_from_raw_moment(mean))
    (Td,) = dims_of(m_ts)
    m_ts, v_ts = m_ts.order(Td).cpu().double(), v_ts.order(Td).cpu().double()
    ess = marg.min_ess()
    assert bool(((m_ts - post_mean).abs() < 7 * (v_ts / ess).sqrt() + 0.05).all()), (m_ts, post_mean, ess)
    isamp = sample.importance_sample(N)
    est = isamp._moments("ts", mean).order(Td).cpu().double()
    assert bool(((est - m_ts).abs() < 6 * (v_ts / N).sqrt() + 1e-3).all()), (est, m_ts)
    d = isamp.dump()["ts"]
    assert set(d.names) == {"N", "T"}


def test_timeseries_posterior_host_logic(oracle_backend):
    _ts_posterior_check("cpu", 60, 4000)


@pytest.mark.gpu
def test_timeseries_posterior_gpu():
    _ts_posterior_check("cuda", 300, 20000)
