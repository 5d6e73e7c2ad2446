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
    cpu_prob, _ = kalman_problem(1000)
    cs = _same_sample_on_cpu(sample, cpu_prob, 30)
    with backend.installed():
        cpu = float(cs.elbo_nograd(alan.no_checkpoint))
    assert abs(gpu - cpu) <= 1e-4 * abs(cpu) + 1e-3, (gpu, cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("T,K", [(1000, 30), (37, 16), (64, 32)])
def test_final_contraction_inside_the_chains_last_launch_is_the_same_elbo(T, K, monkeypatch):
    """native.CHAIN_FINAL: the evaluation's last log-sum-exp (over K_init, of the chain's result plus the initial state's
    factor) run by the chain's last launch behind its last round -- against the separate launch, eagerly and as a replayed
    graph (through the result ring), and it is actually taken."""
    from alan_amd import native as N
    prob, _ = kalman_problem(T)
    prob.to("cuda")
    t.manual_seed(7)
    sample = prob.sample(K, reparam=False)
    monkeypatch.setattr(N, "CHAIN_FINAL", False)
    separate = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    monkeypatch.setattr(N, "CHAIN_FINAL", True)
    taken = []
    real = N._PendingChain.try_final
    monkeypatch.setattr(N._PendingChain, "try_final", lambda self, *a: (lambda r: (taken.append(r), r)[1])(real(self, *a)))
    eager = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    assert taken and taken[-1] is True, taken
    replayed = [float(sample.elbo_nograd(alan.no_checkpoint, graph=True)) for _ in range(4)]
    for got in (eager, *replayed):
        assert abs(got - separate) <= 2e-6 * abs(separate) + 1e-5, (got, separate)


@pytest.fixture
def smoothing(monkeypatch):
    """posterior.TIMESERIES_POSTERIOR = "smoothing": exact joint trajectories -- what agrees with marginals() and the
    Kalman smoother (the default, "reference", draws the reference's per-timestep filtering marginals, which do not)."""
    from alan_amd import posterior as PS
    monkeypatch.setattr(PS, "TIMESERIES_POSTERIOR", "smoothing")


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


def test_timeseries_posterior_host_logic(oracle_backend, smoothing):
    _ts_posterior_check("cpu", 60, 4000)


@pytest.mark.gpu
def test_timeseries_posterior_gpu(smoothing):
    _ts_posterior_check("cuda", 300, 20000)


# ---------------------------------------------------------------------------------------------------------------
# timeseries plates NESTED under another plate and depending on a latent of a higher plate (the shape of the
# reference's examples/models/covid/covid.py:55-75): lp keeps the enclosing plate and the parent's K as batch dims of
# the chain (logpq.py:133-139).  Everything is linear-Gaussian, so the evidence is a multivariate Normal.
DRIFT, REGIONS = 0.3, 3


def nested_problem(T, seed=0, opt=False):
    """drift ~ N(0, DRIFT);  region r: init_r ~ N(0, INIT_SCALE), ts_rt ~ N(A ts_r,t-1 + drift, NOISE), obs ~ N(ts, OBS)"""
    R = REGIONS
    g = t.Generator().manual_seed(seed)
    # ts as a linear map of z = (drift, init_1..R, eps_11..eps_RT), by running the recursion on basis vectors
    nz = 1 + R + R * T
    M = t.zeros(R, T, nz, dtype=t.float64)
    for r in range(R):
        prev = t.zeros(nz, dtype=t.float64)
        prev[1 + r] = 1.0
        for i in range(T):
            cur = A * prev
            cur[0] += 1.0
            cur[1 + R + r * T + i] += 1.0
            M[r, i] = cur
            prev = cur
    var = t.cat([t.tensor([DRIFT ** 2]), t.full((R,), INIT_SCALE ** 2), t.full((R * T,), NOISE ** 2)]).double()
    Mf = M.reshape(R * T, nz)
    total = Mf @ t.diag(var) @ Mf.T + OBS ** 2 * t.eye(R * T, dtype=t.float64)
    L = t.linalg.cholesky(total)
    y = L @ t.randn(R * T, generator=g, dtype=t.float64)
    known = t.distributions.MultivariateNormal(t.zeros(R * T, dtype=t.float64), scale_tril=L).log_prob(y)
    P = Plate(drift=Normal(0, DRIFT),
              R=Plate(init=Normal(0, INIT_SCALE),
                      T=Plate(ts=Timeseries("init", Normal(lambda prev, drift: A * prev + drift, NOISE)),
                              obs=Normal("ts", OBS))))
    if opt:
        from alan_amd import OptParam
        Q = Plate(drift=Normal(OptParam(0.05), DRIFT),
                  R=Plate(init=Normal(OptParam(0.1), 1),
                          T=Plate(ts=Normal(OptParam(-0.1), OptParam(0.1, transformation=t.exp)), obs=Data())))
    else:
        Q = Plate(drift=Normal(0, DRIFT), R=Plate(init=Normal(0, 1), T=Plate(ts=Normal(0, 1), obs=Data())))
    sizes = {"R": R, "T": T}
    prob = Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes),
                   {"obs": y.reshape(R, T).float().refine_names("R", "T")})
    return prob, float(known)


def _same_sample_on_cpu(sample, cpu_prob, K):
    from alan_amd.dims import dims_of
    Kd = {g: alan.dims.Dim(str(d), K) for g, d in sample.groupvarname2Kdim.items()}
    by = {**{str(d): d for d in Kd.values()}, **{n: d for n, d in cpu_prob.all_platedims.items()}}

    def cpu_tree(tree):
        out = {}
        for k, v in tree.items():
            if isinstance(v, dict):
                out[k] = cpu_tree(v)
            else:
                ds = dims_of(v)
                out[k] = v.order(*ds).cpu()[tuple(by[str(d)] for d in ds)]
        return out

    return alan.Sample(problem=cpu_prob, sample=cpu_tree(sample.detached_sample), groupvarname2Kdim=Kd,
                       sampler=alan.PermutationSampler, reparam=False)


def test_nested_timeseries_elbo_host_logic(oracle_backend):
    prob, known = nested_problem(4)
    mean, stderr, vals = _band(prob, known, 200, 6, "cpu")
    assert mean - 6 * stderr - 0.05 < known
    assert known < mean + 6 * stderr + 1.0, (known, mean, stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [60, 300], ids=["chain_kernel", "matvec_scan"])
def test_nested_timeseries_elbo_ground_truth_gpu(K):
    prob, known = nested_problem(4)
    mean, stderr, vals = _band(prob, known, K, 12, "cuda")
    assert mean - 6 * stderr - 0.05 < known
    assert known < mean + 6 * stderr + (1.5 if K < 100 else 0.6), (known, mean, stderr)


@pytest.mark.gpu
def test_nested_timeseries_hip_equals_oracle_on_the_same_sample_and_has_gradients():
    """Same particles through the batched HIP chain and through the CPU oracle backend; then elbo_rws backward through
    the batched chain's backward kernel against autograd through the oracle."""
    from oracle import backend
    K = 30
    prob, _ = nested_problem(25, opt=True)
    prob.to("cuda")
    t.manual_seed(2)
    sample = prob.sample(K, reparam=False)
    gpu = float(sample.elbo_nograd(alan.no_checkpoint))
    sample.elbo_rws(alan.no_checkpoint).backward()
    cpu_prob, _ = nested_problem(25, opt=True)
    with backend.installed():
        cs = _same_sample_on_cpu(sample, cpu_prob, K)
        cpu = float(cs.elbo_nograd(alan.no_checkpoint))
        cs.elbo_rws(alan.no_checkpoint).backward()
    assert math.isfinite(gpu) and abs(gpu - cpu) <= 1e-4 * abs(cpu) + 1e-3, (gpu, cpu)
    grads = 0
    for (n1, p1), (n2, p2) in zip(prob.Q.named_parameters(), cpu_prob.Q.named_parameters()):
        assert n1 == n2 and p1.grad is not None and p2.grad is not None, n1
        scale = float(p2.grad.abs().max()) + 1e-6
        t.testing.assert_close(p1.grad.cpu(), p2.grad, rtol=2e-3, atol=2e-4 * scale, msg=lambda m: f"{n1}: {m}")
        grads += 1
    assert grads >= 4


@pytest.mark.gpu
def test_nested_timeseries_vi_gradients_fused_against_torch_distributions():
    """elbo_vi on the nested timeseries model with learnable Q: gradients through the fused producers (+ their HIP
    gradients), the batched chain and its tree backward against the same evaluation with the producers on
    torch.distributions.  Same seed, same reparameterised particles."""
    from alan_amd import dist as D

    def grads(fused):
        old = D.FUSE_NORMAL
        D.FUSE_NORMAL = fused
        try:
            prob, _ = nested_problem(12, opt=True)
            prob.to("cuda")
            t.manual_seed(4)
            t.cuda.manual_seed_all(4)
            sample = prob.sample(8, reparam=True)
            elbo = sample.elbo_vi(alan.no_checkpoint)
            elbo.backward()
            return float(elbo.detach()), {n: p.grad.detach().cpu().double().clone() for n, p in prob.Q.named_parameters()}
        finally:
            D.FUSE_NORMAL = old

    e1, g1 = grads(True)
    e0, g0 = grads(False)
    assert abs(e1 - e0) <= 2e-5 * abs(e0) + 1e-4
    assert len(g0) >= 4
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-6
        t.testing.assert_close(g1[n], g0[n], rtol=5e-3, atol=5e-4 * scale, msg=lambda m: f"{n}: {m}")


@pytest.mark.gpu
def test_timeseries_fp64_matches_the_oracle_to_rounding():
    """The Kalman model in fp64 (T = 300, K = 30): chain_tree_kernel<double>, the chain's terms added on load, the fp64
    producers -- against the CPU oracle backend on the same particles, to 1e-10 relative."""
    from oracle import backend
    prob, _ = kalman_problem(300)
    prob.to("cuda").double()
    t.manual_seed(6)
    sample = prob.sample(30, reparam=False)
    gpu = sample.elbo_nograd(alan.no_checkpoint)
    assert gpu.dtype == t.float64
    cpu_prob, _ = kalman_problem(300)
    cpu_prob.double()
    with backend.installed():
        cpu = float(_same_sample_on_cpu(sample, cpu_prob, 30).elbo_nograd(alan.no_checkpoint))
    assert abs(float(gpu) - cpu) <= 1e-10 * abs(cpu) + 1e-9, (float(gpu), cpu)
    assert float(sample.elbo_nograd(alan.no_checkpoint, graph=True)) == float(gpu)


# ---------------------------------------------------------------------------------------------------------------
# posterior sampling of timeseries K indices beyond the bare chain (sample_Ks_timeseries, reduce_Ks.py:85-232, handles
# both): a timeseries plate nested under another plate and a parent's K, and a timeseries plate that also holds an
# ordinary latent group.  Checked against the exact K-marginals (the path's backward), which the tests above pin to
# the Kalman smoother.
def with_group_problem(T, seed=0):
    """ts_t ~ N(A ts_{t-1}, NOISE);  z_t ~ N(ts_t, 0.5) (an ordinary latent in the same plate);  obs_t ~ N(z_t, OBS)"""
    g = t.Generator().manual_seed(seed)
    y = t.randn(T, generator=g)
    P = Plate(init=Normal(0, INIT_SCALE),
              T=Plate(ts=Timeseries("init", Normal(lambda prev: A * prev, NOISE)), z=Normal("ts", 0.5), obs=Normal("z", OBS)))
    Q = Plate(init=Normal(0, 1), T=Plate(ts=Normal(0, 1), z=Normal(0, 1), obs=Data()))
    sizes = {"T": T}
    return Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes), {"obs": y.refine_names("T")})


def _posterior_samples_match_marginals(prob, device, K, N, varnames):
    from alan_amd import mean
    from alan_amd.dims import dims_of
    prob.to(device)
    t.manual_seed(7)
    if device == "cuda":
        t.cuda.manual_seed_all(7)
    sample = prob.sample(K, reparam=False)
    marg = sample.marginals()
    isamp = sample.importance_sample(N)
    for v in varnames:
        m = marg._moments(v, mean)
        var = marg._moments(v, alan.var_from_raw_moment(mean))
        est = isamp._moments(v, mean)
        ds = dims_of(m)
        m, var, est = ((x.order(*ds) if ds else x).cpu().double() for x in (m, var, est))
        assert bool(((est - m).abs() < 6 * (var / N).sqrt() + 2e-3).all()), (v, (est - m).abs().max())
    return isamp


def test_posterior_of_a_nested_timeseries_host_logic(oracle_backend, smoothing):
    prob, _ = nested_problem(4)
    isamp = _posterior_samples_match_marginals(prob, "cpu", 40, 3000, ["ts", "init", "drift"])
    assert set(isamp.dump()["ts"].names) == {"N", "R", "T"}


def test_posterior_of_a_timeseries_plate_with_another_group_host_logic(oracle_backend, smoothing):
    _posterior_samples_match_marginals(with_group_problem(5), "cpu", 40, 3000, ["ts", "z", "init"])


@pytest.mark.gpu
def test_posterior_of_a_nested_timeseries_gpu(smoothing):
    prob, _ = nested_problem(6)
    isamp = _posterior_samples_match_marginals(prob, "cuda", 100, 20000, ["ts", "init", "drift"])
    assert set(isamp.dump()["ts"].names) == {"N", "R", "T"}


@pytest.mark.gpu
def test_posterior_of_a_timeseries_plate_with_another_group_gpu(smoothing):
    _posterior_samples_match_marginals(with_group_problem(8), "cuda", 100, 20000, ["ts", "z", "init"])


@pytest.mark.gpu
def test_timeseries_posterior_reference_mode_runs_end_to_end(monkeypatch):
    """posterior.TIMESERIES_POSTERIOR = "reference": per-timestep draws from the filtering marginals, as
    reduce_Ks.py:85-232 evaluates them (tests/test_gpu_posterior.py pins the tables)."""
    from alan_amd import posterior as PS
    monkeypatch.setattr(PS, "TIMESERIES_POSTERIOR", "reference")
    prob, _ = kalman_problem(6)
    prob.to("cuda")
    t.manual_seed(1)
    isamp = prob.sample(50, reparam=False).importance_sample(500)
    d = isamp.dump()["ts"]
    assert set(d.names) == {"N", "T"} and bool(t.isfinite(d.rename(None)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["nested", "with_group"])
def test_timeseries_posterior_reference_mode_with_chains_that_depend_on_the_sample(which, monkeypatch):
    """The same mode where a drawn K of another group is plugged into the chain (a chain per posterior sample): the
    forward recursion of sample n runs on chain n (round 2 raised NotImplementedError here)."""
    from alan_amd import posterior as PS
    monkeypatch.setattr(PS, "TIMESERIES_POSTERIOR", "reference")
    prob = nested_problem(5)[0] if which == "nested" else with_group_problem(5)
    prob.to("cuda")
    t.manual_seed(2)
    isamp = prob.sample(20, reparam=False).importance_sample(64)
    d = isamp.dump()["ts"]
    assert "N" in d.names and "T" in d.names and bool(t.isfinite(d.rename(None)).all())


@pytest.mark.gpu
def test_constant_multiple_lambda_is_folded_into_the_normal_producer(monkeypatch):
    """``lambda prev: 0.9 * prev`` (the Kalman model's transition mean) stays lazy on gradient-free evaluations: the
    Normal producer multiplies the location itself (factor scale field).  Same ELBO as with the lambda evaluated."""
    from alan_amd import dist as D, engine as E
    monkeypatch.setattr(D, "LAZY_TRANSITION", False)      # (else the transition factor is not produced at all: next test)
    # (a literal constant: a lambda that reads a module-level name -- kalman_problem's ``A * prev`` -- is evaluated as
    # written, the global could change between evaluations)
    assert D._scaled_form(lambda prev: A * prev) is None
    y = t.randn(50, generator=t.Generator().manual_seed(0))
    P = Plate(init=Normal(0, INIT_SCALE),
              T=Plate(ts=Timeseries("init", Normal(lambda prev: 0.9 * prev, NOISE)), obs=Normal("ts", OBS)))
    Q = Plate(init=Normal(0, 1), T=Plate(ts=Normal(0, 1), obs=Data()))
    prob = Problem(BoundPlate(P, {"T": 50}), BoundPlate(Q, {"T": 50}), {"obs": y.refine_names("T")})
    prob.to("cuda")
    t.manual_seed(3)
    sample = prob.sample(10, reparam=False)
    seen = []
    orig = E.normal_logprob

    def spy(*a, **k):
        seen.append(k.get("loc_scale", 1.0))
        return orig(*a, **k)

    monkeypatch.setattr(E, "normal_logprob", spy)
    lazy = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    assert any(abs(c - 0.9) < 1e-12 for c in seen), seen
    monkeypatch.setattr(D, "LAZY_SCALED", False)
    seen.clear()
    plain = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    assert all(c == 1.0 for c in seen)
    assert abs(lazy - plain) <= 2e-6 * abs(plain), (lazy, plain)
    monkeypatch.setattr(D, "LAZY_SCALED", True)
    assert abs(float(sample.elbo_nograd(alan.no_checkpoint, graph=True)) - lazy) <= 1e-6 * abs(lazy)
    assert D._scaled_form(lambda v: v * v) is None and D._scaled_form(lambda v: 2 * v + 1) is None


@pytest.mark.gpu
@pytest.mark.parametrize("K", [10, 20, 40])
def test_transition_factor_computed_on_load_by_the_chain_equals_the_materialised_one(K, monkeypatch):
    """The Normal transition factor of a timeseries stays unevaluated and the chain's first round computes it on load
    (alan_chain_logmmexp_terms_normal; K = 10: the LDS tree kernel, K = 20: one wave per product, K = 40: one workgroup
    per product): same ELBO as with the [T, K_init, K] factor written by the producer, and as the CPU oracle on the same
    particles."""
    from alan_amd import dist as D, native as N
    from oracle import backend
    prob, _ = kalman_problem(64)
    prob.to("cuda")
    t.manual_seed(4)
    sample = prob.sample(K, reparam=False)
    used = []
    orig = N.chain_logmmexp_terms

    def spy(terms, normal=None):
        used.append(normal is not None)
        return orig(terms, normal)

    monkeypatch.setattr(N, "chain_logmmexp_terms", spy)
    lazy = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    assert used == [True], used
    monkeypatch.setattr(D, "LAZY_TRANSITION", False)
    plain = float(sample.elbo_nograd(alan.no_checkpoint, graph=False))
    assert used[-1] is False
    assert abs(lazy - plain) <= 2e-6 * abs(plain), (lazy, plain)
    monkeypatch.setattr(D, "LAZY_TRANSITION", True)
    assert abs(float(sample.elbo_nograd(alan.no_checkpoint, graph=True)) - lazy) <= 1e-6 * abs(lazy)
    cpu_prob, _ = kalman_problem(64)
    cs = _same_sample_on_cpu(sample, cpu_prob, K)
    with backend.installed():
        cpu = float(cs.elbo_nograd(alan.no_checkpoint))
    assert abs(lazy - cpu) <= 1e-4 * abs(cpu) + 1e-3, (lazy, cpu)


def test_reference_mode_posterior_is_the_default_and_draws_the_filtering_marginals(oracle_backend):
    """posterior.TIMESERIES_POSTERIOR defaults to "reference" (VERDICT r3 item 7): importance_sample draws every timestep
    from the filtering marginal mixed over the sampled initial states -- on the host-logic route (one alan_reduce per
    timestep) the per-timestep frequencies match the table computed directly from the factor."""
    from alan_amd import posterior as PS
    assert PS.TIMESERIES_POSTERIOR == "reference"
    g = t.Generator().manual_seed(2)
    C_, T, K, N = 2, 5, 6, 40000
    flat = t.randn(C_, T, K, K, generator=g)
    init = t.randint(0, K, (N, C_), generator=g)
    logp = PS._filtering_by_steps(flat, init, N, C_, False)
    # brute force: alpha_t[n] by explicit recursion in fp64, mixed and normalised
    want = t.empty(C_, T, K, dtype=t.float64)
    for b in range(C_):
        alpha = flat[b, 0].double()[init[:, b]]                              # [N, K]
        for step in range(T):
            if step:
                alpha = t.logsumexp(alpha[:, :, None] + flat[b, step].double()[None], 1)
            mix = t.logsumexp(alpha, 0)
            want[b, step] = mix - t.logsumexp(mix, 0)
    t.testing.assert_close(logp.double(), want, rtol=1e-5, atol=1e-5)
    assert abs(float(logp.exp().sum(-1).mean()) - 1.0) < 1e-5
