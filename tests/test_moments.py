"""Posterior moments / marginals = the hot path's backward in production use.  Follows the reference's
tests/test_problem_vs_itself.py: `test_moments_sample_marginal` (:71-88, two autograd routes agree to
rtol 1e-4) and `test_moments_ground_truth` (:121-158, closed-form posterior of tests/linear_gaussian.py
and tests/linear_gaussian_latents.py within 7 standard errors, ESS-based)."""
import math

import pytest
import torch as t

import alan_amd as alan
from alan_amd import mean, mean2, var
from alan_amd.dims import dims_of
from conftest import load_golden
import models


def _posterior_a(model, data):
    """Closed-form posterior mean / precision of `a` (tests/linear_gaussian.py:8-24, _latents.py:8-24)."""
    n = data.shape[0]
    prior_prec = 1 / 2 ** 2
    if model == "linear_gaussian":
        like_prec, mult = 1 / 3 ** 2, 2.5
        prec = prior_prec + n * like_prec * mult ** 2
        mu = (prior_prec * 2 + like_prec * mult ** 2 * (data.sum() / mult)) / prec
    else:
        like_prec = 1 / (1.3 ** 2 + 1.5 ** 2)
        prec = prior_prec + n * like_prec
        mu = (prior_prec * 2 + like_prec * data.sum()) / prec
    return float(mu), float(prec)


CASES = [("linear_gaussian", "e2e_linear_gaussian.pt", 3000), ("linear_gaussian_latents", "e2e_linear_gaussian_latents.pt", 300)]


def _run(model, fixture, K, device, sampler):
    fx = load_golden(fixture)
    prob = models.BUILDERS[model](fx).to(device)
    t.manual_seed(7)
    sample = prob.sample(K, reparam=False, sampler=sampler)
    marg = sample.marginals()
    # (1) the two routes through the backward agree
    for vn, m in [("a", mean), ("a", mean2)] + ([("z", mean)] if model.endswith("latents") else []):
        a = sample._moments(vn, m)
        b = marg._moments(vn, m)
        ds = dims_of(a)
        a, b = (a.order(*ds), b.order(*ds)) if ds else (a, b)
        t.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-4, atol=1e-5)
    # (2) weights are distributions over K
    for key, w in marg.weights.items():
        Kd = [d for d in dims_of(w) if str(d).startswith("K_")]
        tot = w.sum(Kd)
        tot = tot.order(*dims_of(tot)) if dims_of(tot) else tot
        t.testing.assert_close(tot.cpu(), t.ones_like(tot.cpu()), rtol=1e-4, atol=1e-4)
    # (3) analytic ground truth within 7 standard errors (ESS = min over all latents)
    mu, prec = _posterior_a(model, fx["data"]["d"][0])
    ess = marg.min_ess()
    for m, truth in ((mean, mu), (mean2, mu ** 2 + 1 / prec)):
        est = float(marg._moments("a", m))
        v = float(marg._moments("a", alan.var_from_raw_moment(m)))
        stderr = math.sqrt(v / ess)
        assert abs(est - truth) < 7 * stderr + 1e-3, (m, est, truth, stderr, ess)
    named = sample.moments("a", mean)
    assert isinstance(named, t.Tensor) and named.ndim == 0
    assert float(marg.moments("a", var)) > 0


@pytest.mark.parametrize("model,fixture,K", CASES)
@pytest.mark.parametrize("sampler", alan.samplers, ids=lambda s: s.__name__)
def test_moments_host_logic(model, fixture, K, sampler, oracle_backend):
    _run(model, fixture, min(K, 300), "cpu", sampler)


@pytest.mark.gpu
@pytest.mark.parametrize("model,fixture,K", CASES)
@pytest.mark.parametrize("sampler", alan.samplers, ids=lambda s: s.__name__)
def test_moments_gpu(model, fixture, K, sampler):
    _run(model, fixture, K, "cuda", sampler)


def test_joint_marginals_and_errors(oracle_backend):
    fx = load_golden("e2e_model1.pt")
    prob = models.BUILDERS["model1"](fx)
    sample = models.sample_from_fixture(prob, fx, "cpu")
    marg = sample.marginals(joints=[("ab", "c")])
    w = marg.weights[frozenset(["ab", "c"])]
    assert {str(d) for d in dims_of(w)} == {"K_ab", "K_c"}
    assert abs(float(w.sum(dims_of(w))) - 1.0) < 1e-4
    # the joint marginalises to the univariates
    wa = marg.weights[frozenset(["ab"])]
    Kc = [d for d in dims_of(w) if str(d) == "K_c"]
    t.testing.assert_close(w.sum(Kc).order(*dims_of(wa)), wa.order(*dims_of(wa)), rtol=1e-4, atol=1e-5)
    with pytest.raises(Exception):
        sample.marginals(joints=["ab"])
    with pytest.raises(Exception):
        sample.marginals(joints=[("a", "c")])          # variable name, not group name
    with pytest.raises(Exception):
        sample.moments("a", var)                       # compound moments need Marginals


# ---------------------------------------------------------------------------------------------
# importance_sample: N joint draws over the K particles.  `marginals.moments` is the exact moment under
# that distribution, so the two must agree within 6 standard errors of the N-sample mean
# (tests/test_problem_vs_itself.py:90-119 `test_moments_importance_sample`).
def _importance(model, fixture, K, N, device, builder_fx=True):
    fx = load_golden(fixture)
    prob = models.BUILDERS[model](fx).to(device)
    t.manual_seed(11)
    sample = prob.sample(K, reparam=False) if model != "model1" else models.sample_from_fixture(prob, fx, device)
    marg = sample.marginals()
    isamp = sample.importance_sample(N)
    names = {"linear_gaussian": ["a"], "linear_gaussian_latents": ["a", "z"], "model1": ["a", "b", "c", "d"]}[model]
    for vn in names:
        for m in (mean, mean2):
            exact = marg._moments(vn, m)
            est = isamp._moments(vn, m)
            v = marg._moments(vn, alan.var_from_raw_moment(m))
            ds = dims_of(exact)
            if ds:
                exact, est, v = exact.order(*ds), est.order(*ds), v.order(*ds)
            bound = 6 * (v / N).sqrt() + 1e-4
            assert bool(((est - exact).abs() < bound).all()), (vn, est, exact, bound)
    dumped = isamp.dump()
    assert all("N" in x.names for x in dumped.values())


@pytest.mark.parametrize("model,fixture,K", [("linear_gaussian", "e2e_linear_gaussian.pt", 30),
                                             ("linear_gaussian_latents", "e2e_linear_gaussian_latents.pt", 10),
                                             ("model1", "e2e_model1.pt", 3)])
def test_importance_sample_host_logic(model, fixture, K, oracle_backend):
    _importance(model, fixture, K, 4000, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("model,fixture,K", [("linear_gaussian", "e2e_linear_gaussian.pt", 100),
                                             ("linear_gaussian_latents", "e2e_linear_gaussian_latents.pt", 30),
                                             ("model1", "e2e_model1.pt", 3)])
def test_importance_sample_gpu(model, fixture, K):
    _importance(model, fixture, K, 20000, "cuda")
