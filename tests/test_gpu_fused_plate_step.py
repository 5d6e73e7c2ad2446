"""alan_normal_lse -- the plate step with the factor producer fused in: against the two-launch
route it replaces (ALAN_MODE_NORMAL producer + the rows kernel) and against the CPU oracle, then end to end."""
import math

import pytest
import torch as t

import alan_amd as alan
from alan_amd import engine as E, dist as D
from alan_amd.dims import Dim, PT
from conftest import load_golden
from oracle import alan_oracle as orc
import models

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("M,NK,NL,NS,Ev,n_small,log_scale", [
    (300, 30, 30, 30, 18, 2, False), (38, 100, 100, 100, 18, 2, False), (7, 5, 4, 3, 3, 0, False),
    (11, 33, 9, 70, 20, 1, True), (5, 8, 40, 31, 1, 3, True), (3, 64, 2, 32, 32, 4, False), (4, 16, 5, 128, 9, 1, False),
    (6, 10, 7, 50, 18, 2, True), (3, 9, 5, 200, 2, 1, False), (9, 97, 3, 33, 17, 2, False), (2, 130, 2, 5, 31, 0, True), (5, 12, 6, 40, 32, 2, False)])
def test_fused_plate_step_matches_the_two_launch_route_and_the_oracle(M, NK, NL, NS, Ev, n_small, log_scale):
    g = t.Generator().manual_seed(M + NK + NS)
    pl, K, dl, ds = Dim("plate", M), Dim("K", NK), Dim("Kl", NL), Dim("Ks", NS)
    z = t.randn(M, NK, Ev, generator=g)
    mu = t.randn(NL, Ev, generator=g)
    raw = 0.3 * t.randn(NS, Ev, generator=g)
    sc = raw if log_scale else raw.exp()
    small_dims = [(pl, K), (K,), (pl,), (K, pl)]
    smalls = []
    for i in range(n_small):
        dims = small_dims[i]
        x = t.randn(*[d.size for d in dims], generator=g)
        if i == 0:
            x[0, 0] = float("-inf")
        smalls.append((x, dims))
    res = E.normal_lse((z.to(DEV), (pl, K)), (mu.to(DEV), (dl,)), (sc.to(DEV), (ds,)),
                       [(x.to(DEV), d) for x, d in smalls], pl, K, log_scale=log_scale)
    assert res is not None
    out, odims = res
    assert odims[0] is dl and odims[1] is ds and out.shape == (NL, NS)
    # (a) the route it replaces: producer launch, then log-sum-exp over K + plate sum
    F = E.normal_logprob((z.to(DEV), (pl, K)), (mu.to(DEV), (dl,)), (sc.to(DEV), (ds,)), (pl, dl, ds, K),
                         log_scale=log_scale)
    two, tdims = E.reduce_factors([(F, (pl, dl, ds, K)), *[(x.to(DEV), d) for x, d in smalls]], reduce=(K,), plate=(pl,))
    two = two if tdims[0] is dl else two.t()
    t.testing.assert_close(out, two, rtol=2e-5, atol=2e-4 * max(1.0, M / 30))
    # (b) the CPU oracle on torch.distributions' log-prob
    sigma = raw.exp()
    lp = t.distributions.Normal(mu[None, :, None, None, :], sigma[None, None, :, None, :]).log_prob(
        z[:, None, None, :, :]).sum(-1)                                     # [M, NL, NS, NK]
    facs = [(lp, ("m", "l", "s", "k"))] + [(x, tuple({id(pl): "m", id(K): "k"}[id(d)] for d in dims)) for x, dims in smalls]
    ref = orc.plate_sum(orc.logsumexp_sum(("k",), *facs), "m")
    want = orc.align(ref, ("l", "s"))
    t.testing.assert_close(out.cpu(), want, rtol=3e-5, atol=3e-4 * max(1.0, M / 30))


def test_fused_plate_step_declines_other_shapes():
    pl, K, dl, ds, other = Dim("plate", 4), Dim("K", 5), Dim("Kl", 3), Dim("Ks", 3), Dim("other", 2)
    z, mu, sc = t.randn(4, 5, 2).to(DEV), t.randn(3, 2).to(DEV), t.rand(3, 2).to(DEV) + 0.5
    assert E.normal_lse((z, (pl, K)), (mu, (dl,)), (sc, (ds,)), [(t.randn(2, 5).to(DEV), (other, K))], pl, K) is None
    assert E.normal_lse((z.double(), (pl, K)), (mu, (dl,)), (sc, (ds,)), [], pl, K) is None
    assert E.normal_lse((t.randn(4, 5, 40).to(DEV), (pl, K)), (t.randn(3, 40).to(DEV), (dl,)),
                        (t.rand(3, 40).to(DEV) + 0.5, (ds,)), [], pl, K) is None       # event length > 32


@pytest.mark.parametrize("fixture", ["e2e_movielens_K3.pt", "e2e_movielens_K10.pt"])
def test_movielens_elbo_with_the_fused_plate_step(fixture, monkeypatch):
    """End to end on the reference's own sample tree: the fused route gives the reference's ELBO (1e-4, north_star)
    and the default route's to 1e-6, eagerly, as a replayed graph and under Split."""
    fx = load_golden(fixture)
    if fx["data"]["obs"][0].dtype != t.float32:
        fx = dict(fx, data={k: (v[0].float(), v[1]) for k, v in fx["data"].items()})     # the fused kernel is fp32
    prob = models.BUILDERS["movielens"](fx).to(DEV)
    sample = models.sample_from_fixture(prob, fx, DEV)
    base = {n: float(sample.elbo_nograd(s)) for n, s in (("no", alan.no_checkpoint), ("split", alan.Split("plate_1", 38)))}
    calls = []
    real = E.normal_lse
    monkeypatch.setattr(E, "normal_lse", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(D, "FUSE_PLATE_STEP", True)
    fused = float(sample.elbo_nograd(alan.no_checkpoint))
    assert len(calls) == 1
    fused_split = float(sample.elbo_nograd(alan.Split("plate_1", 38)))
    assert len(calls) == 1 + 8
    graph = float(sample.elbo_nograd(alan.no_checkpoint, graph=True))
    graph2 = float(sample.elbo_nograd(alan.no_checkpoint, graph=True))
    ref = float(fx["elbo"]["no_checkpoint"])
    assert abs(fused - base["no"]) <= 1e-6 * abs(base["no"])
    assert abs(fused_split - base["split"]) <= 1e-6 * abs(base["split"])
    assert graph == graph2 and abs(graph - fused) <= 1e-6 * abs(fused)
    assert abs(fused - ref) <= 1e-4 * abs(ref) + 1e-5
    # with gradients to record the factor is materialised as before
    monkeypatch.setattr(E, "normal_lse", lambda *a, **k: pytest.fail("the fused kernel has no backward"))
    sample.elbo_rws(alan.no_checkpoint).backward()


def test_fused_plate_step_fuzz_against_the_two_launch_route():
    """Random shapes (every tile-edge case of the MFMA kernel: K and scale rows around multiples of 32, odd and even
    event lengths up to 32, one plate element, one scale row, 0-4 small factors with broadcast strides, log-scale
    or scale) against the producer + log-sum-exp + plate-sum route."""
    import random
    rnd = random.Random(7)
    g = t.Generator().manual_seed(7)
    sizes = [1, 2, 3, 5, 16, 31, 32, 33, 47, 63, 64, 65, 96, 100]
    for it in range(40):
        M, NK, NL, NS = rnd.choice([1, 2, 5, 9, 17]), rnd.choice(sizes), rnd.choice([1, 2, 7, 30]), rnd.choice(sizes)
        Ev = rnd.choice([1, 2, 3, 4, 7, 8, 17, 18, 31, 32])
        n_small, log_scale = rnd.randrange(5), rnd.random() < 0.5
        pl, K, dl, ds = Dim("plate", M), Dim("K", NK), Dim("Kl", NL), Dim("Ks", NS)
        z = t.randn(M, NK, Ev, generator=g).to(DEV)
        mu = t.randn(NL, Ev, generator=g).to(DEV)
        raw = (0.3 * t.randn(NS, Ev, generator=g)).to(DEV)
        sc = raw if log_scale else raw.exp()
        small_dims = [(pl, K), (K,), (pl,), (K, pl)]
        smalls = [(t.randn(*[d.size for d in small_dims[i]], generator=g).to(DEV), small_dims[i]) for i in range(n_small)]
        res = E.normal_lse((z, (pl, K)), (mu, (dl,)), (sc, (ds,)), smalls, pl, K, log_scale=log_scale)
        assert res is not None, (M, NK, NL, NS, Ev, n_small)
        out, odims = res
        F = E.normal_logprob((z, (pl, K)), (mu, (dl,)), (sc, (ds,)), (pl, dl, ds, K), log_scale=log_scale)
        two, tdims = E.reduce_factors([(F, (pl, dl, ds, K)), *smalls], reduce=(K,), plate=(pl,))
        two = two if tdims[0] is dl else two.t()
        t.testing.assert_close(out, two, rtol=3e-5, atol=3e-4, msg=lambda m: f"{(M, NK, NL, NS, Ev, n_small, log_scale)}: {m}")


def test_a_captured_graph_is_not_reused_under_other_routing_switches(monkeypatch):
    fx = load_golden("e2e_movielens_K3.pt")
    if fx["data"]["obs"][0].dtype != t.float32:
        fx = dict(fx, data={k: (v[0].float(), v[1]) for k, v in fx["data"].items()})
    prob = models.BUILDERS["movielens"](fx).to(DEV)
    sample = models.sample_from_fixture(prob, fx, DEV)
    a = float(sample.elbo_nograd(alan.no_checkpoint, graph=True))
    calls = []
    real = E.normal_lse
    monkeypatch.setattr(E, "normal_lse", lambda *x, **k: (calls.append(1), real(*x, **k))[1])
    monkeypatch.setattr(D, "FUSE_PLATE_STEP", True)
    b = float(sample.elbo_nograd(alan.no_checkpoint, graph=True))         # a NEW capture, through the fused kernel
    assert len(calls) >= 1 and abs(a - b) <= 1e-6 * abs(a)
    assert len(sample._graphs) == 2
