"""The Bernoulli producer that computes its own logits (alan_reduce mode BERNOULLI_LINEAR; engine.
bernoulli_linear_logprob; dims.LinearPT): against torch (the lambda evaluated, then td.Bernoulli.log_prob, as
TorchDimDist.py:127-162 does through torchdim) on the movielens and bus_breakdown shapes and on odd layouts; in a
multi-problem launch; and end to end, switched on and off."""
import ctypes
import math

import pytest
import torch as t

import alan_amd as alan
import models
from conftest import load_golden
from alan_amd import engine as E
from alan_amd import native as N

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(value, vdims, terms, out_dims):
    """fp64 torch: named broadcast by einsum-free alignment."""
    names = []
    for _, d in [(value, vdims)] + [p for term in terms for p in term]:
        for n in d:
            if n not in names:
                names.append(n)

    def al(x, d, extra=0):
        x = x.double()
        perm = [d.index(n) for n in names if n in d]
        x = x.permute([*perm, *range(len(d), x.ndim)])
        idx = tuple(slice(None) if n in d else None for n in names)
        return x[idx] if idx else x

    logits = 0
    for term in terms:
        if len(term) == 1:
            logits = logits + al(*term[0])
        else:
            (a, ad), (b, bd) = term
            logits = logits + (al(a, ad) * al(b, bd)).sum(-1)
    v, logits = t.broadcast_tensors(al(value, vdims), logits)
    lp = t.distributions.Bernoulli(logits=logits).log_prob(v)
    drop = [i for i, n in enumerate(names) if n not in out_dims]
    lp = lp.sum(drop) if drop else lp
    left = [n for n in names if n in out_dims]
    return lp.permute([left.index(n) for n in out_dims])


@pytest.mark.parametrize("M,K,Nf,Ev", [(300, 30, 5, 18), (300, 100, 5, 18), (7, 3, 5, 18), (33, 10, 1, 1), (5, 4, 300, 3),
                                       (2, 2, 1000, 7)])
def test_movielens_logits_dot_matches_torch(M, K, Nf, Ev):
    g = t.Generator().manual_seed(M + K)
    z = t.randn(K, M, Ev, generator=g).to(DEV)                     # a sample: [K_z, plate_1, d_z]
    x = t.randn(M, Nf, Ev, generator=g).to(DEV)
    obs = (t.rand(M, Nf, generator=g) < 0.4).float().to(DEV)
    val, terms = (obs, ("m", "n")), [((z, ("k", "m")), (x, ("m", "n")))]
    for out_dims, aff in ((("m", "k"), (1.0, 0.0)), (("k", "m"), (-1.0, 0.25)), (("m", "k", "n"), (1.0, 0.0))):
        got = E.bernoulli_linear_logprob(val, terms, out_dims, aff)
        assert got is not None and got.shape == tuple({"m": M, "k": K, "n": Nf}[d] for d in out_dims)
        want = aff[0] * _ref(obs.cpu(), ("m", "n"), [((z.cpu(), ("k", "m")), (x.cpu(), ("m", "n")))], out_dims) + aff[1]
        t.testing.assert_close(got.cpu().double(), want, rtol=3e-6, atol=3e-5 * max(1, Nf * Ev) ** 0.5)


def test_bus_breakdown_logits_sum_of_terms_and_strided_operands():
    g = t.Generator().manual_seed(11)
    Ka, Kg, Y, B, I, nb, nr = 6, 5, 2, 3, 150, 11, 4
    alpha = t.randn(Ka, Y, B, generator=g).to(DEV)
    phi = t.randn(Kg, nb, generator=g).to(DEV)
    psi = t.randn(nr, Kg, generator=g).to(DEV).t()                 # a transposed view: dot stride Kg
    bus = t.randn(Y, B, I, nb, generator=g).to(DEV)
    run = t.randn(Y, B, I, 2 * nr, generator=g).to(DEV)[..., ::2]   # strided along the contracted dim
    obs = (t.rand(I, B, Y, generator=g) < 0.5).float().to(DEV).permute(2, 1, 0)   # permuted storage
    val = (obs, ("Y", "B", "I"))
    terms = [((alpha, ("Ka", "Y", "B")),), ((phi, ("Kg",)), (bus, ("Y", "B", "I"))), ((psi, ("Kg",)), (run, ("Y", "B", "I")))]
    out_dims = ("Ka", "Y", "B", "Kg")
    got = E.bernoulli_linear_logprob(val, terms, out_dims)
    cpu = lambda term: tuple((x.cpu(), d) for x, d in term)
    want = _ref(obs.cpu(), val[1], [cpu(tm) for tm in terms], out_dims)
    t.testing.assert_close(got.cpu().double(), want, rtol=3e-6, atol=2e-4)
    # nothing summed: one value per element
    got2 = E.bernoulli_linear_logprob(val, terms, ("Ka", "Y", "B", "Kg", "I"))
    assert got2 is None or got2.shape == (Ka, Y, B, Kg, I)          # (5 keep dims: the library may decline)
    keep4 = E.bernoulli_linear_logprob((obs[0], ("B", "I")), [((phi, ("Kg",)), (bus[0], ("B", "I")))], ("B", "Kg", "I"))
    want4 = _ref(obs[0].cpu(), ("B", "I"), [((phi.cpu(), ("Kg",)), (bus[0].cpu(), ("B", "I")))], ("B", "Kg", "I"))
    t.testing.assert_close(keep4.cpu().double(), want4, rtol=3e-6, atol=2e-5)


@pytest.mark.parametrize("Ka,Kg,Y,B,I", [(100, 100, 2, 3, 150), (37, 130, 1, 2, 255), (64, 48, 3, 1, 9)],
                         ids=["bus_K100", "ragged", "short_sum"])
def test_plain_terms_meeting_only_in_the_summed_dim_take_the_tile_kernel(Ka, Kg, Y, B, I):
    """logits = alpha[Ka, Y, B] + d1[Y, B, I, Kg] + d2[Kg, Y, B, I] (bus_breakdown once its dot products, which lack the
    K_alpha dim, are evaluated): >= 2^20 elements go to pair.hip's tiles (the Bernoulli variant), smaller ones to the
    lane-group kernel -- both against torch in fp64; permuted storage of every operand."""
    g = t.Generator().manual_seed(13)
    alpha = t.randn(Ka, Y, B, generator=g).to(DEV)
    d1 = (2 * t.randn(Y, B, I, Kg, generator=g)).to(DEV)
    d2 = t.randn(Kg, Y, B, I, generator=g).to(DEV)
    obs = (t.rand(I, B, Y, generator=g) < 0.5).float().to(DEV).permute(2, 1, 0)
    val = (obs, ("Y", "B", "I"))
    terms = [((alpha, ("Ka", "Y", "B")),), ((d1, ("Y", "B", "I", "Kg")),), ((d2, ("Kg", "Y", "B", "I")),)]
    for out_dims, aff in ((("Y", "B", "Ka", "Kg"), (1.0, 0.0)), (("Kg", "Y", "B", "Ka"), (-1.0, 0.25))):
        got = E.bernoulli_linear_logprob(val, terms, out_dims, aff)
        cpu = lambda term: tuple((x.cpu(), d) for x, d in term)
        want = aff[0] * _ref(obs.cpu(), val[1], [cpu(tm) for tm in terms], out_dims) + aff[1]
        t.testing.assert_close(got.cpu().double(), want, rtol=3e-6, atol=3e-5 * I ** 0.5)


def test_unsupported_shapes_are_declined_not_miscomputed():
    g = t.Generator().manual_seed(2)
    z, x = t.randn(4, 6, 3, generator=g).to(DEV), t.randn(6, 5, 3, generator=g).to(DEV)
    obs = (t.rand(6, 5, generator=g) < 0.5).to(DEV)
    term = [((z, ("k", "m")), (x, ("m", "n")))]
    assert E.bernoulli_linear_logprob((obs.double(), ("m", "n")), term, ("m", "k")) is None          # fp64 value
    assert E.bernoulli_linear_logprob((obs.float(), ("m", "n")), [((z.double(), ("k", "m")), (x.double(), ("m", "n")))],
                                      ("m", "k")) is None
    four = [((z, ("k", "m")), (x, ("m", "n")))] * 4                                                    # > 3 terms / > 6 factors
    assert E.bernoulli_linear_logprob((obs.float(), ("m", "n")), four, ("m", "k")) is None
    # the C entry point: malformed descriptors
    d = N.ReduceDesc()
    d.mode, d.ndim, d.n_factors = N.MODE_BERNOULLI_LINEAR, 1, 1
    d.size[0], d.role[0] = 6, N.KEEP
    assert N.lib().alan_reduce_check(ctypes.byref(d)) == -1                                            # value only
    d.n_factors = 2
    N.fill_tensor(d.factor[0], obs.float(), [5])
    N.fill_tensor(d.factor[1], z, [3], 2.0)                                                            # terms must count from 1
    N.fill_tensor(d.out, t.empty(6, device=DEV), [1])
    assert N.lib().alan_reduce_check(ctypes.byref(d)) == -1
    d.role[0] = N.DOT                                                                                   # value on a DOT dim
    d.factor[1].scale = 1.0
    assert N.lib().alan_reduce_check(ctypes.byref(d)) == -1


def test_in_a_multi_problem_launch_the_values_are_those_of_a_lone_launch():
    from alan_amd.dims import Dim
    g = t.Generator().manual_seed(4)
    M, K, Nf, Ev = 300, 30, 5, 18
    z = t.randn(K, M, Ev, generator=g).to(DEV)
    x = t.randn(M, Nf, Ev, generator=g).to(DEV)
    obs = (t.rand(M, Nf, generator=g) < 0.4).float().to(DEV)
    mu, sc = t.randn(M, Ev, generator=g).to(DEV), t.rand(M, Ev, generator=g).to(DEV) + 0.5
    calls = [
        lambda: E.normal_logprob((z, ("k", "m")), (mu, ("m",)), (sc, ("m",)), ("m", "k"), affine=(-1.0, -math.log(K))),
        lambda: E.bernoulli_linear_logprob((obs, ("m", "n")), [((z, ("k", "m")), (x, ("m", "n")))], ("m", "k")),
        lambda: E.normal_logprob((z[:, 0], ("k",)), (mu[0], ()), (sc[0], ()), ("k",)),
        lambda: E.bernoulli_linear_logprob((obs, ("m", "n")), [((z, ("k", "m")), (x, ("m", "n")))], ("k", "m")),   # a second one
    ]
    want = [c() for c in calls]
    t.cuda.synchronize()
    with t.no_grad(), N.deferring():
        with N.may_defer():
            got = [c() for c in calls]
            assert N.n_pending() == len(calls)
        N.flush()
    # (the same values -- not the same bits since round 4: a multi-problem launch takes lanes from its biggest problems until
    # it is about one chipful, plan.h fill_small_multi, and a sum over fewer lanes is added up in another order)
    for a, b in zip(got, want):
        t.testing.assert_close(a, b, rtol=2e-6, atol=2e-6 * float(b.abs().max()))


@pytest.mark.parametrize("fixture,model", [("e2e_movielens_K10.pt", "movielens"), ("e2e_bus_breakdown_K3.pt", "bus_breakdown"),
                                           ("e2e_movielens_K30.pt", "movielens"), ("e2e_bus_breakdown_K30.pt", "bus_breakdown")])
def test_end_to_end_the_elbo_is_the_same_with_the_logits_computed_in_the_producer(fixture, model, monkeypatch):
    """fp32 observations (the fixtures' are fp64, which keeps the evaluated-lambda route): the ELBO with the lambda left
    to the producer equals the ELBO with the lambda evaluated by torch, and both the reference's."""
    from alan_amd import dist as D
    fx = load_golden(fixture)
    prob = models.BUILDERS[model](fx).to(DEV).float()
    sample = models.sample_from_fixture(prob, fx, DEV)
    ref = float(fx["elbo"]["no_checkpoint"])
    taken = []
    orig = E.bernoulli_linear_logprob

    def spy(*a, **k):
        out = orig(*a, **k)
        taken.append(out is not None)
        return out

    monkeypatch.setattr(E, "bernoulli_linear_logprob", spy)
    strategies = [alan.no_checkpoint, alan.checkpoint]
    if model == "movielens":
        strategies.append(alan.Split("plate_1", 38))
    for strat in strategies:
        on = float(sample.elbo_nograd(strat, graph=False))
        assert taken and all(taken), taken
        monkeypatch.setattr(D, "LINEAR_LOGITS", False)
        n = len(taken)
        off = float(sample.elbo_nograd(strat, graph=False))
        assert len(taken) == n                      # the lambda was evaluated by torch
        monkeypatch.setattr(D, "LINEAR_LOGITS", True)
        assert abs(on - off) <= 2e-6 * abs(off), (on, off)
        assert abs(on - ref) <= 1e-4 * abs(ref), (on, ref)
    # replayed graphs and gradient-carrying evaluations agree as well (RWS: the likelihood's operands carry no gradient)
    g1 = float(sample.elbo_nograd(alan.no_checkpoint, graph=True))
    assert abs(g1 - float(sample.elbo_nograd(alan.no_checkpoint, graph=False))) <= 1e-6 * abs(g1)
    rws = float(sample.elbo_rws(alan.no_checkpoint))
    assert abs(rws - g1) <= 2e-6 * abs(g1)


@pytest.mark.parametrize("dtype", [t.float32, t.float64])
def test_dot_sum_matches_einsum(dtype):
    """alan_reduce mode DOT: sum over the event dim of a * b, broadcast over first-class dims (a lambda's `phi @ x`)."""
    g = t.Generator().manual_seed(9)
    phi = t.randn(7, 11, generator=g, dtype=dtype).to(DEV)
    bus = t.randn(2, 3, 150, 11, generator=g, dtype=dtype).to(DEV)
    out = E.dot_sum((phi, ("Kg",)), (bus, ("Y", "B", "I")), ("Kg", "Y", "B", "I"))
    want = t.einsum("kz,ybiz->kybi", phi.double(), bus.double())
    t.testing.assert_close(out.double(), want, rtol=3e-6 if dtype == t.float32 else 1e-12, atol=1e-5 if dtype == t.float32 else 1e-12)
    # strided operands, a shared first-class dim, another output order, a first-class dim summed as well
    a = t.randn(5, 4, 6, generator=g, dtype=dtype).to(DEV).transpose(0, 1)          # dims (m, k), event 6
    b = t.randn(4, 9, 12, generator=g, dtype=dtype).to(DEV)[..., ::2]               # dims (m, n), event 6 strided
    out2 = E.dot_sum((a, ("m", "k")), (b, ("m", "n")), ("n", "k", "m"))
    t.testing.assert_close(out2.double(), t.einsum("mkz,mnz->nkm", a.double(), b.double()), rtol=3e-6, atol=1e-5)
    out3 = E.dot_sum((a, ("m", "k")), (b, ("m", "n")), ("k",))
    t.testing.assert_close(out3.double(), t.einsum("mkz,mnz->k", a.double(), b.double()), rtol=3e-6, atol=1e-4)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_linear_logits_and_dot_against_torch(seed):
    """Random index spaces, term structures, storage orders and strides: alan_reduce modes BERNOULLI_LINEAR and DOT
    against fp64 torch on the materialised broadcast."""
    import random
    rnd = random.Random(seed)
    g = t.Generator().manual_seed(100 + seed)
    names = ["a", "b", "c", "d", "e"][: rnd.randint(2, 5)]
    size = {n: rnd.choice([1, 2, 3, 5, 7, 33]) for n in names}

    def tensor(dims, extra=()):
        """A random tensor over ``dims`` (+ trailing ``extra`` positional sizes), stored in a random dim order, sometimes
        as a strided slice of a larger one."""
        order = list(range(len(dims)))
        rnd.shuffle(order)
        shape = [size[dims[i]] for i in order] + list(extra)
        if rnd.random() < 0.3 and shape:
            big = t.randn([2 * s for s in shape], generator=g).to(DEV)
            x = big[tuple(slice(None, None, 2) for _ in shape)]
        else:
            x = t.randn(shape, generator=g).to(DEV)
        inv = [order.index(i) for i in range(len(dims))]
        return x.permute([*inv, *range(len(dims), x.ndim)])

    def some(k_min=0):
        k = rnd.randint(k_min, len(names))
        return tuple(rnd.sample(names, k))

    vdims = some(1)
    value = ((t.rand([size[n] for n in vdims], generator=g) < 0.5).float().to(DEV), vdims)
    terms, used = [], set(vdims)
    for _ in range(rnd.randint(1, 3)):
        if rnd.random() < 0.6 and len(terms) * 2 + 3 <= 5:
            L = rnd.choice([1, 2, 4, 9, 18])
            da, db = some(), some()
            terms.append(((tensor(da, (L,)), da), (tensor(db, (L,)), db)))
            used |= set(da) | set(db)
        else:
            dp = some()
            terms.append(((tensor(dp), dp),))
            used |= set(dp)
    if sum(len(tm) for tm in terms) > 5 or not any(len(tm) == 2 for tm in terms):
        L = 3
        terms = [((tensor(vdims, (L,)), vdims), (tensor(vdims[:1], (L,)), vdims[:1]))]
        used = set(vdims)
    every = [n for n in names if n in used]
    keep = tuple(n for n in every if rnd.random() < 0.6)[:4]
    out_dims = tuple(rnd.sample(keep, len(keep)))
    n_sum = len([n for n in every if n not in keep and size[n] > 1])
    got = E.bernoulli_linear_logprob(value, terms, out_dims, (1.0, 0.0))
    if got is None:
        assert n_sum > 2 or len([n for n in keep if size[n] > 1]) > 4 or len(every) + sum(len(tm) == 2 for tm in terms) > N.MAX_DIMS
        return
    cpu = lambda tm: tuple((x.cpu(), d) for x, d in tm)
    want = _ref(value[0].cpu(), vdims, [cpu(tm) for tm in terms], out_dims)
    scale = float(want.abs().max()) + 1.0
    t.testing.assert_close(got.cpu().double(), want.reshape(got.shape), rtol=1e-5, atol=3e-6 * scale)
    # the first dot term alone, through mode DOT
    (a, da), (b, db) = next(tm for tm in terms if len(tm) == 2)
    dd = tuple(dict.fromkeys((*da, *db)))
    od = tuple(rnd.sample(dd, len(dd)))
    dot = E.dot_sum((a, da), (b, db), od)
    letters = {n: chr(ord("a") + i) for i, n in enumerate(names)}
    ref = t.einsum(f"{''.join(letters[n] for n in da)}z,{''.join(letters[n] for n in db)}z->{''.join(letters[n] for n in od)}",
                   a.double().cpu(), b.double().cpu())
    t.testing.assert_close(dot.cpu().double(), ref, rtol=1e-5, atol=1e-5 * (float(ref.abs().max()) + 1.0))


def test_chain_terms_normal_rejects_malformed_descriptors():
    L = N.lib()
    x = t.zeros(1, 4, 3, 3, device=DEV)
    ptrs = (ctypes.c_void_p * 1)(x.data_ptr())
    strides = (ctypes.c_int64 * 4)(*x.stride())
    vec = t.empty(1, 3, device=DEV)
    nbytes = L.alan_chain_batched_workspace_bytes(1, 4, 3, 0)
    ws = t.empty(max(nbytes, 1), dtype=t.uint8, device=DEV)
    nd = N.ChainNormal()                                                 # null operands
    assert L.alan_chain_logmmexp_terms_final(ptrs, strides, 1, ctypes.byref(nd), None, 0, 1, 4, 3, None, vec.data_ptr(),
                                              ws.data_ptr(), nbytes, None) == -1
    assert L.alan_chain_logmmexp_terms_final(None, strides, 1, None, None, 0, 1, 4, 3, None, vec.data_ptr(), ws.data_ptr(),
                                              nbytes, None) == -1
    assert L.alan_chain_logmmexp_terms_final(ptrs, strides, 1, None, None, 0, 1, 4, 3, None, None, ws.data_ptr(), nbytes,
                                              None) == -1


@pytest.mark.parametrize("M,K,Nf,Ev,extra", [(300, 30, 5, 18, False), (300, 100, 5, 18, False), (7, 3, 5, 18, True), (33, 10, 1, 1, False),
                                             (5, 4, 40, 3, True), (2, 2, 9, 32, False)])
def test_linear_logits_gradient_launch_matches_autograd(M, K, Nf, Ev, extra):
    """alan_reduce mode BERNOULLI_LINEAR_GRAD (engine.bernoulli_linear_grad): d / d z of sum_n log Bernoulli(obs; z . x
    [+ a plain term]) weighted by a random upstream gradient, against fp64 autograd through the lambda's matmul and
    torch.distributions; z stored [K, plate, event] (the sample's layout) or the other way round."""
    g = t.Generator().manual_seed(M + K + Nf)
    for z_dims in (("k", "m"), ("m", "k")):
        shape = (K, M, Ev) if z_dims == ("k", "m") else (M, K, Ev)
        z = t.randn(*shape, generator=g).to(DEV)
        x = t.randn(M, Nf, Ev, generator=g).to(DEV)
        obs = (t.rand(M, Nf, generator=g) < 0.4).float().to(DEV)
        terms = [((z, z_dims), (x, ("m", "n")))]
        if extra:
            terms.append(((t.randn(M, generator=g).to(DEV), ("m",)),))
        out_dims = z_dims
        G = t.randn(*shape[:2], generator=g).to(DEV)
        got = E.bernoulli_linear_grad(G, (obs, ("m", "n")), terms, out_dims, scale=-0.5)
        assert got is not None and got.shape == z.shape
        zr = z.double().requires_grad_(True)
        rterms = [((zr, z_dims), (x, ("m", "n"))), *terms[1:]]
        ref = _ref(obs, ("m", "n"), rterms, out_dims)
        (want,) = t.autograd.grad((ref * G.double() * -0.5).sum(), [zr])
        t.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-6 * float(want.abs().max()) + 1e-7)


@pytest.mark.parametrize("fixture", ["e2e_movielens_K3.pt", "e2e_movielens_K10.pt"])
def test_movielens_vi_gradients_with_the_logits_gradient_launch(fixture, monkeypatch):
    """elbo_vi on movielens: the likelihood lambda `z @ x` and its gradient as library launches (dist.LINEAR_LOGITS_GRAD)
    against the same through torch's batched GEMMs and autograd -- ELBO and every parameter gradient."""
    from alan_amd import dist as D
    fx = load_golden(fixture)
    if fx["data"]["obs"][0].dtype != t.float32:
        fx = dict(fx, data={k: (v[0].float(), v[1]) for k, v in fx["data"].items()})

    def run(flag):
        monkeypatch.setattr(D, "LINEAR_LOGITS_GRAD", flag)
        prob = models.BUILDERS["movielens"](fx).to(DEV)
        t.manual_seed(5)
        t.cuda.manual_seed_all(5)
        calls = []
        real = E.bernoulli_linear_grad
        monkeypatch.setattr(E, "bernoulli_linear_grad", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        elbo = prob.sample(int(fixture.split("_K")[1].split(".")[0]), reparam=True).elbo_vi(alan.no_checkpoint)
        elbo.backward()
        monkeypatch.setattr(E, "bernoulli_linear_grad", real)
        return float(elbo), {n: p.grad.detach().double().cpu().clone() for n, p in prob.named_parameters() if p.grad is not None}, len(calls)

    e1, g1, n1 = run(True)
    e0, g0, n0 = run(False)
    assert n1 == 1 and n0 == 0
    assert abs(e1 - e0) <= 1e-5 * abs(e0)
    assert set(g1) == set(g0) and g1
    for n in g0:
        t.testing.assert_close(g1[n], g0[n], rtol=2e-4, atol=2e-5 * float(g0[n].abs().max()) + 1e-7, msg=lambda m: f"{n}: {m}")
