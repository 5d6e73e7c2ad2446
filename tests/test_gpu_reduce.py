"""GPU parity tests of the HIP contraction engine (through the C ABI) against the CPU oracle and
the reference-generated golden fixtures.  Tolerances: fp32 results within 2e-5 relative / 2e-5
absolute of the reference's own fp32 output (north_star: ELBO within 1e-4 relative); fp64 1e-10."""
import ctypes
import itertools
import math

import pytest
import torch as t

from conftest import load_golden
from oracle import alan_oracle as orc
from alan_amd import engine as E
from alan_amd import engine as E_
from alan_amd import native as N

pytestmark = pytest.mark.gpu
DEV = "cuda"


def tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == t.float32 else dict(rtol=1e-10, atol=1e-10)


def same(a, a_dims, b, b_dims, **kw):
    assert set(a_dims) == set(b_dims), (a_dims, b_dims)
    a = a.detach().cpu()
    a = orc.align((a, tuple(a_dims)), tuple(b_dims)) if a_dims else a
    kw = kw or tol(b.dtype)
    t.testing.assert_close(a.reshape(b.shape), b, equal_nan=True, **kw)


def dev(factors, requires_grad=False):
    return [(x.to(DEV).requires_grad_(requires_grad), d) for x, d in factors]


# ------------------------------------------------------------------ golden: logsumexp_dims
@pytest.mark.parametrize("case", load_golden("lse_dims.pt"), ids=lambda c: c["name"])
def test_logsumexp_dims_golden(case):
    if not case["reduce"]:
        pytest.skip("identity")
    out, dims = E.reduce_factors(dev([(case["x"], case["names"])]), reduce=case["reduce"])
    assert out.dtype == case["out"].dtype
    same(out, dims, case["out"], case["out_names"])
    if "mean_out" in case:
        sz = dict(zip(case["names"], case["x"].shape))
        c = -sum(math.log(sz[d]) for d in case["reduce"])
        out, dims = E.reduce_factors(dev([(case["x"], case["names"])]), reduce=case["reduce"], add_const=c)
        same(out, dims, case["mean_out"], case["mean_out_names"])


# ------------------------------------------------------------------ golden: reduce_Ks seam
@pytest.mark.parametrize("case", load_golden("seam_synthetic.pt"), ids=lambda c: c["name"])
def test_reduce_Ks_synthetic_golden(case):
    want_grad = "grads_weighted" in case
    factors = dev(case["factors"], want_grad)
    out, dims, _ = E.contract(factors, case["Ks"])
    assert out.dtype == case["out"].dtype
    same(out, dims, case["out"], case["out_names"])
    if not case["name"].startswith("all_neg_inf"):
        same(out.double(), dims, case["brute_f64"], case["brute_names"], rtol=2e-5, atol=2e-5)
    if want_grad:
        go = orc.align((case["grad_out"], case["out_names"]), dims) if dims else case["grad_out"]
        grads = t.autograd.grad(out, [x for x, _ in factors], go.reshape(out.shape).to(DEV))
        for g, ref in zip(grads, case["grads_weighted"]):
            t.testing.assert_close(g.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", load_golden("seam_recorded.pt"),
                         ids=lambda c: f"{c['model']}-K{c['K']}-{c['call']}")
def test_reduce_Ks_recorded_golden(case):
    out, dims, _ = E.contract(dev(case["factors"]), case["Ks"])
    assert out.dtype == case["out"].dtype
    same(out, dims, case["out"], case["out_names"], rtol=2e-5, atol=1e-4)


# ------------------------------------------------------------------ oracle: fused plate sum + backward
@pytest.mark.parametrize("K,M", [(3, 7), (10, 33), (30, 40)])
@pytest.mark.parametrize("layout", ["ref", "kz_first", "kz_mid"])
def test_movielens_plate_fused(K, M, layout):
    g = t.Generator().manual_seed(K * 100 + M)
    F = -0.5 * t.randn(M, K, K, K, generator=g) ** 2 - 0.92 - math.log(K)
    gz = -0.5 * t.randn(M, K, generator=g) ** 2 - 0.92 - math.log(K)
    Fd = ("plate_1", "K_mu", "K_psi", "K_z")
    if layout == "kz_first":
        F, Fd = F.permute(3, 0, 1, 2).contiguous(), ("K_z", "plate_1", "K_mu", "K_psi")
    elif layout == "kz_mid":
        F, Fd = F.permute(0, 3, 1, 2).contiguous(), ("plate_1", "K_z", "K_mu", "K_psi")
    cpu = [(F, Fd), (gz, ("plate_1", "K_z"))]
    (ref, ref_dims), ref_grads = orc.reduce_Ks_grads(cpu, ("K_z",), plate="plate_1")
    factors = dev(cpu, True)
    out, dims, _ = E.contract(factors, ("K_z",), plate=("plate_1",))
    same(out, dims, ref, ref_dims, rtol=2e-5, atol=2e-4)
    grads = t.autograd.grad(out.sum(), [x for x, _ in factors])
    for gd, rg in zip(grads, ref_grads):
        t.testing.assert_close(gd.cpu(), rg, rtol=1e-4, atol=1e-6)


def test_split_chunks_sum_to_full_plate():
    """logpq.py:43-57,151-153: summing per-chunk results == the unsplit plate (Split.py:84-95 sizes)."""
    K, M = 10, 50
    g = t.Generator().manual_seed(3)
    F = (-0.5 * t.randn(M, K, K, K, generator=g) ** 2).to(DEV)
    gz = (-0.5 * t.randn(M, K, generator=g) ** 2).to(DEV)
    full, fd, _ = E.contract([(F, ("p", "a", "b", "z")), (gz, ("p", "z"))], ("z",), plate=("p",))
    acc, start = None, 0
    for n in orc.split_sizes(M, 7):
        part, pd, _ = E.contract([(F[start:start + n], ("p", "a", "b", "z")), (gz[start:start + n], ("p", "z"))],
                                 ("z",), plate=("p",))
        assert pd == fd
        acc = part if acc is None else acc + part
        start += n
    t.testing.assert_close(acc, full, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------ oracle: layout / geometry fuzz
def _rand_case(seed):
    g = t.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(t.randint(lo, hi + 1, (1,), generator=g))
    ndim = ri(1, 5)
    names = [f"d{i}" for i in range(ndim)]
    sizes = {n: ri(1, 9) if ri(0, 3) else ri(20, 70) for n in names}
    nf = ri(1, 4)
    factors = []
    for f in range(nf):
        own = [n for n in names if ri(0, 2) > 0] or [names[0]]
        if f == 0:
            own = list(names)
        perm = [own[i] for i in t.randperm(len(own), generator=g).tolist()]
        dtype = t.float64 if ri(0, 5) == 0 else t.float32
        x = (t.randn([sizes[n] for n in perm], generator=g, dtype=dtype) * 3.0)
        if ri(0, 3) == 0 and x.ndim >= 2:           # non-contiguous view
            x = x.transpose(0, 1).contiguous().transpose(0, 1)
        factors.append((x, tuple(perm)))
    nred = ri(0, ndim)
    red = tuple(names[i] for i in t.randperm(ndim, generator=g).tolist()[:nred])
    rest = [n for n in names if n not in red]
    plate = tuple(rest[:1]) if (rest and ri(0, 2) == 0) else ()
    return factors, red, plate


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_against_oracle(seed):
    factors, red, plate = _rand_case(seed)
    ref = orc.logsumexp_sum(red, *factors)
    for p in plate:
        ref = orc.plate_sum(ref, p)
    out, dims = E.reduce_factors(dev(factors), reduce=red, plate=plate)
    assert out.dtype == ref[0].dtype
    same(out, dims, ref[0], ref[1], rtol=3e-5, atol=3e-4)


@pytest.mark.parametrize("n_out,n_red", [(1, 5), (1, 900), (1, 10000), (3, 70000), (5000, 2), (70000, 1),
                                         (257, 64), (64, 257), (100000, 30), (2, 300000)])
def test_geometry_extremes(n_out, n_red):
    g = t.Generator().manual_seed(n_out + n_red)
    x = t.randn(n_out, n_red, generator=g) * 5
    ref, _ = orc.logsumexp_dims((x, ("o", "r")), ("r",))
    out, dims = E.reduce_factors(dev([(x, ("o", "r"))]), reduce=("r",))
    same(out, dims, ref, ("o",), rtol=2e-5, atol=2e-5)
    xt = x.t().contiguous()
    out, dims = E.reduce_factors(dev([(xt, ("r", "o"))]), reduce=("r",))
    same(out, dims, ref, ("o",), rtol=2e-5, atol=2e-5)


def test_neg_inf_semantics():
    x = t.randn(6, 40)
    x[0, 3] = float("-inf")          # single -inf: fine
    x[1, 0] = float("-inf")          # leading -inf: fine
    x[2, :] = float("-inf")          # whole slice: NaN, as utils.py:218-220 gives (x - max = NaN)
    ref, _ = orc.logsumexp_dims((x, ("o", "r")), ("r",))
    assert t.isnan(ref[2]) and t.isfinite(ref[[0, 1, 3, 4, 5]]).all()
    for xx, dd in [(x, ("o", "r")), (x.t().contiguous(), ("r", "o"))]:
        out, dims = E.reduce_factors(dev([(xx, dd)]), reduce=("r",))
        same(out, dims, ref, ("o",))


def test_empty_Ks_is_identity_or_plain_sum():
    x = t.randn(10, 3)
    y = t.randn(3)
    out, dims, _ = E.contract(dev([(x, ("T", "K"))]), ())
    same(out, dims, x, ("T", "K"), rtol=0, atol=0)
    out, dims, _ = E.contract(dev([(x, ("T", "K")), (y, ("K",))]), ())
    same(out, dims, x + y, ("T", "K"), rtol=1e-6, atol=1e-6)
    out, dims, _ = E.contract(dev([(x, ("T", "K"))]), (), plate=("T",))
    same(out, dims, x.sum(0), ("K",), rtol=1e-6, atol=1e-5)


def test_errors_mirror_reference():
    x = t.randn(4, 3, device=DEV)
    with pytest.raises(Exception):
        E.contract([(x, ("a", "k"))], ("nope",))
    with pytest.raises(AssertionError):
        E.contract([(x, ("a",))], ())


# ------------------------------------------------------------------ chain (timeseries)
def _chain_input(case):
    if case["ms"] is not None:
        return case["ms"]
    g = t.Generator().manual_seed(case["seed"])
    dtype = getattr(t, case["dtype"].split(".")[-1])
    T, K = case["T"], case["K"]
    return -0.5 * t.randn(T, K, K, generator=g, dtype=dtype) ** 2 - 0.9189 - math.log(K)


@pytest.mark.parametrize("case", [c for c in load_golden("chain.pt") if "T" in c],
                         ids=lambda c: f"T{c['T']}K{c['K']}")
def test_chain_logmmexp_golden(case):
    ms = _chain_input(case).to(DEV)
    vec, chain, _ = N.chain_logmmexp(ms, want_chain=True)
    kw = dict(rtol=2e-5, atol=5e-5) if ms.dtype == t.float32 else dict(rtol=1e-10, atol=1e-10)
    t.testing.assert_close(chain.cpu(), case["chain"], **kw)
    t.testing.assert_close(vec.cpu(), case["out"], **kw)
    # strided input (time axis last in memory)
    ms2 = ms.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    vec2, _, _ = N.chain_logmmexp(ms2)
    t.testing.assert_close(vec2, vec, rtol=0, atol=0)


# ------------------------------------------------------------------ fused Normal factor producer
@pytest.mark.parametrize("M,K,E,dtype", [(7, 3, 18, t.float32), (40, 10, 18, t.float32), (300, 30, 18, t.float32),
                                         (5, 4, 1, t.float64), (9, 5, 7, t.float32),
                                         (301, 26, 5, t.float32), (13, 28, 31, t.float32)])   # (LDS-transposed stores)
def test_normal_producer_matches_torch_distributions(M, K, E, dtype):
    """alan_reduce(mode NORMAL) == td.Normal(loc, scale).log_prob(x).sum(event) over the K cross product
    (TorchDimDist.py:127-162, utils.py:147-152), fp32 to 2e-5 relative."""
    g = t.Generator().manual_seed(M * K + E)
    x = t.randn(M, K, E, generator=g, dtype=dtype)
    loc = t.randn(K, E, generator=g, dtype=dtype)
    scale = (0.3 * t.randn(K, E, generator=g, dtype=dtype)).exp()
    ref = t.distributions.Normal(loc[None, :, None, None, :], scale[None, None, :, None, :]).log_prob(
        x[:, None, None, :, :]).sum(-1)                      # [M, Kmu, Kpsi, Kz]
    out = E_.normal_logprob((x.to(DEV), ("m", "kz")), (loc.to(DEV), ("kmu",)), (scale.to(DEV), ("kpsi",)),
                            ("m", "kmu", "kpsi", "kz"))
    kw = dict(rtol=2e-5, atol=2e-4) if dtype == t.float32 else dict(rtol=1e-12, atol=1e-10)
    t.testing.assert_close(out.cpu(), ref, **kw)


def test_fused_and_torch_log_prob_paths_agree():
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT
    g = t.Generator().manual_seed(0)
    M, K, E = 12, 6, 18
    dm, dz, dmu, dpsi = Dim("plate_1", M), Dim("K_z", K), Dim("K_mu", K), Dim("K_psi", K)
    x = PT(t.randn(M, K, E, generator=g).to(DEV), (dm, dz))
    loc = PT(t.randn(K, E, generator=g).to(DEV), (dmu,))
    sc = PT(t.rand(K, E, generator=g).to(DEV) + 0.5, (dpsi,))
    order = ([dm], [dz])
    D.FUSE_NORMAL = True
    a = D.TorchDimDist(t.distributions.Normal, loc=loc, scale=sc).log_prob_pt(x, order)
    D.FUSE_NORMAL = False
    try:
        b = D.TorchDimDist(t.distributions.Normal, loc=loc, scale=sc).log_prob_pt(x, order)
    finally:
        D.FUSE_NORMAL = True
    assert [str(d) for d in a.dims] == [str(d) for d in b.dims] == ["plate_1", "K_mu", "K_psi", "K_z"]
    t.testing.assert_close(a.x, b.x, rtol=2e-5, atol=2e-4)
    assert a.x.is_contiguous()


@pytest.mark.parametrize("case", [c for c in load_golden("chain.pt") if "T" in c],
                         ids=lambda c: f"T{c['T']}K{c['K']}")
def test_chain_backward_golden(case):
    """d/d ms of sum(grad_out * logsumexp(chain_logmmexp(ms), -1)) vs the reference's autograd."""
    from alan_amd.contract import chain_logmmexp_lse
    ms = _chain_input(case).to(DEV).requires_grad_(True)
    out = chain_logmmexp_lse(ms)
    (grad,) = t.autograd.grad(out, ms, case["grad_out"].to(DEV))
    if case["grad"] is not None:
        kw = dict(rtol=1e-4, atol=1e-5) if ms.dtype == t.float32 else dict(rtol=1e-9, atol=1e-10)
        t.testing.assert_close(grad.cpu(), case["grad"], **kw)
    s, a = case["grad_checksum"]
    # the gradient's total mass is sum_i g_i exp(chain[i,:] - out[i]): an fp32 `out` of magnitude ~2000 (T = 1000) is
    # only known to 2.4e-4, in the reference's run as much as in ours
    tol = 2e-4 + 2 * float(t.finfo(ms.dtype).eps) * float(case["out"].abs().max())
    assert abs(float(grad.double().sum()) - s) <= tol * max(1.0, a)
    assert abs(float(grad.double().abs().sum()) - a) <= tol * max(1.0, a)


PEAKED = load_golden("chain_peaked.pt")


@pytest.mark.parametrize("case", PEAKED, ids=[f"B{c['B']}T{c['T']}K{c['K']}" for c in PEAKED])
def test_chain_on_the_eps_floor_matches_the_reference_forward_and_backward(case):
    """Sharply peaked transition matrices (a random walk with a small noise scale): 23-38 % of the chain's entries
    sit on the floor that the +eps inside the log creates (utils.py:506), so the value depends on the tree's
    bracketing and autograd differentiates through the floor and through amax.  The kernels keep the reference's
    tree and differentiate it node by node: values and gradients agree with the reference's own."""
    from alan_amd.contract import chain_logmmexp_lse, chain_logmmexp
    assert case["floored"] > 0.2
    f32 = case["ms"].dtype == t.float32
    for b in range(case["B"]):                               # unbatched entry points, one chain at a time
        ms = case["ms"][b].to(DEV).requires_grad_(True)
        out = chain_logmmexp_lse(ms)
        t.testing.assert_close(out.cpu(), case["out"][b], **(dict(rtol=2e-5, atol=2e-4) if f32 else dict(rtol=1e-10, atol=1e-9)))
        (grad,) = t.autograd.grad(out, ms, case["grad_out"][b].to(DEV))
        kw = dict(rtol=2e-3, atol=2e-5) if f32 else dict(rtol=1e-8, atol=1e-10)
        t.testing.assert_close(grad.cpu(), case["grad"][b], **kw)
        if b:
            continue
        ms2 = case["ms"][b].to(DEV).requires_grad_(True)     # the full [K,K] chain and ITS backward
        ch = chain_logmmexp(ms2)
        t.testing.assert_close(ch.cpu(), case["chain"][b], **(dict(rtol=2e-5, atol=2e-4) if f32 else dict(rtol=1e-10, atol=1e-9)))
        (gc,) = t.autograd.grad(ch, ms2, case["grad_chain_out"][b].to(DEV))
        # one product of the batch's gradient: compare against the reference's per-chain slice
        if case["B"] == 1:
            t.testing.assert_close(gc.cpu(), case["grad_chain"][b], **(dict(rtol=2e-3, atol=2e-4) if f32 else dict(rtol=1e-8, atol=1e-9)))
    msb = case["ms"].to(DEV).requires_grad_(True)            # the batched entry points
    outb = chain_logmmexp_lse(msb)
    t.testing.assert_close(outb.cpu(), case["out"], **(dict(rtol=2e-5, atol=2e-4) if f32 else dict(rtol=1e-10, atol=1e-9)))
    (gb,) = t.autograd.grad(outb, msb, case["grad_out"].to(DEV))
    t.testing.assert_close(gb.cpu(), case["grad"], **(dict(rtol=2e-3, atol=2e-5) if f32 else dict(rtol=1e-8, atol=1e-10)))
    msc = case["ms"].to(DEV).requires_grad_(True)
    (gcb,) = t.autograd.grad(chain_logmmexp(msc), msc, case["grad_chain_out"].to(DEV))
    t.testing.assert_close(gcb.cpu(), case["grad_chain"], **(dict(rtol=2e-3, atol=2e-4) if f32 else dict(rtol=1e-8, atol=1e-9)))


@pytest.mark.parametrize("shapes", ["movielens", "shared_dims", "scalar_event"])
def test_fused_normal_backward_matches_torch_autograd(shapes):
    """Gradients of the fused Normal producer (HIP forward + einsum backward) vs torch.distributions autograd."""
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT
    g = t.Generator().manual_seed(3)
    M, K, Ev = 9, 5, 18
    dm, dz, dmu, dpsi = Dim("plate_1", M), Dim("K_z", K), Dim("K_mu", K), Dim("K_psi", K)
    if shapes == "movielens":
        spec = [((M, K, Ev), (dm, dz)), ((K, Ev), (dmu,)), ((K, Ev), (dpsi,))]
    elif shapes == "shared_dims":       # loc and scale share a dim with each other and with the value
        spec = [((M, K, Ev), (dm, dz)), ((M, K, Ev), (dm, dmu)), ((K, Ev), (dmu,))]
    else:
        spec = [((M, K), (dm, dz)), ((K,), (dmu,)), ((K,), (dpsi,))]
    raw = [t.randn(s, generator=g) for s, _ in spec]
    raw[2] = raw[2].abs() + 0.5

    def run(fuse):
        D.FUSE_NORMAL = fuse
        try:
            leaves = [r.clone().to(DEV).requires_grad_(True) for r in raw]
            pts = [PT(x, d) for x, (_, d) in zip(leaves, spec)]
            lp = D.TorchDimDist(t.distributions.Normal, loc=pts[1], scale=pts[2]).log_prob_pt(pts[0], ([dm], [dz]))
            wgt = t.randn(lp.x.shape, generator=t.Generator().manual_seed(1)).to(DEV)
            grads = t.autograd.grad((lp.x * wgt).sum(), leaves)
            return lp, grads
        finally:
            D.FUSE_NORMAL = True

    (lp_f, gf), (lp_t, gt) = run(True), run(False)
    assert [str(d) for d in lp_f.dims] == [str(d) for d in lp_t.dims]
    t.testing.assert_close(lp_f.x, lp_t.x, rtol=2e-5, atol=2e-4)
    for a, b in zip(gf, gt):
        t.testing.assert_close(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()))


# ------------------------------------------------------------------ fused Bernoulli(logits) producer
@pytest.mark.parametrize("vdtype,ldtype", [(t.float32, t.float32), (t.float64, t.float32), (t.float64, t.float64)])
@pytest.mark.parametrize("plate_sum", [False, True])
def test_bernoulli_producer_matches_torch_distributions(vdtype, ldtype, plate_sum):
    """alan_reduce(mode BERNOULLI) == td.Bernoulli(logits=l).log_prob(y) over the broadcast of value [m, n] and
    logits [m, n, kz], optionally with the data plate n summed by the same launch (logpq.py:149).  Large |logits|
    exercise both branches of logsigmoid."""
    g = t.Generator().manual_seed(11)
    M, Nf, K = 37, 5, 30
    y = (t.rand(M, Nf, generator=g) < 0.4).to(vdtype)
    l = (6 * t.randn(M, Nf, K, generator=g)).to(ldtype)
    l[0, 0, :4] = t.tensor([-90.0, 90.0, 0.0, -1e-8]).to(ldtype)
    ref = t.distributions.Bernoulli(logits=l).log_prob(y[:, :, None].to(t.promote_types(vdtype, ldtype)))
    out_dims = ("m", "kz") if plate_sum else ("m", "n", "kz")
    if plate_sum:
        ref = ref.sum(1)
    out = E_.bernoulli_logprob((y.to(DEV), ("m", "n")), (l.to(DEV), ("m", "n", "kz")), out_dims)
    assert out.dtype == ref.dtype
    # (mixed dtypes: torch evaluates logsigmoid in the logits' fp32 and only then promotes)
    kw = dict(rtol=2e-5, atol=2e-5) if t.float32 in (vdtype, ldtype) else dict(rtol=1e-12, atol=1e-10)
    t.testing.assert_close(out.cpu(), ref, **kw)


@pytest.mark.parametrize("with_event", [False, True])
def test_fused_bernoulli_forward_and_backward_match_torch_path(with_event):
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT
    g = t.Generator().manual_seed(5)
    M, Nf, K, Ev = 6, 4, 5, 3
    dm, dn, dz = Dim("plate_1", M), Dim("plate_2", Nf), Dim("K_z", K)
    ev = (Ev,) if with_event else ()
    y = PT((t.rand(M, Nf, *ev, generator=g) < 0.5).float().to(DEV), (dm, dn))
    raw = t.randn(M, K, Nf, *ev, generator=g)               # logits stored [plate_1, K_z, plate_2, event]

    def run(fuse, sum_dims):
        D.FUSE_NORMAL = fuse
        try:
            leaf = raw.clone().to(DEV).requires_grad_(True)
            dist = D.TorchDimDist(t.distributions.Bernoulli, logits=PT(leaf, (dm, dz, dn)))
            lp = dist.log_prob_pt(y, ([dm], []), sum_dims=sum_dims)
            wgt = t.randn(lp.x.shape, generator=t.Generator().manual_seed(1)).to(DEV)
            (grad,) = t.autograd.grad((lp.x * wgt).sum(), [leaf])
            return lp, grad
        finally:
            D.FUSE_NORMAL = True

    for sum_dims in ((), (dn,)):
        (lp_f, gf), (lp_t, gt) = run(True, sum_dims), run(False, sum_dims)
        assert [str(d) for d in lp_f.dims] == [str(d) for d in lp_t.dims]
        assert ("plate_2" in [str(d) for d in lp_f.dims]) == (not sum_dims)
        t.testing.assert_close(lp_f.x, lp_t.x, rtol=2e-5, atol=2e-5)
        t.testing.assert_close(gf, gt, rtol=1e-4, atol=1e-5)


def test_normal_producer_sums_a_data_plate():
    """A Normal likelihood in a data-only plate: the plate sum rides in the producer launch."""
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT
    g = t.Generator().manual_seed(9)
    M, Nf, K = 8, 6, 4
    dm, dn, dz = Dim("plate_1", M), Dim("plate_2", Nf), Dim("K_z", K)
    y = PT(t.randn(M, Nf, generator=g).to(DEV), (dm, dn))
    loc = PT(t.randn(M, K, generator=g).to(DEV), (dm, dz))
    sc = PT((t.rand(Nf, generator=g) + 0.5).to(DEV), (dn,))
    dist = D.TorchDimDist(t.distributions.Normal, loc=loc, scale=sc)
    full = dist.log_prob_pt(y, ([dm, dn], []))
    summed = dist.log_prob_pt(y, ([dm], []), sum_dims=(dn,))
    assert [str(d) for d in summed.dims] == ["plate_1", "K_z"]
    pos = [str(d) for d in full.dims].index("plate_2")
    t.testing.assert_close(summed.x, full.x.sum(pos), rtol=2e-5, atol=2e-5)
    with pytest.raises(Exception, match="neither the value nor the parameters"):
        dist.log_prob_pt(y, ([dm], []), sum_dims=(Dim("plate_9", 3),))


@pytest.mark.parametrize("shapes", ["movielens", "mean_field_q"])
def test_fused_normal_log_scale_and_affine(shapes):
    """Normal(loc, exp(raw)) with the exp folded into the producer (mode NORMAL_LOGSCALE) and the
    -(log Q + log K) epilogue (out.scale / add_const): values and gradients wrt (value, loc, RAW scale) against
    the unfused path (exp kernel + torch.distributions + t.add)."""
    import math
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT, ExpPT
    g = t.Generator().manual_seed(4)
    M, K, Ev = 9, 5, 18
    dm, dz, dmu, dpsi = Dim("plate_1", M), Dim("K_z", K), Dim("K_mu", K), Dim("K_psi", K)
    if shapes == "movielens":
        spec = [((M, K, Ev), (dm, dz)), ((K, Ev), (dmu,)), ((K, Ev), (dpsi,))]
    else:                                # q(z) = N(loc[m], exp(raw[m])): the shape the affine epilogue applies to
        spec = [((M, K, Ev), (dm, dz)), ((M, Ev), (dm,)), ((M, Ev), (dm,))]
    raw = [t.randn(s, generator=g) for s, _ in spec]
    raw[2] = 0.4 * raw[2]
    own = {id(dm), id(dz)}
    aff = (-1.0, -math.log(K), own)

    def run(fuse):
        D.FUSE_NORMAL = fuse
        try:
            leaves = [r.clone().to(DEV).requires_grad_(True) for r in raw]
            x, loc = PT(leaves[0], spec[0][1]), PT(leaves[1], spec[1][1])
            sc = ExpPT(leaves[2], spec[2][1])
            lp = D.TorchDimDist(t.distributions.Normal, loc=loc, scale=sc).log_prob_pt(x, ([dm], [dz]), affine=aff)
            assert sc.materialised == (not fuse)
            wgt = t.randn(lp.x.shape, generator=t.Generator().manual_seed(1)).to(DEV)
            grads = t.autograd.grad((lp.x * wgt).sum(), leaves)
            return lp, grads
        finally:
            D.FUSE_NORMAL = True

    (lp_f, gf), (lp_t, gt) = run(True), run(False)
    assert [str(d) for d in lp_f.dims] == [str(d) for d in lp_t.dims]
    t.testing.assert_close(lp_f.x, lp_t.x, rtol=2e-5, atol=2e-4)
    plain = D.TorchDimDist(t.distributions.Normal, loc=PT(raw[1].to(DEV), spec[1][1]),
                           scale=PT(raw[2].to(DEV).exp(), spec[2][1])).log_prob_pt(PT(raw[0].to(DEV), spec[0][1]),
                                                                                  ([dm], [dz]))
    applied = {id(d) for d in plain.dims} <= own
    assert applied == (shapes == "mean_field_q")
    want = (-plain.x - math.log(K)) if applied else plain.x
    t.testing.assert_close(lp_f.x, want, rtol=2e-5, atol=2e-4)
    for a, b in zip(gf, gt):
        t.testing.assert_close(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max()))


@pytest.mark.parametrize("q_log_scale", [False, True])
def test_normal_p_minus_q_single_launch(q_log_scale):
    """log P - log Q - log K of a latent with Normal prior and Normal approximate posterior on the same dims
    (logpq.py:221-235) as ONE alan_reduce(mode NORMAL, 6 factors) launch vs the two separate log-probs."""
    import math
    import alan_amd.dist as D
    from alan_amd.dims import Dim, PT, ExpPT
    g = t.Generator().manual_seed(8)
    M, K, Ev = 7, 6, 18
    dm, dz = Dim("plate_1", M), Dim("K_z", K)
    x = PT(t.randn(M, K, Ev, generator=g).to(DEV), (dm, dz))
    P = D.TorchDimDist(t.distributions.Normal, loc=PT(t.tensor(0.3, device=DEV), ()),
                       scale=PT(t.tensor(1.7, device=DEV), ()))
    qraw = 0.3 * t.randn(M, Ev, generator=g).to(DEV)
    qscale = ExpPT(qraw, (dm,)) if q_log_scale else PT(qraw.exp(), (dm,))
    Q = D.TorchDimDist(t.distributions.Normal, loc=PT(t.randn(M, Ev, generator=g).to(DEV), (dm,)), scale=qscale)
    own = {id(dm), id(dz)}
    with t.no_grad():
        pq = D.TorchDimDist.log_p_minus_q(P, Q, x, ([dm], [dz]), own, math.log(K))
    assert pq is not None and [str(d) for d in pq.dims] == ["plate_1", "K_z"]
    if q_log_scale:
        assert not qscale.materialised
    want = P.log_prob_pt(x, ([dm], [dz])).x - Q.log_prob_pt(x, ([dm], [dz])).x - math.log(K)
    t.testing.assert_close(pq.x, want, rtol=2e-5, atol=2e-4)
    # does not apply: a prior carrying a parent K, or a gradient to record
    dmu = Dim("K_mu", K)
    P2 = D.TorchDimDist(t.distributions.Normal, loc=PT(t.randn(K, Ev, generator=g).to(DEV), (dmu,)),
                        scale=PT(t.tensor(1.0, device=DEV), ()))
    with t.no_grad():
        assert D.TorchDimDist.log_p_minus_q(P2, Q, x, ([dm], [dz]), own, math.log(K)) is None
    Q3 = D.TorchDimDist(t.distributions.Normal, loc=PT(t.randn(M, Ev, generator=g).to(DEV).requires_grad_(True), (dm,)),
                        scale=PT(qraw.exp(), (dm,)))
    assert D.TorchDimDist.log_p_minus_q(P, Q3, x, ([dm], [dz]), own, math.log(K)) is None


# ------------------------------------------------------------------ deferred small launches (alan_reduce_batch)
@pytest.mark.parametrize("n_out,n_red", [(20000, 40), (9000, 18), (300, 700), (70000, 6)])
def test_a_batch_bigger_than_the_chip_gives_up_lanes_and_keeps_the_values(n_out, n_red):
    """Eight problems that together would be several chipfuls of workgroups with the lanes each takes alone: the
    multi-problem launch sizes them together (plan.h fill_small_multi: lanes given up until the launch is ~1,024
    workgroups, problems dealt by the work of a lane) -- every output equals the lone launch's to rounding, whatever the
    mode: Normal producers (log and plain scale, broadcast locations), Bernoulli producers."""
    from alan_amd.dims import Dim
    g = t.Generator().manual_seed(n_out + n_red)
    do, dr = Dim("o", n_out), Dim("r", n_red)
    r = lambda *s: t.randn(*s, generator=g).to(DEV)
    x, loc, sc = r(n_out, n_red), r(n_out, n_red), 0.3 * r(n_out, n_red)
    y, lg = (t.rand(n_out, n_red, generator=g) < 0.5).float().to(DEV), r(n_out, n_red)
    f1, f2 = r(n_out, n_red), r(n_red)
    calls = [
        lambda: E.normal_logprob((x, (do,)), (loc, (do,)), (sc, (do,)), (do,), log_scale=True),
        lambda: E.normal_logprob((x, (do,)), (loc, (do,)), (sc.exp(), (do,)), (do,)),
        lambda: E.bernoulli_logprob((y, (do, dr)), (lg, (do, dr)), (do,)),
        lambda: E.normal_logprob((f1, (do,)), (f2, ()), (sc.exp(), (do,)), (do,)),
        lambda: E.normal_logprob((loc, (do,)), (x, (do,)), (sc, (do,)), (do,), log_scale=True, affine=(-1.0, 0.25)),
        lambda: E.bernoulli_logprob((y, (do,)), (lg, (do,)), (do,), affine=(2.0, -1.0)),
        lambda: E.bernoulli_logprob((y, (do, dr)), (f1, (do, dr)), (do,)),
        lambda: E.normal_logprob((f1, (do,)), (lg, (do,)), (sc, (do,)), (do,), log_scale=True),
    ]
    want = [c() for c in calls]
    t.cuda.synchronize()
    with t.no_grad(), N.deferring():
        with N.may_defer():
            got = [c() for c in calls]
        queued = N.n_pending()
        N.flush()
    assert queued >= 6                                         # (a problem the small kernel does not take goes out on its own)
    for a, b in zip(got, want):
        t.testing.assert_close(a, b, rtol=3e-6, atol=3e-6 * float(b.abs().max()))


def test_deferred_producers_go_out_as_one_launch_and_match_immediate_ones():
    """Independent per-variable producers queued under native.deferring() / may_defer() and issued by
    alan_reduce_batch (small ones as ONE multi-problem kernel, the big factor on its own) write the results of the same
    launches issued one by one."""
    from alan_amd.dims import Dim
    g = t.Generator().manual_seed(11)
    M, K, Ev = 300, 30, 18
    dm, dz, dmu, dpsi, dn = Dim("plate_1", M), Dim("K_z", K), Dim("K_mu", K), Dim("K_psi", K), Dim("plate_2", 5)
    da = Dim("a", 7)
    r = lambda *s: t.randn(*s, generator=g).to(DEV)
    z, mu, psi = r(M, K, Ev), r(K, Ev), r(K, Ev)
    obs, logits = (t.rand(M, 5, generator=g) < 0.5).float().to(DEV), r(M, K, 5)
    calls = [
        lambda: E.normal_logprob_pq((mu, (dmu,)), ((t.zeros(Ev, device=DEV), ()), (t.ones(Ev, device=DEV), ()), False),
                                    ((r(Ev), ()), (r(Ev), ()), True), (dmu,), affine=(1.0, -math.log(K))),
        lambda: E.normal_logprob_pq((psi, (dpsi,)), ((t.zeros(Ev, device=DEV), ()), (t.ones(Ev, device=DEV), ()), False),
                                    ((r(Ev), ()), (r(Ev), ()), True), (dpsi,), affine=(1.0, -math.log(K))),
        lambda: E.normal_logprob((z, (dm, dz)), (mu, (dmu,)), (psi.exp(), (dpsi,)), (dm, dmu, dpsi, dz)),   # the big one
        lambda: E.normal_logprob((z, (dm, dz)), (r(M, Ev), (dm,)), (r(M, Ev), (dm,)), (dm, dz), log_scale=True,
                                 affine=(-1.0, -math.log(K))),
        lambda: E.bernoulli_logprob((obs, (dm, dn)), (logits, (dm, dz, dn)), (dm, dz)),
        lambda: E.normal_logprob((r(7, 3), (da,)), (r(3), ()), (t.ones(3, device=DEV), ()), (da,)),
    ]
    gen_state = g.get_state()
    want = [c() for c in calls]
    t.cuda.synchronize()
    g.set_state(gen_state)
    assert N.n_pending() == 0
    with t.no_grad(), N.deferring():
        with N.may_defer():
            got = [c() for c in calls]
            assert N.n_pending() == len(calls)          # nothing launched yet
        # a consumer flushes the queue before it reads
        total, _ = E.reduce_factors([(got[3], (dm, dz)), (got[4], (dm, dz))], reduce=(dz,), plate=(dm,))
        assert N.n_pending() == 0
    # (the same values, not the same bits: a multi-problem launch may give its biggest problems fewer lanes per output than
    # they take alone -- plan.h fill_small_multi, round 4 -- and a sum over fewer lanes is added up in another order)
    for a, b in zip(got, want):
        t.testing.assert_close(a, b, rtol=2e-6, atol=2e-6 * float(b.abs().max()))
    ref, _ = E.reduce_factors([(want[3], (dm, dz)), (want[4], (dm, dz))], reduce=(dz,), plate=(dm,))
    t.testing.assert_close(total, ref, rtol=2e-6, atol=2e-6 * float(ref.abs().max()))
    # outside deferring(), or with gradients enabled, may_defer() is inert
    with N.may_defer():
        calls[0]()
        assert N.n_pending() == 0


def test_reduce_batch_c_abi_orders_and_rejects():
    L = N.lib()
    assert L.alan_reduce_batch(None, 0, None) == -1
    arr = (ctypes.POINTER(N.ReduceDesc) * 1)(ctypes.POINTER(N.ReduceDesc)())
    assert L.alan_reduce_batch(arr, 1, None) == -1       # null descriptor


# ------------------------------------------------------------------ backward of the fused producers (PRODUCER_GRAD)
def test_producer_grads_match_torch_distributions_autograd():
    """alan_reduce mode PRODUCER_GRAD (d/d value, d/d loc, d/d scale or log scale of a Normal; d/d logits of a
    Bernoulli) against autograd through torch.distributions on the materialised broadcast, over random dim
    assignments: shared and private first-class dims, event or scalar parameters, fp32 and fp64, an affine output."""
    import random
    from alan_amd.dims import Dim
    rnd = random.Random(3)
    g = t.Generator().manual_seed(3)
    pool = [Dim("a", 3), Dim("b", 4), Dim("c", 5), Dim("d", 2)]
    has = lambda ds, d: any(x is d for x in ds)          # (Dim overloads ==)
    for it in range(25):
        dtype = t.float64 if it % 5 == 4 else t.float32
        Ev = rnd.choice([1, 3, 7])
        pick = lambda: tuple(d for d in pool if rnd.random() < 0.5)
        vd, ld, sd = pick() or (pool[0],), pick(), pick()
        od = tuple(d for d in pool if has(vd, d) or has(ld, d) or has(sd, d))
        od = tuple(d for d in od if rnd.random() < 0.7) or od[:1]
        mk = lambda ds, ev: t.randn(*[d.size for d in ds], *([Ev] if ev else []), generator=g, dtype=dtype).to(DEV)
        value, loc = mk(vd, True), mk(ld, rnd.random() < 0.6)
        raw = 0.3 * mk(sd, rnd.random() < 0.6)
        log_scale, a = rnd.random() < 0.5, rnd.choice([1.0, -1.0, 0.5])
        G = t.randn(*[d.size for d in od], generator=g, dtype=dtype).to(DEV)

        def full(x, ds):          # broadcast over the whole pool, event last
            idx = tuple(slice(None) if has(ds, d) else None for d in pool)
            x = x if x.ndim > len(ds) else x.unsqueeze(-1)
            return x[idx + (slice(None),)]

        leaves = [x.clone().requires_grad_(True) for x in (value, loc, raw)]
        sc = leaves[2].exp() if log_scale else leaves[2].exp().detach().requires_grad_(True)
        lp = t.distributions.Normal(full(leaves[1], ld), full(sc, sd)).log_prob(full(leaves[0], vd)).sum(-1)
        red = [i for i, d in enumerate(pool) if not has(od, d)]
        lp = lp.sum(red) if red else lp
        keep = [d for d in pool if has(od, d)]
        lp = lp.reshape([d.size for d in keep])
        perm = [[i for i, k in enumerate(keep) if k is d][0] for d in od]
        want = t.autograd.grad(((a * lp).permute(perm) * G).sum(), [leaves[0], leaves[1], leaves[2] if log_scale else sc])
        scale_arg = raw if log_scale else raw.exp()
        with t.no_grad():
            got = E.producer_grads(G, od, [(value, vd), (loc, ld), (scale_arg, sd)], (True, True, True),
                                   (N.GRAD_VALUE, N.GRAD_LOC, N.GRAD_SCALE), log_scale=log_scale, scale=a)
        assert got is not None
        kw = dict(rtol=2e-4, atol=2e-4) if dtype == t.float32 else dict(rtol=1e-9, atol=1e-9)
        for name, x, w in zip(("value", "loc", "scale"), got, want):
            t.testing.assert_close(x, w.reshape(x.shape), msg=lambda m: f"case {it} d/d {name}: {m}", **kw)
    # Bernoulli(logits)
    dm, dk, dn = Dim("m", 6), Dim("k", 4), Dim("n", 5)
    y = (t.rand(6, 5, generator=g) < 0.5).float().to(DEV)
    x = t.randn(6, 4, 5, generator=g).to(DEV)
    G = t.randn(6, 4, generator=g).to(DEV)
    xl = x.clone().requires_grad_(True)
    lp = t.distributions.Bernoulli(logits=xl).log_prob(y[:, None, :]).sum(-1)
    (want,) = t.autograd.grad((-0.5 * lp * G).sum(), xl)
    with t.no_grad():
        got = E.producer_grads(G, (dm, dk), [(y, (dm, dn)), (x, (dm, dk, dn))], (False, True), (0.0, N.GRAD_LOGITS), scale=-0.5)
    t.testing.assert_close(got[1], want, rtol=2e-5, atol=2e-5)
    # fp64 observations with fp32 logits (real data sets): computed in fp64, handed back in the logits' dtype -- and
    # only cast AFTER the queued launch has gone out (a cast inside the queueing block once read unwritten memory)
    with t.no_grad():
        mixed = E.producer_grads(G.double(), (dm, dk), [(y.double(), (dm, dn)), (x, (dm, dk, dn))], (False, True),
                                 (0.0, N.GRAD_LOGITS), scale=-0.5)
    assert mixed[1].dtype == t.float32
    t.testing.assert_close(mixed[1], want, rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------ few outputs over a huge reduce space
@pytest.mark.parametrize("shape,keys,reduce,dtype", [
    ((300, 30, 30, 30), ("m", "a", "b", "z"), ("m", "a", "z"), t.float32),
    ((64, 64, 64), ("a", "b", "c"), ("a", "b", "c"), t.float32),
    ((50, 40, 30, 7), ("m", "a", "b", "k"), ("m", "a", "b"), t.float64),
    ((4096, 33), ("r", "c"), ("r", "c"), t.float32),
    ((270000, 30), ("r", "c"), ("r",), t.float32), ((65536 * 3,), ("r",), ("r",), t.float32),
    ((100003, 5), ("r", "c"), ("r",), t.float32),       # (a prime length: no split, one workgroup per output)
    ((9000, 540), ("r", "c"), ("r",), t.float32), ((64, 70, 600), ("a", "b", "c"), ("a", "b"), t.float32)])
def test_few_outputs_over_a_huge_reduce_space_take_two_launches_and_agree(shape, keys, reduce, dtype):
    """logsumexp_dims / sums of a whole factor down to a handful of values (alan_reduce peels the largest reduce dim
    into a first launch): against the oracle, with a second broadcast factor, in both modes."""
    g = t.Generator().manual_seed(sum(shape))
    x = (-0.5 * t.randn(*shape, generator=g, dtype=t.float64) ** 2 - 0.9).to(dtype)
    small_keys = tuple(k for k in keys if k not in reduce) or (keys[-1],)
    y = t.randn(*[shape[keys.index(k)] for k in small_keys], generator=g, dtype=t.float64).to(dtype)
    facs = [(x.to(DEV), keys), (y.to(DEV), small_keys)]
    got, gd = E.reduce_factors(facs, reduce=reduce)
    want = orc.align(orc.reduce_Ks([(x.double(), keys), (y.double(), small_keys)], reduce), tuple(gd))
    kw = dict(rtol=2e-5, atol=2e-5) if dtype == t.float32 else dict(rtol=1e-10, atol=1e-10)
    t.testing.assert_close(got.cpu().double(), want, **kw)
    s_got, sd = E.reduce_factors([(x.to(DEV), keys)], plate=reduce)
    s_want = x.double().sum([keys.index(k) for k in reduce])
    t.testing.assert_close(s_got.cpu().double().reshape(s_want.shape), s_want, rtol=2e-5 if dtype == t.float32 else 1e-10,
                           atol=1e-3 if dtype == t.float32 else 1e-8)


@pytest.mark.parametrize("seed", range(14))
def test_large_few_output_reductions_fuzz(seed):
    """Random 2-8 Mi element problems with few outputs (the two-launch peel / long-dim split of alan_reduce, and its
    neighbours that stay single-launch): random dims, roles, permuted storage, a second broadcast factor, log-sum-exp
    and sum, against torch in fp64 on the device."""
    import random
    rng = random.Random(100 + seed)
    g = t.Generator().manual_seed(100 + seed)
    nd = rng.randint(2, 4)
    names = ["a", "b", "c", "d"][:nd]
    target = rng.choice([2, 4, 8]) * (1 << 20)
    sizes = [rng.choice([2, 3, 5, 8, 30, 64, 100]) for _ in range(nd - 1)]
    rest = 1
    for s_ in sizes:
        rest *= s_
    sizes.insert(rng.randrange(nd), max(2, target // rest))
    keep = [n for n, s_ in zip(names, sizes) if s_ <= 64 and rng.random() < 0.4]
    keep_sz = 1
    for n in keep:
        keep_sz *= sizes[names.index(n)]
    if keep_sz > 4096:
        keep = keep[:1]
    red = tuple(n for n in names if n not in keep)
    x = (-0.5 * t.randn(*sizes, generator=g) ** 2 - 0.9).to(DEV)
    perm = list(range(nd))
    rng.shuffle(perm)
    inv = [perm.index(i) for i in range(nd)]
    x = x.permute(*perm).contiguous().permute(*inv)                     # same values, shuffled storage order
    small = tuple(n for n in names if rng.random() < 0.5) or (names[0],)
    y = t.randn(*[sizes[names.index(n)] for n in small], generator=g).to(DEV)
    facs = [(x, tuple(names)), (y, small)]
    got, gd = E.reduce_factors(facs, reduce=red)
    full = x.double() + y.double()[tuple(slice(None) if n in small else None for n in names)]
    axes = [names.index(n) for n in red]
    want = t.logsumexp(full, axes) if axes else full
    kd = [n for n in names if n not in red]
    want = want.permute([kd.index(n) for n in gd]) if len(kd) > 1 else want
    t.testing.assert_close(got.double(), want, rtol=3e-5, atol=3e-5, msg=lambda m_: f"{sizes} keep {keep}: {m_}")
    s_got, sd = E.reduce_factors([(x, tuple(names))], plate=red)
    s_want = x.double().sum(axes) if axes else x.double()
    s_want = s_want.permute([kd.index(n) for n in sd]) if len(kd) > 1 else s_want
    t.testing.assert_close(s_got.double(), s_want, rtol=3e-5, atol=2e-2 * (target / (1 << 21)) ** 0.5)


@pytest.mark.gpu
def test_small_strided_view_into_a_huge_tensor_does_not_wrap_32_bit_offsets():
    """A 3 x 5 problem whose rows sit 2^30 elements apart inside a tensor of more than 2^31 elements (a strided slice
    of a K=100 factor has this shape): every stride fits 31 bits, the largest reachable offset does not -- the compact
    small-problem kernel (32-bit offsets) must decline it and the 64-bit kernel take it."""
    big = t.empty((1 << 31) + 64, dtype=t.float32, device="cuda")
    view = big.as_strided((3, 5), (1 << 30, 7))
    vals = t.randn(3, 5, generator=t.Generator().manual_seed(0))
    view.copy_(vals.to("cuda"))
    out, dims = E.reduce_factors([(view, ("a", "k"))], reduce=("k",))
    want = t.logsumexp(vals.double(), 1)
    got = out if dims == ("a",) else out
    t.testing.assert_close(got.cpu().double(), want, rtol=2e-6, atol=2e-6)
    o2, _ = E.reduce_factors([(view, ("a", "k"))], plate=("a",))            # a SUM over the huge stride
    t.testing.assert_close(o2.cpu().double(), vals.double().sum(0), rtol=2e-6, atol=2e-6)
    del big
    t.cuda.empty_cache()


@pytest.mark.gpu
def test_result_ring_c_abi_delivers_to_successive_slots_and_declines_other_shapes():
    """alan_reduce_desc_t.ring_*: a one-value, one-workgroup call writes through ring_slots[*ring_counter] and advances
    the counter modulo ring_n, leaving ``out`` alone; calls with several outputs, a plate stage or a two-launch plan
    return ALAN_ERR_UNSUPPORTED having enqueued nothing; malformed rings are ALAN_ERR_BAD_DESC."""
    g = t.Generator().manual_seed(3)
    a = t.randn(30, 30, generator=g).cuda()
    b = t.randn(30, generator=g).cuda()
    ring = E.ResultRing(a.device)
    sizes = {"i": 30, "j": 30}
    roles = {"i": N.REDUCE, "j": N.REDUCE}
    want = float(t.logsumexp((a.double() + b.double()[None, :]).reshape(-1), 0)) + 1.5
    out = t.full((), 7.0, device="cuda")
    for call in range(ring.n + 3):
        assert E._launch(N.MODE_LSE, [(a, ("i", "j")), (b, ("j",))], sizes, roles, out, (), add_const=1.5, ring=ring)
        t.cuda.synchronize()
        assert int(ring.counter) == (call + 1) % ring.n
        assert abs(float(ring.slots[call % ring.n]) - want) <= 2e-6 * abs(want)
        ring.slots[call % ring.n].zero_()
    assert float(out) == 7.0
    assert all(float(s) == 0.0 for s in ring.slots)
    # several outputs / a plate stage / a long reduce dim that is split into two launches: declined, nothing written
    keep = t.empty(30, device="cuda")
    assert not E._launch(N.MODE_LSE, [(a, ("i", "j"))], sizes, {"i": N.KEEP, "j": N.REDUCE}, keep, ("i",), ring=ring)
    assert not E._launch(N.MODE_LSE, [(a, ("i", "j"))], sizes, {"i": N.PLATE, "j": N.REDUCE}, out, (), ring=ring)
    long = t.randn(1 << 20, generator=g).cuda()
    assert not E._launch(N.MODE_LSE, [(long, ("n",))], {"n": 1 << 20}, {"n": N.REDUCE}, out, (), ring=ring)
    t.cuda.synchronize()
    assert int(ring.counter) == 3 and float(out) == 7.0 and all(float(s) == 0.0 for s in ring.slots)
    # SUM mode takes the ring too
    assert E._launch(N.MODE_SUM, [(a, ("i", "j"))], sizes, roles, out, (), ring=ring)
    t.testing.assert_close(ring.slots[3].cpu().double(), a.double().sum().cpu(), rtol=1e-5, atol=1e-4)
    # malformed: slots without a counter
    d = N.ReduceDesc()
    d.mode, d.ndim, d.n_factors = N.MODE_LSE, 1, 1
    d.size[0], d.role[0] = 30, N.REDUCE
    N.fill_tensor(d.factor[0], b, [1])
    N.fill_tensor(d.out, out, [0])
    d.ring_slots, d.ring_n = ring.table.data_ptr(), ring.n
    assert N.lib().alan_reduce(ctypes.byref(d), None, 0, None) == -1


# ------------------------------------------------------------------ oracle: pair contraction (pair.hip)
@pytest.mark.parametrize("shape", [
    dict(Y=2, B=3, G=100, A=100, Ky=100, plate=("B",)),          # bus_breakdown's Borough plate at K = 100
    dict(Y=1, B=2, G=137, A=33, Ky=150, plate=("B",)),           # ragged tiles, odd reduce length
    dict(Y=3, B=1, G=64, A=256, Ky=70, plate=()),                # no plate sum, the longest reduce dim taken
    dict(Y=2, B=5, G=48, A=40, Ky=48, plate=("B", "Y")),         # two plate dims, nothing kept besides the tile
], ids=["bus_K100", "ragged", "no_plate", "two_plates"])
def test_pair_contraction_against_the_oracle(shape, monkeypatch):
    """out[y, g, k] = sum_b LSE_a(L[y,b,g,a] + q[y,b,a] + P[y,b,k,a]): the output is bigger than every factor, the tile
    kernel of pair.hip takes it (>= 2^20 (output, a) pairs) -- against the oracle's logsumexp_sum + plate_sum, incl. -inf
    entries and a wholly -inf row (NaN, utils.py:219), and against the generic kernels (ALAN_PAIR=0 is read once per
    process, so those run in the same call only for sizes the tile kernel declines)."""
    g = t.Generator().manual_seed(7)
    Y, B, G, A, Ky = (shape[k] for k in ("Y", "B", "G", "A", "Ky"))
    L = t.randn(Y, B, G, A, generator=g) * 3
    q = t.randn(Y, B, A, generator=g)
    P = t.randn(Y, B, Ky, A, generator=g) * 3
    L[0, 0, 1, ::3] = float("-inf")
    P[0, 0, 2, :] = float("-inf")                                  # outputs [.., 2] of (y=0, b=0): every term -inf
    factors = [(L, ("Y", "B", "G", "A")), (q, ("Y", "B", "A")), (P, ("Y", "B", "Ky", "A"))]
    out, dims = E.reduce_factors(dev(factors), reduce=("A",), plate=shape["plate"])
    ref = orc.logsumexp_sum(("A",), *factors)
    for p in shape["plate"]:
        ref = orc.plate_sum(ref, p)
    same(out, dims, ref[0], ref[1], rtol=2e-5, atol=2e-5)
    assert t.isnan(out).any() and not t.isnan(out).all()
