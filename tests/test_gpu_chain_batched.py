"""Batched timeseries chains (alan_chain_logmmexp_batched and its backward): a timeseries plate nested under other
plates / parent K dims.  The reference reaches this through torchdim batch dims left on ``lp`` after
``lp.order(T, K_init, K_curr)`` (logpq.py:133-139), i.e. one independent chain per batch element -- so the checker
is the oracle's chain (pinned on tests/golden/chain.pt) applied to each element."""
import pytest
import torch as t

from alan_amd import native as N
from alan_amd.contract import chain_logmmexp_lse, chain_logmmexp
from oracle import alan_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ms(B, T, K, dtype, seed):
    g = t.Generator().manual_seed(seed)
    # log-density-like entries with a per-step offset, so normalisation matters
    return (-0.5 * t.randn(B, T, K, K, generator=g, dtype=t.float64) ** 2 - 0.9
            + t.randn(B, T, 1, 1, generator=g, dtype=t.float64)).to(dtype)


def _oracle(ms):
    chain = t.stack([orc.chain_logmmexp(m) for m in ms.double()], 0)
    return t.logsumexp(chain, -1), chain


# (K in 33..100, fp32: the matrix-core pair kernel, 2x2 and 4x4 tiles with ragged edges; an odd leftover per round)
SHAPES = [(1, 1, 3), (3, 2, 3), (5, 7, 3), (4, 50, 30), (700, 9, 10), (2, 33, 40), (3, 12, 100), (2500, 4, 30),
          (2, 9, 33), (1, 17, 64), (2, 5, 65), (1, 1, 70), (3, 3, 97)]


@pytest.mark.parametrize("B,T,K", SHAPES, ids=[f"B{b}_T{T}_K{k}" for b, T, k in SHAPES])
@pytest.mark.parametrize("dtype", [t.float32, t.float64], ids=["f32", "f64"])
def test_batched_chain_matches_the_oracle_per_element(B, T, K, dtype):
    ms = _ms(B, T, K, dtype, 7 * B + T + K)
    want_vec, want_chain = _oracle(ms)
    if dtype == t.float64 and K > 96:          # two fp64 [K,K] operands no longer fit LDS: the host takes the scan
        with pytest.raises(N.NativeError):
            N.chain_logmmexp(ms.to(DEV))
        t.testing.assert_close(chain_logmmexp_lse(ms.to(DEV)).cpu(), want_vec, rtol=1e-11, atol=1e-10)
        return
    vec, chain, _ = N.chain_logmmexp(ms.to(DEV), want_chain=True)
    kw = dict(rtol=2e-5, atol=2e-5 * (1 + T ** 0.5)) if dtype == t.float32 else dict(rtol=1e-11, atol=1e-10)
    t.testing.assert_close(vec.cpu().double(), want_vec, **kw)
    t.testing.assert_close(chain.cpu().double(), want_chain, **kw)
    # the batch is the unbatched kernel run B times: bit-identical per element
    if B <= 5:
        for b in range(B):
            v1, _, _ = N.chain_logmmexp(ms[b].to(DEV))
            assert t.equal(v1, vec[b])


def test_batched_chain_takes_strided_batches():
    """[T, K, B, K]-ordered storage viewed as [B, T, K, K]: strides travel through the C ABI, no copy."""
    B, T, K = 6, 11, 30
    ms = _ms(B, T, K, t.float32, 3)
    store = ms.permute(1, 2, 0, 3).contiguous().to(DEV)          # [T, K, B, K]
    view = store.permute(2, 0, 1, 3)
    assert not view.is_contiguous()
    vec, _, _ = N.chain_logmmexp(view)
    want, _ = _oracle(ms)
    t.testing.assert_close(vec.cpu().double(), want, rtol=2e-5, atol=1e-4)
    exp = store[:, :, :1, :].expand(T, K, B, K).permute(2, 0, 1, 3)     # stride-0 batch: B copies of one chain
    vec2, _, _ = N.chain_logmmexp(exp)
    assert t.equal(vec2[0], vec2[B - 1])


@pytest.mark.parametrize("B,T,K", [(1, 5, 3), (4, 9, 10), (300, 4, 30), (3, 40, 64)])
@pytest.mark.parametrize("dtype", [t.float32, t.float64], ids=["f32", "f64"])
def test_batched_chain_backward_matches_autograd_of_the_oracle(B, T, K, dtype):
    ms = _ms(B, T, K, dtype, 11 + B)
    g = t.randn(B, K, generator=t.Generator().manual_seed(5), dtype=t.float64)
    x = ms.double().requires_grad_(True)
    out = t.stack([orc.timeseries_plate(m) for m in x], 0)
    (want,) = t.autograd.grad(out, x, g)
    y = ms.to(DEV).requires_grad_(True)
    got_out = chain_logmmexp_lse(y)
    (got,) = t.autograd.grad(got_out, y, g.to(DEV, dtype))
    kw = dict(rtol=2e-3, atol=2e-5) if dtype == t.float32 else dict(rtol=1e-8, atol=1e-10)
    t.testing.assert_close(got.cpu().double(), want, **kw)


def test_large_K_batch_goes_through_the_matvec_scan():
    """K > 100 (the reference's ground-truth tests use K = 1000): the right-to-left log-matvec scan, batched."""
    B, T, K = 3, 4, 130
    ms = _ms(B, T, K, t.float32, 9)
    got = chain_logmmexp_lse(ms.to(DEV))
    want, _ = _oracle(ms)
    t.testing.assert_close(got.cpu().double(), want, rtol=2e-5, atol=1e-4)
    assert chain_logmmexp(ms[:, :, :3, :3].to(DEV)).shape == (B, 3, 3)


@pytest.mark.parametrize("B,T,K,dtype", [(1, 7, 5, t.float32), (3, 40, 30, t.float32), (2, 9, 40, t.float64), (1, 1, 3, t.float32)])
def test_chain_of_terms_adds_the_factors_on_load(B, T, K, dtype):
    """alan_chain_logmmexp_terms: the chain of a SUM of up to three factors (a timeseries plate's transition factor
    [T,K_init,K], its -(log Q + log K) [T,K] and a likelihood [T,K], the latter two without the K_init dim: stride 0)
    against adding them first."""
    g = t.Generator().manual_seed(B + T + K)
    a = _ms(B, T, K, dtype, 5).to(DEV)
    b = t.randn(B, T, 1, K, generator=g, dtype=t.float64).to(dtype).to(DEV)
    c = t.randn(1, T, 1, K, generator=g, dtype=t.float64).to(dtype).to(DEV)
    shape = (B, T, K, K)
    for terms in ([a], [a, b.expand(shape)], [a, b.expand(shape), c.expand(shape)]):
        got = N.chain_logmmexp_terms(terms)
        want, _ = _oracle(sum(x.cpu().double() for x in terms))
        kw = dict(rtol=2e-5, atol=2e-5 * (1 + T ** 0.5)) if dtype == t.float32 else dict(rtol=1e-11, atol=1e-10)
        t.testing.assert_close(got.cpu().double(), want, **kw)
    one, _, _ = N.chain_logmmexp(a)
    assert t.equal(N.chain_logmmexp_terms([a]), one)                       # one term = the plain entry point


def test_chain_terms_are_added_on_load_by_the_matrix_core_kernel_too():
    """alan_chain_logmmexp_terms at K = 50 (2x2 tiles): the plate's factors, one with a broadcast K_init dim, are summed
    by the first round as it loads."""
    B, T, K = 2, 21, 50
    a, b3 = _ms(B, T, K, t.float32, 1), _ms(B, T, K, t.float32, 2)[:, :, :1, :]
    want, _ = _oracle(a + b3)
    vec = N.chain_logmmexp_terms([a.to(DEV), b3.to(DEV).expand(B, T, K, K)])
    t.testing.assert_close(vec.cpu().double(), want, rtol=2e-5, atol=2e-4)
