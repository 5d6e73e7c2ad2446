"""Targeted parity tests of the LDS-staged rows kernel (csrc/rows.hip) against the CPU oracle: every
template variant (lanes-per-row G, 8-byte vs rotated LDS reads, shared / general small factors),
fused and unfused plate sum, saved LSE values + backward, SUM mode, unaligned slab starts and the
16-byte-load tail guard.  Sizes are chosen above the kernel's 16384-element threshold."""
import math

import pytest
import torch as t

from oracle import alan_oracle as orc
from alan_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cmp(out, dims, ref, ref_dims, rtol=3e-5, atol=3e-4):
    a = orc.align((out.detach().cpu(), tuple(dims)), tuple(ref_dims)) if dims else out.detach().cpu()
    t.testing.assert_close(a.reshape(ref.shape), ref, rtol=rtol, atol=atol, equal_nan=True)


# L -> (G, read path): <=32 G=1, <=64 G=2, <=128 G=4, <=256 G=8; gcd(L,64)>=16 -> rotated b32; even -> b64; odd -> b32
@pytest.mark.parametrize("L", [8, 9, 16, 30, 31, 32, 33, 48, 50, 64, 100, 127, 128, 200, 256])
@pytest.mark.parametrize("plate", [False, True])
def test_lse_over_contiguous_dim(L, plate):
    g = t.Generator().manual_seed(L * 2 + plate)
    M, A = 7, max(3, 20000 // (7 * L) + 1)
    F = t.randn(M, A, L, generator=g) * 4
    shared = t.randn(M, L, generator=g)            # constant over the window (no inner keep dim)
    facs = [(F, ("m", "a", "k")), (shared, ("m", "k"))]
    ref = orc.logsumexp_sum(("k",), *facs)
    if plate:
        ref = orc.plate_sum(ref, "m")
    out, dims = E.reduce_factors([(x.to(DEV), d) for x, d in facs], reduce=("k",), plate=("m",) if plate else ())
    _cmp(out, dims, ref[0], ref[1])


@pytest.mark.parametrize("L,n_shared", [(30, 2), (30, 4), (100, 3), (32, 5), (9, 3)])
def test_several_window_constant_small_factors(L, n_shared):
    """movielens' plate step carries two of them (-(log Q + log K) and the data likelihood); the kernel keeps the
    first two as pending loads and adds any further ones in a rolled loop."""
    g = t.Generator().manual_seed(L * 7 + n_shared)
    M, A = 6, max(3, 20000 // (6 * L) + 1)
    F = t.randn(M, A, L, generator=g) * 4
    facs = [(F, ("m", "a", "k"))]
    for i in range(n_shared):
        facs.append((t.randn(M, L, generator=g), ("m", "k")) if i % 2 == 0 else (t.randn(L, generator=g), ("k",)))
    ref = orc.plate_sum(orc.logsumexp_sum(("k",), *facs), "m")
    out, dims = E.reduce_factors([(x.to(DEV), d) for x, d in facs], reduce=("k",), plate=("m",))
    _cmp(out, dims, ref[0], ref[1])


@pytest.mark.parametrize("L", [30, 32, 100])
def test_general_small_factor_depends_on_inner_keep_dim(L):
    g = t.Generator().manual_seed(L)
    M, A = 5, 20000 // (5 * L) + 2
    F = t.randn(M, A, L, generator=g) * 3
    gen = t.randn(A, L, generator=g)                # varies with the row inside a window
    sh = t.randn(L, generator=g)
    facs = [(F, ("m", "a", "k")), (gen, ("a", "k")), (sh, ("k",))]
    ref = orc.plate_sum(orc.logsumexp_sum(("k",), *facs), "m")
    out, dims = E.reduce_factors([(x.to(DEV), d) for x, d in facs], reduce=("k",), plate=("m",))
    _cmp(out, dims, ref[0], ref[1])


@pytest.mark.parametrize("L,off", [(30, 1), (30, 2), (30, 3), (9, 1), (100, 1), (33, 2)])
def test_unaligned_views_and_tail(L, off):
    """Factor is a view starting `off` floats into its storage: 16-byte alignment is broken -> the
    library must fall back (or handle it); the last slab ends exactly at the end of the storage."""
    g = t.Generator().manual_seed(L + off)
    M, A = 6, 20000 // (6 * L) + 1
    store = t.randn(off + M * A * L, generator=g).to(DEV)
    F = store[off:].view(M, A, L)
    ref = orc.plate_sum(orc.logsumexp_dims((F.cpu(), ("m", "a", "k")), ("k",)), "m")
    out, dims = E.reduce_factors([(F, ("m", "a", "k"))], reduce=("k",), plate=("m",))
    _cmp(out, dims, ref[0], ref[1])
    # a plate slice (what Split produces): starts at a multiple of the plate stride
    Fs = F[2:5]
    ref = orc.plate_sum(orc.logsumexp_dims((Fs.cpu(), ("m", "a", "k")), ("k",)), "m")
    out, dims = E.reduce_factors([(Fs, ("m", "a", "k"))], reduce=("k",), plate=("m",))
    _cmp(out, dims, ref[0], ref[1])


@pytest.mark.parametrize("L", [30, 64, 150])
def test_plate_sum_mode_rows(L):
    """ALAN_MODE_SUM over a contiguous dim (bus_breakdown's sum over plate_ID)."""
    g = t.Generator().manual_seed(L)
    A = 20000 // L + 5
    X = t.randn(A, L, generator=g)
    y = t.randn(A, generator=g)
    out, dims = E.reduce_factors([(X.to(DEV), ("a", "i")), (y.to(DEV), ("a",))], plate=("i",))
    _cmp(out, dims, (X + y[:, None]).sum(1), ("a",), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("L,K2", [(30, 30), (100, 12), (16, 40)])
def test_saved_lse_and_backward_through_fused_plate(L, K2):
    g = t.Generator().manual_seed(L + K2)
    M = 9
    F = t.randn(M, K2, K2, L, generator=g) * 2
    gz = t.randn(M, L, generator=g)
    cpu = [(F, ("m", "a", "b", "z")), (gz, ("m", "z"))]
    (ref, ref_dims), ref_grads = orc.reduce_Ks_grads(cpu, ("z",), plate="m")
    dev = [(x.to(DEV).requires_grad_(True), d) for x, d in cpu]
    out, dims = E.reduce_factors(dev, reduce=("z",), plate=("m",))
    _cmp(out, dims, ref, ref_dims)
    w = t.randn(out.shape, generator=g)
    wref = orc.align((w, tuple(dims)), tuple(ref_dims)).reshape(ref.shape)
    (_, _), ref_grads = orc.reduce_Ks_grads(cpu, ("z",), plate="m", grad_out=wref)
    grads = t.autograd.grad(out, [x for x, _ in dev], w.to(DEV))
    for a, b in zip(grads, ref_grads):
        t.testing.assert_close(a.cpu(), b, rtol=2e-4, atol=2e-5)


def test_neg_inf_and_nan_rows():
    L, A = 30, 800
    x = t.randn(A, L)
    x[0, :] = float("-inf")
    x[1, 5] = float("-inf")
    x[2, 0] = float("nan")
    x[3, :] = float("inf")
    ref, _ = orc.logsumexp_dims((x, ("a", "k")), ("k",))
    out, dims = E.reduce_factors([(x.to(DEV), ("a", "k"))], reduce=("k",))
    got = out.cpu()
    assert t.isnan(got[0]) and t.isnan(ref[0])
    assert t.isnan(got[2]) and t.isnan(ref[2])
    assert t.isnan(got[3]) and t.isnan(ref[3])
    t.testing.assert_close(got[[1] + list(range(4, A))], ref[[1] + list(range(4, A))], rtol=3e-5, atol=3e-5)


# ------------------------------------------------------------------ BASELINE.json full sizes
def _sml(M, K, seed):
    g = t.Generator().manual_seed(seed)
    F = -0.5 * t.randn(M, K, K, K, generator=g) ** 2 - 0.9189 - math.log(K)
    gz = -0.5 * t.randn(M, K, generator=g) ** 2 - 0.9189 - math.log(K)
    return F, gz


@pytest.mark.parametrize("M,K", [(300, 30), (38, 100)], ids=["C2-movielens-K30-M300", "C4-chunk-K100-M38"])
def test_full_size_against_oracle_and_properties(M, K):
    """The plate step at the sizes BASELINE.json names (C2: F[300,30,30,30] = 32 MB; one C4 chunk:
    F[38,100,100,100] = 152 MB): direct comparison with the CPU oracle, plus size-independent
    properties -- Split chunks sum to the whole plate (logpq.py:151-153), a constant added to a factor
    shifts the result by M*c, permuting the plate leaves the result unchanged, LSE >= max."""
    F, gz = _sml(M, K, 7)
    dims = (("m", "a", "b", "z"), ("m", "z"))
    ref = orc.plate_sum(orc.logsumexp_sum(("z",), (F, dims[0]), (gz, dims[1])), "m")
    Fd, gd = F.to(DEV), gz.to(DEV)
    out, od = E.reduce_factors([(Fd, dims[0]), (gd, dims[1])], reduce=("z",), plate=("m",))
    _cmp(out, od, ref[0], ref[1], rtol=2e-5, atol=2e-3)
    # chunks of the plate sum to the whole
    acc, start = None, 0
    for n in orc.split_sizes(M, max(2, M // 8 + 1)):
        part, pd = E.reduce_factors([(Fd[start:start + n], dims[0]), (gd[start:start + n], dims[1])],
                                    reduce=("z",), plate=("m",))
        acc = part if acc is None else acc + part
        start += n
    t.testing.assert_close(acc, out, rtol=1e-5, atol=2e-3)
    # shift by a constant
    shifted, _ = E.reduce_factors([(Fd, dims[0]), (gd + 0.25, dims[1])], reduce=("z",), plate=("m",))
    t.testing.assert_close(shifted, out + 0.25 * M, rtol=1e-5, atol=2e-3)
    # plate permutation invariance
    perm = t.randperm(M, generator=t.Generator().manual_seed(1)).to(DEV)
    permuted, _ = E.reduce_factors([(Fd[perm].contiguous(), dims[0]), (gd[perm].contiguous(), dims[1])],
                                   reduce=("z",), plate=("m",))
    t.testing.assert_close(permuted, out, rtol=1e-5, atol=2e-3)
    # per-row bound: LSE over z >= max over z  (unfused call)
    rows, rd = E.reduce_factors([(Fd, dims[0]), (gd, dims[1])], reduce=("z",))
    mx = (Fd + gd[:, None, None, :]).amax(-1)
    assert bool((orc.align((rows.cpu(), rd), ("m", "a", "b")) >= mx.cpu() - 1e-5).all())


@pytest.mark.parametrize("L,K2,plate", [(30, 30, True), (100, 12, True), (16, 40, True), (9, 50, True), (31, 20, True),
                                        (30, 30, False), (64, 20, False)])
def test_one_pass_backward_matches_per_factor_backward(L, K2, plate, monkeypatch):
    """alan_reduce_backward (one pass over F: grad F streamed out, the small factors' gradients from the slab's column
    sums) against the generic route (one WEXPSUM launch per factor) and against torch autograd on the CPU."""
    from alan_amd import native as N
    g = t.Generator().manual_seed(L * 3 + K2)
    M = 9
    shapes = [((M, K2, K2, L), ("m", "a", "b", "k")), ((M, L), ("m", "k")), ((L,), ("k",)), ((M,), ("m",))]
    raw = [2 * t.randn(s, generator=g) for s, _ in shapes]
    dims = [d for _, d in shapes]
    pl = ("m",) if plate else ()

    def run(fused):
        calls = []
        real = N.run_reduce_backward
        monkeypatch.setattr(N, "run_reduce_backward",
                            (lambda d, dev: calls.append(1) or real(d, dev)) if fused else (lambda d, dev: False))
        leaves = [x.clone().to(DEV).requires_grad_(True) for x in raw]
        out, odims = E.reduce_factors(list(zip(leaves, dims)), reduce=("k",), plate=pl)
        w = t.randn(out.shape, generator=t.Generator().manual_seed(1)).to(DEV)
        grads = t.autograd.grad((out * w).sum(), leaves)
        monkeypatch.setattr(N, "run_reduce_backward", real)
        return out, odims, w, grads, len(calls)

    out_f, od, w, gf, n_f = run(True)
    out_g, _, _, gg, n_g = run(False)
    assert n_f == 1 and n_g == 0                                  # the one-pass path was taken (and completed)
    for a, b in zip(gf, gg):
        t.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)
    cpu = [x.clone().requires_grad_(True) for x in raw]
    ref = orc.logsumexp_sum(("k",), *zip(cpu, dims))
    if plate:
        ref = orc.plate_sum(ref, "m")
    wc = orc.align((w.cpu(), tuple(od)), tuple(ref[1])).reshape(ref[0].shape) if od else w.cpu()
    gref = t.autograd.grad((ref[0] * wc).sum(), cpu)
    for a, b in zip(gf, gref):
        t.testing.assert_close(a.cpu(), b, rtol=2e-4, atol=2e-4)


def test_one_pass_backward_declines_row_dependent_small_factors():
    """A small factor that varies over the window's rows is not the one-pass kernel's shape: the library says so
    (ALAN_ERR_UNSUPPORTED) and autograd takes the per-factor route -- same gradients."""
    g = t.Generator().manual_seed(3)
    M, A, L = 6, 120, 30
    raw = [t.randn(M, A, L, generator=g), t.randn(A, L, generator=g)]
    dims = [("m", "a", "k"), ("a", "k")]
    leaves = [x.clone().to(DEV).requires_grad_(True) for x in raw]
    out, od = E.reduce_factors(list(zip(leaves, dims)), reduce=("k",), plate=("m",))
    grads = t.autograd.grad(out.sum(), leaves)
    cpu = [x.clone().requires_grad_(True) for x in raw]
    ref = orc.plate_sum(orc.logsumexp_sum(("k",), *zip(cpu, dims)), "m")
    gref = t.autograd.grad(ref[0].sum(), cpu)
    for a, b in zip(grads, gref):
        t.testing.assert_close(a.cpu(), b, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("plate", [False, True])
@pytest.mark.parametrize("L", [30, 100])
def test_fp64_small_factor_keeps_the_streaming_path(L, plate):
    """A real data set's fp64 observations make the likelihood factor fp64 while the big factor stays fp32 (movielens:
    F[plate_1,K,K,K] fp32 + obs term [plate_1,K_z] fp64).  The result has the promoted dtype the reference would
    return; the reduction itself runs in fp32, so it agrees with the fp64 oracle to fp32 rounding."""
    g = t.Generator().manual_seed(L + plate)
    M, A = 7, max(3, 20000 // (7 * L) + 1)
    F = t.randn(M, A, L, generator=g) * 4
    sh64 = t.randn(M, L, generator=g, dtype=t.float64)
    sh32 = t.randn(L, generator=g)
    facs = [(F, ("m", "a", "k")), (sh64, ("m", "k")), (sh32, ("k",))]
    ref = orc.logsumexp_sum(("k",), *facs)
    if plate:
        ref = orc.plate_sum(ref, "m")
    out, dims = E.reduce_factors([(x.to(DEV), d) for x, d in facs], reduce=("k",), plate=("m",) if plate else ())
    assert out.dtype == t.float64 == ref[0].dtype
    _cmp(out, dims, ref[0], ref[1], rtol=3e-5, atol=3e-4)
    # gradients through the mixed-dtype step (per-factor route; the one-pass backward is fp32 only)
    leaves = [x.clone().to(DEV).requires_grad_(True) for x, _ in facs]
    o2, _ = E.reduce_factors([(x, d) for x, (_, d) in zip(leaves, facs)], reduce=("k",), plate=("m",) if plate else ())
    grads = t.autograd.grad(o2.sum(), leaves)
    cpu = [x.clone().requires_grad_(True) for x, _ in facs]
    r2 = orc.logsumexp_sum(("k",), *[(x, d) for x, (_, d) in zip(cpu, facs)])
    if plate:
        r2 = orc.plate_sum(r2, "m")
    gref = t.autograd.grad(r2[0].sum(), cpu)
    for a, b in zip(grads, gref):
        assert a.dtype == b.dtype
        t.testing.assert_close(a.cpu(), b, rtol=2e-4, atol=2e-4)
