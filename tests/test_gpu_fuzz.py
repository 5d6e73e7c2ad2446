"""Seeded differential test: random small contraction problems (dims, roles, which factor carries which dim,
memory layout, dtype, forward and backward) through the C ABI against the CPU oracle.  Complements the targeted
kernel tests: every dispatch decision of csrc/plan.hip (small-problem kernel, generic group kernel, block mode,
rows kernel, two-stage plate sum) gets exercised by some seed."""
import random

import pytest
import torch as t

from oracle import alan_oracle as orc
from alan_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ["a", "b", "c", "d", "e"]


def _layout(x, rng):
    """Same values, different strides: a permuted-storage view or a strided slice."""
    kind = rng.choice(["contig", "perm", "slice", "contig"])
    if kind == "perm" and x.ndim >= 2:
        perm = list(range(x.ndim))
        rng.shuffle(perm)
        inv = [perm.index(i) for i in range(x.ndim)]
        return x.permute(*perm).contiguous().permute(*inv)
    if kind == "slice" and x.ndim >= 1:
        big = t.zeros(*[2 * s for s in x.shape], dtype=x.dtype)
        idx = tuple(slice(0, 2 * s, 2) for s in x.shape)
        big[idx] = x
        return big[idx]
    return x


def _case(seed):
    rng = random.Random(seed)
    g = t.Generator().manual_seed(seed)
    nd = rng.randint(2, 5)
    dims = NAMES[:nd]
    sizes = {d: rng.choice([1, 2, 3, 5, 7]) for d in dims}
    sizes[rng.choice(dims)] = rng.choice([30, 64, 100, 257, 600])          # one long dim
    nf = rng.randint(1, 5)
    dtype = rng.choice([t.float32, t.float32, t.float32, t.float64])
    facs = []
    for i in range(nf):
        own = [d for d in dims if rng.random() < 0.6] or [rng.choice(dims)]
        if i == 0:
            own = list(dims)                                               # the union covers every dim
        rng.shuffle(own)
        x = (2.5 * t.randn(*[sizes[d] for d in own], generator=g)).to(dtype if rng.random() < 0.8 else t.float32)
        if rng.random() < 0.15:
            x[tuple(0 for _ in own)] = float("-inf")
        facs.append((_layout(x, rng), tuple(own)))
    n_red = rng.randint(0, max(0, nd - 1))
    reduce = tuple(rng.sample(dims, n_red))
    rest = [d for d in dims if d not in reduce]
    plate = tuple(rng.sample(rest, 1)) if (rest and rng.random() < 0.4) else ()
    return facs, reduce, plate


@pytest.mark.parametrize("seed", range(160))
def test_random_contraction_forward(seed):
    facs, reduce, plate = _case(seed)
    ref = orc.logsumexp_sum(reduce, *facs) if reduce else orc.broadcast_sum(facs)
    for p in plate:
        ref = orc.plate_sum(ref, p)
    out, dims = E.reduce_factors([(x.to(DEV), d) for x, d in facs], reduce=reduce, plate=plate)
    got = orc.align((out.cpu(), tuple(dims)), tuple(ref[1])) if dims else out.cpu()
    # mixed dtypes: torch (and the reference's sum(lps)) adds two fp32 factors in fp32 before promoting; the kernel
    # promotes every factor first -- fp32-level differences are expected there
    all64 = all(x.dtype == t.float64 for x, _ in facs)
    tol = dict(rtol=1e-10, atol=1e-10) if all64 else dict(rtol=3e-5, atol=3e-4)
    assert out.dtype == ref[0].dtype
    t.testing.assert_close(got.reshape(ref[0].shape), ref[0], equal_nan=True, **tol)


@pytest.mark.parametrize("seed", range(1000, 1060))
def test_random_contraction_backward(seed):
    facs, reduce, plate = _case(seed)
    facs = [(x.nan_to_num(neginf=-3.0), d) for x, d in facs]               # gradients at -inf are NaN by design
    if not reduce:
        pytest.skip("no log-sum-exp in this case")
    g = t.Generator().manual_seed(seed)
    cpu = [x.clone().requires_grad_(True) for x, _ in facs]
    ref = orc.logsumexp_sum(reduce, *[(x, d) for x, (_, d) in zip(cpu, facs)])
    for p in plate:
        ref = orc.plate_sum(ref, p)
    w = t.randn(ref[0].shape, generator=g, dtype=ref[0].dtype)
    gref = t.autograd.grad((ref[0] * w).sum(), cpu)
    dev = [x.detach().to(DEV).requires_grad_(True) for x, _ in facs]
    out, dims = E.reduce_factors([(x, d) for x, (_, d) in zip(dev, facs)], reduce=reduce, plate=plate)
    wd = orc.align((w, tuple(ref[1])), tuple(dims)).reshape(out.shape) if dims else w
    ggot = t.autograd.grad((out * wd.to(DEV)).sum(), dev)
    all64 = all(x.dtype == t.float64 for x, _ in facs)
    for a, b in zip(ggot, gref):
        tol = dict(rtol=1e-9, atol=1e-9) if all64 else dict(rtol=2e-4, atol=2e-4)
        t.testing.assert_close(a.cpu(), b, **tol)
