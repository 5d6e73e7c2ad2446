"""Pins the CPU oracle (oracle/alan_oracle.py) to golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import math

import pytest
import torch as t

from oracle import alan_oracle as orc
from conftest import load_golden


def _same(a, a_names, b, b_names, rtol=1e-6, atol=1e-6):
    assert set(a_names) == set(b_names)
    a = orc.align((a, tuple(a_names)), tuple(b_names)) if a_names else a
    t.testing.assert_close(a.reshape(b.shape), b, rtol=rtol, atol=atol, equal_nan=True)


@pytest.mark.parametrize("case", load_golden("lse_dims.pt"), ids=lambda c: c["name"])
def test_logsumexp_dims(case):
    out, names = orc.logsumexp_dims((case["x"], case["names"]), case["reduce"])
    # same op sequence as the reference => bitwise on CPU
    _same(out, names, case["out"], case["out_names"], rtol=0, atol=0)
    if "mean_out" in case:
        out, names = orc.logmeanexp_dims((case["x"], case["names"]), case["reduce"])
        _same(out, names, case["mean_out"], case["mean_out_names"], rtol=0, atol=0)


def test_logsumexp_dims_errors():
    x = (t.zeros(2, 3), ("a", "b"))
    with pytest.raises(Exception):
        orc.logsumexp_dims(x, ("c",))
    with pytest.raises(Exception):
        orc.logsumexp_dims(x, ("a", "a"))
    out, names = orc.logsumexp_dims(x, ("c",), ignore_extra_dims=True)
    assert names == ("a", "b")


@pytest.mark.parametrize("case", load_golden("seam_synthetic.pt"), ids=lambda c: c["name"])
def test_reduce_Ks_synthetic(case):
    out, names = orc.reduce_Ks(case["factors"], case["Ks"])
    assert out.dtype == case["out"].dtype
    _same(out, names, case["out"], case["out_names"], rtol=2e-6, atol=2e-6)
    # order-free fp64 brute force (independent of any planner)
    bf, bn = orc.reduce_Ks_bruteforce(case["factors"], case["Ks"])
    _same(bf, bn, case["brute_f64"], case["brute_names"], rtol=1e-12, atol=1e-12)
    if not case["name"].startswith("all_neg_inf"):
        _same(out.double(), names, case["brute_f64"], case["brute_names"], rtol=1e-5, atol=1e-5)
    if "grads_weighted" in case:
        go = orc.align((case["grad_out"], case["out_names"]), names) if names else case["grad_out"]
        _, grads = orc.reduce_Ks_grads(case["factors"], case["Ks"], grad_out=go.reshape(out.shape))
        for g, ref in zip(grads, case["grads_weighted"]):
            t.testing.assert_close(g, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("case", load_golden("seam_recorded.pt"),
                         ids=lambda c: f"{c['model']}-K{c['K']}-{c['call']}")
def test_reduce_Ks_recorded(case):
    out, names = orc.reduce_Ks(case["factors"], case["Ks"])
    assert out.dtype == case["out"].dtype
    _same(out, names, case["out"], case["out_names"], rtol=1e-6, atol=1e-5)


def test_split_sizes():
    assert orc.split_sizes(10, 4) == [4, 4, 2]        # tests/linear_gaussian.py:61
    assert orc.split_sizes(10, 3) == [3, 3, 2, 2]     # Split.py:92-95 borrow
    assert orc.split_sizes(300, 38) == [38] * 7 + [34]
    assert orc.split_sizes(5, 2) == [2, 2, 1]
    with pytest.raises(AssertionError):
        orc.split_sizes(4, 4)


def _chain_input(case):
    if case["ms"] is not None:
        return case["ms"]
    g = t.Generator().manual_seed(case["seed"])
    dtype = getattr(t, case["dtype"].split(".")[-1])
    T, K = case["T"], case["K"]
    return -0.5 * t.randn(T, K, K, generator=g, dtype=dtype) ** 2 - 0.9189 - math.log(K)


@pytest.mark.parametrize("case", [c for c in load_golden("chain.pt") if "T" in c],
                         ids=lambda c: f"T{c['T']}K{c['K']}")
def test_chain_logmmexp(case):
    ms = _chain_input(case)
    kk = orc.chain_logmmexp(ms)
    t.testing.assert_close(kk, case["chain"], rtol=1e-6, atol=1e-5)
    t.testing.assert_close(orc.timeseries_plate(ms), case["out"], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("case", load_golden("chain_peaked.pt"), ids=lambda c: f"B{c['B']}T{c['T']}K{c['K']}")
def test_chain_on_the_eps_floor(case):
    """Peaked transition matrices: a quarter or more of the result sits on the +eps floor (utils.py:506); the oracle
    follows the reference's bracketing there and its autograd is the reference's, floor and amax paths included."""
    x = case["ms"].clone().requires_grad_(True)
    chains = t.stack([orc.chain_logmmexp(m) for m in x], 0)
    out = t.logsumexp(chains, -1)
    t.testing.assert_close(chains, case["chain"], rtol=1e-6, atol=1e-5)
    t.testing.assert_close(out, case["out"], rtol=1e-6, atol=1e-5)
    (g,) = t.autograd.grad((out * case["grad_out"]).sum(), x, retain_graph=True)
    t.testing.assert_close(g, case["grad"], rtol=1e-5, atol=1e-7)
    (gc,) = t.autograd.grad((chains * case["grad_chain_out"]).sum(), x)
    t.testing.assert_close(gc, case["grad_chain"], rtol=1e-5, atol=1e-6)


def test_logmmexp():
    (case,) = [c for c in load_golden("chain.pt") if c.get("name") == "logmmexp"]
    t.testing.assert_close(orc.logmmexp(case["prev"], case["curr"]), case["out"], rtol=0, atol=0)


def test_plate_sum_and_split_equivalence():
    g = t.Generator().manual_seed(5)
    F = (t.randn(10, 3, 4, generator=g), ("T", "Ka", "Kz"))
    gz = (t.randn(10, 4, generator=g), ("T", "Kz"))
    full = orc.plate_sum(orc.reduce_Ks([F, gz], ("Kz",)), "T")
    acc, start = None, 0
    for n in orc.split_sizes(10, 3):
        sl = lambda f: (f[0][start:start + n], f[1])
        acc = orc.plate_sum(orc.reduce_Ks([sl(F), sl(gz)], ("Kz",)), "T", prev=acc)
        start += n
    _same(acc[0], acc[1], full[0], full[1], rtol=1e-5, atol=1e-5)
