"""Host logic that needs no GPU: recognition of model lambdas by symbolic tracing (dist._linear_form / _scaled_form /
_is_plain_exp) and the lazy PT kinds built from them."""
import torch as t

from alan_amd import dist as D
from alan_amd.dims import Dim, LinearPT, PT, ScaledPT

SCALE = 2.0


def test_linear_forms_are_recognised_and_everything_else_is_left_alone():
    assert D._linear_form(lambda z, x: z @ x) == (("dot", 0, 1),)
    assert D._linear_form(lambda z, x: t.matmul(z, x)) == (("dot", 0, 1),)
    assert D._linear_form(lambda alpha, phi, psi, run_type, bus_company_name:
                          (alpha + phi @ bus_company_name + psi @ run_type)) == (("arg", 0), ("dot", 1, 4), ("dot", 2, 3))
    for fn in (lambda z, x: (z * x).sum(-1), lambda z, x: z @ x + 1.0, lambda a, b: a + b, lambda z, x: (z @ x).exp(),
               lambda z, x: z @ x @ x, lambda z: z, lambda z, x: 2 * (z @ x)):
        assert D._linear_form(fn) is None
    bias = 3.0
    assert D._linear_form(lambda z, x: z @ x + bias) is None            # (a closure)


def test_scaled_forms_take_literal_constants_only():
    assert D._scaled_form(lambda prev: 0.9 * prev) == 0.9
    assert D._scaled_form(lambda v: v * 2) == 2.0
    assert D._scaled_form(lambda v: SCALE * v) is None                   # a module-level name could change under the cache
    k = 0.5
    assert D._scaled_form(lambda v: k * v) is None                       # a closure
    for fn in (lambda v: v * v, lambda v: 2 * v + 1, lambda v: v.exp(), lambda v: v / 2, lambda a, b: 2 * a):
        assert D._scaled_form(fn) is None
    assert D._is_plain_exp(lambda v: v.exp()) and not D._is_plain_exp(lambda v: (2 * v).exp())


def test_lazy_values_materialise_to_what_the_lambda_computes():
    d = Dim("K", 4)
    x = PT(t.randn(4, 3), (d,))
    s = ScaledPT(x.x, 0.9, x.dims)
    assert not s.materialised and s.n_pos == 1 and s.size_of(id(d)) == 4
    assert t.equal(s.x, x.x * 0.9) and s.materialised
    made = []
    lin = LinearPT([(x, x)], (d,), lambda: made.append(1) or (x.x * x.x).sum(-1))
    assert lin.n_pos == 0 and not lin.materialised and lin.size_of(id(d)) == 4
    assert t.allclose(lin.x, (x.x * x.x).sum(-1)) and lin.materialised
    lin.x
    assert made == [1]                                                   # evaluated once
    # on the CPU the recognisers never fire: call_model_lambda evaluates the lambda as written
    out = D.call_model_lambda(lambda prev: 0.9 * prev, [("prev", x)])
    assert type(out) is PT and t.allclose(out.x, 0.9 * x.x)
