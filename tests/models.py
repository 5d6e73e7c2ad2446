"""The reference's test / example models restated in the alan_amd API (model structure follows
tests/linear_gaussian.py, tests/linear_gaussian_latents.py, tests/model1.py,
examples/models/movielens/movielens.py:39-82 and examples/models/bus_breakdown/bus_breakdown.py:38-100
of the reference), with data / inputs / parameter values taken from the golden fixtures."""
import torch as t

import alan_amd as alan
from alan_amd import Normal, Bernoulli, Plate, BoundPlate, Group, Problem, Data, OptParam, QEMParam, Timeseries
from alan_amd.dims import Dim


def _named(pair):
    x, names = pair
    return x.clone().refine_names(*names) if x.ndim else x.clone()


def linear_gaussian(fx):
    P = Plate(a=Normal(2, 2), T=Plate(d=Normal(lambda a: 2.5 * a, 3)))
    Q = Plate(a=Normal(1, 4), T=Plate(d=Data()))
    sizes = fx["platesizes"]
    return Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes), {"d": _named(fx["data"]["d"])})


def linear_gaussian_latents(fx):
    P = Plate(a=Normal(2, 2), T=Plate(z=Normal("a", 1.3), d=Normal("z", 1.5)))
    Q = Plate(a=Normal(1, 4), T=Plate(z=Normal(lambda a: 1.5 * a, 3.5), d=Data()))
    sizes = fx["platesizes"]
    return Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes), {"d": _named(fx["data"]["d"])})


def model1(fx):
    P = Plate(
        a=Normal(0, 1), b=Normal("a", 1), c=Normal(0, lambda a: a.exp()),
        p1=Plate(d=Normal("a", 1), p2=Plate(e=Normal("d", 1.))),
    )
    qp = fx["Q_inputs_params"]
    Q = Plate(
        ab=Group(a=Normal(QEMParam(qp["a_loc"][0]), QEMParam(qp["a_scale"][0])), b=Normal("a", 1)),
        c=Normal(0, lambda a: a.exp()),
        p1=Plate(d=Normal(OptParam(0.), "d_scale"), p2=Plate(e=Data())),
    )
    sizes = fx["platesizes"]
    Pb = BoundPlate(P, sizes)
    Qb = BoundPlate(Q, sizes, extra_opt_params={"d_scale": _named(qp["d_scale"])})
    prob = Problem(Pb, Qb, {"e": _named(fx["data"]["e"])})
    _load_opt(Qb, qp)
    return prob


def _load_opt(bound, values):
    """Overwrite OptParam raw values so that transformed values equal the fixture's."""
    with t.no_grad():
        store = bound._opt_params
        for k in store._keys:
            raw = getattr(store, f"t_{k}")
            target = values[k][0].to(raw.dtype)
            trans = bound.opt_paramname2trans[k]
            probe = trans(t.zeros(()))
            if float(probe) == 1.0:          # exp transformation
                target = target.log()
            raw.copy_(target.reshape(raw.shape))


d_z = 18


def movielens(fx=None, sizes=None, x=None, obs=None):
    if fx is not None:
        sizes = fx["platesizes"]
        x = _named(fx["P_inputs_params"]["x"])
        obs = _named(fx["data"]["obs"])
    P = Plate(
        mu_z=Normal(t.zeros((d_z,)), t.ones((d_z,))),
        psi_z=Normal(t.zeros((d_z,)), t.ones((d_z,))),
        plate_1=Plate(
            z=Normal("mu_z", lambda psi_z: psi_z.exp()),
            plate_2=Plate(obs=Bernoulli(logits=lambda z, x: z @ x)),
        ),
    )
    Q = Plate(
        mu_z=Normal(OptParam(t.zeros((d_z,))), OptParam(t.zeros((d_z,)), transformation=t.exp)),
        psi_z=Normal(OptParam(t.zeros((d_z,))), OptParam(t.zeros((d_z,)), transformation=t.exp)),
        plate_1=Plate(
            z=Normal(OptParam(t.zeros((d_z,))), OptParam(t.zeros((d_z,)), transformation=t.exp)),
            plate_2=Plate(obs=Data()),
        ),
    )
    Pb = BoundPlate(P, sizes, inputs={"x": x})
    Qb = BoundPlate(Q, sizes, inputs={"x": x})
    prob = Problem(Pb, Qb, {"obs": obs})
    if fx is not None:
        _load_opt(Qb, fx["Q_inputs_params"])
    return prob


def bus_breakdown(fx):
    sizes = fx["platesizes"]
    inp = {k: _named(fx["P_inputs_params"][k]) for k in ("run_type", "bus_company_name")}
    nb, nr = inp["bus_company_name"].shape[-1], inp["run_type"].shape[-1]
    P = Plate(
        psi=Normal(t.zeros((nr,)), t.ones((nr,))),
        phi=Normal(t.zeros((nb,)), t.ones((nb,))),
        sigma_beta=Normal(0, 1),
        mu_beta=Normal(0, 1),
        plate_Year=Plate(
            beta=Normal("mu_beta", lambda sigma_beta: sigma_beta.exp()),
            sigma_alpha=Normal(0, 1),
            plate_Borough=Plate(
                alpha=Normal("beta", lambda sigma_alpha: sigma_alpha.exp()),
                plate_ID=Plate(
                    obs=Bernoulli(logits=lambda alpha, phi, psi, run_type, bus_company_name:
                                  (alpha + phi @ bus_company_name + psi @ run_type)),
                ),
            ),
        ),
    )
    Q = Plate(
        global_latents=Group(
            psi=Normal(OptParam(t.zeros(nr)), OptParam(t.zeros(nr), transformation=t.exp)),
            phi=Normal(OptParam(t.zeros(nb)), OptParam(t.zeros(nb), transformation=t.exp)),
            sigma_beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
            mu_beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
        ),
        plate_Year=Plate(
            year_latents=Group(
                beta=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
                sigma_alpha=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
            ),
            plate_Borough=Plate(
                alpha=Normal(OptParam(0.), OptParam(0., transformation=t.exp)),
                plate_ID=Plate(obs=Data()),
            ),
        ),
    )
    Pb = BoundPlate(P, sizes, inputs=inp)
    Qb = BoundPlate(Q, sizes, inputs=inp)
    prob = Problem(Pb, Qb, {"obs": _named(fx["data"]["obs"])})
    _load_opt(Qb, fx["Q_inputs_params"])
    return prob


def wide_group(fx):
    """Seven Normal latents in ONE Group over a child plate (tests/golden/make_golden.py:gen_wide_group): under
    elbo_vi / elbo_rws that is 14 factors on one K, more than a single alan_reduce launch takes."""
    n = 7
    names = [f"g{i}" for i in range(n)]
    mean = eval("lambda " + ", ".join(names) + ": " + " + ".join(f"{0.5 + 0.25 * i} * {v}" for i, v in enumerate(names)))
    P = Plate(**{v: Normal(0.1 * i, 1.0 + 0.1 * i) for i, v in enumerate(names)}, p=Plate(d=Normal(mean, 1.5)))
    Q = Plate(grp=Group(**{v: Normal(OptParam(0.2 * i - 0.5), OptParam(0.05 * i - 0.1, transformation=t.exp))
                           for i, v in enumerate(names)}),
              p=Plate(d=Data()))
    sizes = fx["platesizes"]
    Qb = BoundPlate(Q, sizes)
    prob = Problem(BoundPlate(P, sizes), Qb, {"d": _named(fx["data"]["d"])})
    _load_opt(Qb, fx["Q_inputs_params"])
    return prob


BUILDERS = {
    "wide_group": wide_group,
    "linear_gaussian": linear_gaussian,
    "linear_gaussian_latents": linear_gaussian_latents,
    "model1": model1,
    "movielens": movielens,
    "bus_breakdown": bus_breakdown,
}


def sample_from_fixture(problem, fx, device="cpu"):
    """Rebuild the reference's sample tree as an alan_amd Sample (fresh Dim objects by name)."""
    K = fx["K"]
    Kdims = {g: Dim(name, K) for g, name in fx["Kdims"].items()}
    by_name = {**{str(d): d for d in Kdims.values()}, **{n: d for n, d in problem.all_platedims.items()}}

    def build(tree):
        out = {}
        for k, v in tree.items():
            if isinstance(v, dict):
                out[k] = build(v)
            else:
                x, names = v
                x = x.to(device)
                out[k] = x[tuple(by_name[n] for n in names)] if names else x
        return out

    return alan.Sample(problem=problem, sample=build(fx["sample"]), groupvarname2Kdim=Kdims,
                       sampler=alan.PermutationSampler, reparam=False)


# ---------------------------------------------------------------------------------------------
# The remaining problems of the reference's tests/test_problem_vs_itself.py:15-30 (constants, data and
# the sample tree come from tests/golden/e2e_small_models.pt).
from alan_amd import Beta, MultivariateNormal


def _data(fx, k):
    return {k: _named(fx["data"][k])}


def _small(name):
    def deco(f):
        SMALL[name] = f
        return f
    return deco


SMALL = {}


@_small("bernoulli_no_plate")
def _m(fx, c):
    P = Plate(p=Beta(2, 1), T=Plate(coin=Bernoulli("p")))
    Q = Plate(p=Beta(1, 1), T=Plate(coin=Data()))
    return P, Q, _data(fx, "coin")


def _two_params(qa, qb, reversed_=False):
    def build(fx, c):
        P = Plate(a=Normal(c["prior_mean"], c["a_scale"]), b=Normal("a", c["b_scale"]),
                  T=Plate(d=Normal("b", c["like_scale"])))
        Q = (Plate(b=qb(), a=qa(), T=Plate(d=Data())) if reversed_ else
             Plate(a=qa(), b=qb(), T=Plate(d=Data())))
        return P, Q, _data(fx, "d")
    return build


SMALL["linear_gaussian_two_params"] = _two_params(lambda: Normal(1, 4), lambda: Normal(1, 4))
SMALL["linear_gaussian_two_params_corr_Q"] = _two_params(lambda: Normal(1, 4), lambda: Normal("a", 1.2))
SMALL["linear_gaussian_two_params_corr_Q_reversed"] = _two_params(lambda: Normal("b", 1.2), lambda: Normal(1, 4), True)


@_small("linear_gaussian_two_params_dangling")
def _m(fx, c):
    mult = c["mult"]
    P = Plate(a=Normal(c["prior_mean"], c["prior_scale"]), b=Normal("a", 1.3),
              T=Plate(d=Normal(lambda a: mult * a, c["like_scale"])))
    Q = Plate(a=Normal(1, 4), b=Normal(lambda a: 1.2 * a, 1.2), T=Plate(d=Data()))
    return P, Q, _data(fx, "d")


@_small("linear_gaussian_latents_dangling")
def _m(fx, c):
    P = Plate(a=Normal(c["prior_mean"], c["prior_scale"]),
              T=Plate(z=Normal("a", c["z_scale"]), zp=Normal("a", 1.), d=Normal("z", c["d_scale"])))
    Q = Plate(a=Normal(1, 4),
              T=Plate(z=Normal(lambda a: 1.5 * a, 3.5), zp=Normal(lambda a: 1.5 * a, 3.5), d=Data()))
    return P, Q, _data(fx, "d")


@_small("linear_gaussian_latents_batch")
def _m(fx, c):
    P = Plate(a=Normal(c["prior_mean"], c["prior_scale"]),
              T=Plate(z=Normal("a", c["z_scale"]), d=Normal("z", c["d_scale"])))
    Q = Plate(a=Normal(t.zeros(2), 4), T=Plate(z=Normal(lambda a: 0.5 * a, 6), d=Data()))
    return P, Q, _data(fx, "d")


def _mvn(plated):
    def build(fx, c):
        like = MultivariateNormal("a", c["like_cov"])
        P = Plate(a=MultivariateNormal(c["prior_mean"], c["prior_cov"]), **({"T": Plate(d=like)} if plated else {"d": like}))
        Q = Plate(a=MultivariateNormal(c["ap_mean"], c["ap_cov"]), **({"T": Plate(d=Data())} if plated else {"d": Data()}))
        return P, Q, _data(fx, "d")
    return build


SMALL["linear_multivariate_gaussian"] = _mvn(False)
SMALL["linear_multivariate_gaussian_batch"] = _mvn(False)
SMALL["linear_multivariate_gaussian_param"] = _mvn(True)


def small_model(name, fx):
    P, Q, data = SMALL[name](fx, fx["consts"])
    sizes = fx["platesizes"]
    return Problem(BoundPlate(P, sizes), BoundPlate(Q, sizes), data)
