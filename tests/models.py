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


BUILDERS = {
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
