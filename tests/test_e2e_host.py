"""End-to-end ELBO of the alan_amd host stack on the reference's own sample trees (golden e2e
fixtures): CPU runs use the test-only oracle backend (host logic), GPU runs use the HIP library."""
import pytest
import torch as t

import alan_amd as alan
from conftest import load_golden
import models

CASES = [
    ("e2e_linear_gaussian.pt", "linear_gaussian", alan.Split("T", 4)),
    ("e2e_linear_gaussian_latents.pt", "linear_gaussian_latents", alan.Split("T", 3)),
    ("e2e_model1.pt", "model1", alan.Split("p1", 3)),
    ("e2e_movielens_K3.pt", "movielens", alan.Split("plate_1", 38)),
    ("e2e_movielens_K10.pt", "movielens", alan.Split("plate_1", 38)),
    ("e2e_bus_breakdown_K3.pt", "bus_breakdown", alan.Split("plate_ID", 40)),
    ("e2e_bus_breakdown_K10.pt", "bus_breakdown", alan.Split("plate_ID", 40)),
]


def _check(fixture, model, split, device):
    fx = load_golden(fixture)
    prob = models.BUILDERS[model](fx)
    prob.to(device)
    sample = models.sample_from_fixture(prob, fx, device)
    strategies = {"no_checkpoint": alan.no_checkpoint, "checkpoint": alan.checkpoint, "split": split}
    for name, strat in strategies.items():
        got = sample.elbo_nograd(strat)
        ref = fx["elbo"][name]
        assert got.ndim == 0
        # north_star tolerance: ELBO within 1e-4 relative of the reference's CPU value
        assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (name, float(got), float(ref))


@pytest.mark.parametrize("fixture,model,split", CASES, ids=[c[0][4:-3] for c in CASES])
def test_elbo_matches_reference_host_logic(fixture, model, split, oracle_backend):
    _check(fixture, model, split, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,model,split", CASES, ids=[c[0][4:-3] for c in CASES])
def test_elbo_matches_reference_hip(fixture, model, split):
    _check(fixture, model, split, "cuda")
