import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---------------------------------------------------------------------------------------------
# CPU checker backend (TEST-ONLY).  The product has no CPU path: alan_amd.engine._launch always
# goes to libalan_mi355.so and refuses CPU tensors.  To exercise the HOST logic (plate recursion,
# Split chunking, planner, autograd wiring, multi-rank sum) on a machine without a GPU, these
# fixtures swap the single launch seam for the oracle.  Nothing outside tests/ can do this.
from oracle.backend import oracle_launch as _oracle_launch, oracle_chain as _oracle_chain  # noqa: E402


@pytest.fixture
def oracle_backend(monkeypatch):
    """Route alan_amd's launch seam to the CPU oracle for the duration of one test."""
    from alan_amd import engine, native
    monkeypatch.setattr(engine, "_launch", _oracle_launch)
    monkeypatch.setattr(native, "chain_logmmexp", _oracle_chain)
    from oracle.backend import oracle_chain_backward
    monkeypatch.setattr(native, "chain_logmmexp_backward", oracle_chain_backward)
    monkeypatch.setattr(native, "require_device", lambda x, what="tensor": None)
    # the one-pass backward is a GPU kernel: "not this shape" sends autograd down the per-factor WEXPSUM seam
    monkeypatch.setattr(native, "run_reduce_backward", lambda desc, device: False)
    yield
