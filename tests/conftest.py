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
def _oracle_launch(mode, factors, sizes, roles, out, out_dims, weight=None, lse_out=None, add_const=0.0,
                   scales=None):
    import torch as t
    from oracle import alan_oracle as orc
    from alan_amd import native as N
    space = tuple(sizes)
    dtype = out.dtype
    x = 0
    for i, (f, dims) in enumerate(factors):
        s = 1.0 if scales is None else scales[i]
        x = x + s * orc.align((f.to(dtype), tuple(dims)), space)
    x = x.expand([sizes[d] for d in space])
    red = [i for i, d in enumerate(space) if roles[d] == N.REDUCE]
    plate = [i for i, d in enumerate(space) if roles[d] == N.PLATE]
    names = list(space)

    def drop(tensor, axes, names):
        return tensor, [n for i, n in enumerate(names) if i not in axes]

    if mode == N.MODE_LSE:
        if red:
            v, vn = orc.logsumexp_dims((x, space), tuple(space[i] for i in red))
        else:
            v, vn = x, space
        if lse_out is not None:
            lse_out[0].copy_(orc.align((v, tuple(vn)), tuple(lse_out[1])).reshape(lse_out[0].shape))
        if plate:
            axes = [vn.index(space[i]) for i in plate]
            v = v.sum(axes)
            vn = tuple(n for n in vn if n not in [space[i] for i in plate])
        res, rn = v + add_const, vn
    elif mode == N.MODE_SUM:
        res = (x.sum(red) if red else x) + add_const
        rn = tuple(n for i, n in enumerate(space) if i not in red)
    else:
        w = orc.align((weight[0].to(dtype), tuple(weight[1])), space)
        y = w * x.exp()
        res = y.sum(red) if red else y
        rn = tuple(n for i, n in enumerate(space) if i not in red)
    present = [d for d in out_dims if d in rn]
    res = orc.align((res, tuple(rn)), tuple(present)) if present else res
    out.copy_(res.reshape(out.shape))


def _oracle_chain(ms, want_chain=False):
    from oracle import alan_oracle as orc
    import torch as t
    chain = orc.chain_logmmexp(ms)
    return t.logsumexp(chain, -1), (chain if want_chain else None)


@pytest.fixture
def oracle_backend(monkeypatch):
    """Route alan_amd's launch seam to the CPU oracle for the duration of one test."""
    from alan_amd import engine, native
    monkeypatch.setattr(engine, "_launch", _oracle_launch)
    monkeypatch.setattr(native, "chain_logmmexp", _oracle_chain)
    monkeypatch.setattr(native, "require_device", lambda x, what="tensor": None)
    yield
