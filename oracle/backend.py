"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see alan_oracle.py).

``oracle_launch`` has the signature of ``alan_amd.engine._launch`` (the single seam between the host
stack and libalan_mi355.so) and computes the same contraction with the CPU oracle.  It exists so
that (a) tests can exercise the HOST logic (plate recursion, Split chunking, planner, autograd
wiring, multi-rank sum) without a GPU and (b) bench.py can time the CPU baseline of a whole ELBO.
It is installed only by tests/conftest.py's ``oracle_backend`` fixture and by bench.py's
``cpu_baseline`` leg -- never by the product, which refuses CPU tensors.
"""
import contextlib


def oracle_launch(mode, factors, sizes, roles, out, out_dims, weight=None, lse_out=None, add_const=0.0,
                   scales=None):
    import torch as t
    from oracle import alan_oracle as orc
    from alan_amd import native as N
    space = tuple(sizes)
    # the library's descriptor limits (include/alan_mi355.h): the checker refuses what the product would refuse
    if len(space) > N.MAX_DIMS:
        raise N.NativeError(f"alan_amd: {len(space)} dims in one contraction step (max {N.MAX_DIMS})")
    if len(factors) > N.MAX_FACTORS:
        raise N.NativeError(f"alan_amd: {len(factors)} factors in one contraction step (max {N.MAX_FACTORS})")
    dtype = out.dtype
    # role PRESUM (the slices of a partial plate sum, logpq.py:149 finished by the consumer): summed within the one
    # factor that carries the dim, before the factors are added
    presum = [d for d in space if roles[d] == getattr(N, "PRESUM", 4)]
    if presum:
        assert len(presum) == 1 and sum(presum[0] in dims for _, dims in factors) == 1
        factors = [(f.sum(list(dims).index(presum[0])), tuple(d for d in dims if d != presum[0])) if presum[0] in dims
                   else (f, dims) for f, dims in factors]
        sizes = {d: n for d, n in sizes.items() if d != presum[0]}
        space = tuple(sizes)
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


def oracle_chain(ms, want_chain=False):
    """[T,K,K] or a batch [B,T,K,K]: the reference keeps batch dims as torchdims, i.e. one independent chain each."""
    from oracle import alan_oracle as orc
    import torch as t
    if ms.ndim == 4:
        chain = t.stack([orc.chain_logmmexp(m) for m in ms], 0)
    else:
        chain = orc.chain_logmmexp(ms)
    return t.logsumexp(chain, -1), (chain if want_chain else None), None


def oracle_chain_backward(ms, tree, out_vec=None, grad_vec=None, grad_chain=None):
    import torch as t
    from oracle import alan_oracle as orc
    with t.enable_grad():
        x = ms.detach().clone().requires_grad_(True)
        chains = t.stack([orc.chain_logmmexp(m) for m in x], 0) if x.ndim == 4 else orc.chain_logmmexp(x)
        total = 0
        if grad_vec is not None:
            out = t.logsumexp(chains, -1)
            total = total + (out * grad_vec.reshape(out.shape)).sum()
        if grad_chain is not None:
            total = total + (chains * grad_chain.reshape(chains.shape)).sum()
        (grad,) = t.autograd.grad(total, x)
    return grad


@contextlib.contextmanager
def installed():
    """Temporarily route alan_amd's launch seam to the oracle (CPU tensors accepted)."""
    from alan_amd import engine, native
    saved = (engine._launch, native.chain_logmmexp, native.require_device, native.chain_logmmexp_backward,
             native.run_reduce_backward)
    engine._launch, native.chain_logmmexp = oracle_launch, oracle_chain
    native.chain_logmmexp_backward = oracle_chain_backward
    native.require_device = lambda x, what="tensor": None
    native.run_reduce_backward = lambda desc, device: False      # "not this shape": per-factor WEXPSUM launches
    try:
        yield
    finally:
        (engine._launch, native.chain_logmmexp, native.require_device, native.chain_logmmexp_backward,
         native.run_reduce_backward) = saved
