"""
Stand-in for the third-party planner ``opt_einsum`` (unpinned in the reference's
setup.py:13; not vendored under /root/reference; not installed in this image; no network).

Used ONLY by tests/golden/make_golden.py, in the build container, so that
``import alan`` (reference, reduce_Ks.py:1) succeeds.  The reference's single call site,
``opt_einsum.contract_path(*args)[0]`` (reduce_Ks.py:265), consumes only the ORDER of
pairwise eliminations; no arithmetic happens here.  Greedy smallest-intermediate-first.
"""
import math


def contract_path(*args):
    operands, out_idx = args[:-1], set(args[-1])
    tensors, idxs = operands[0::2], [tuple(ix) for ix in operands[1::2]]
    size = {}
    for x, ix in zip(tensors, idxs):
        for ax, i in enumerate(ix):
            size[i] = x.shape[ax]
    if len(idxs) == 1:
        return [(0,)], None
    cur = [set(ix) for ix in idxs]
    path = []
    while len(cur) > 1:
        best = None
        for i in range(len(cur)):
            for j in range(i + 1, len(cur)):
                rest = set()
                for k in range(len(cur)):
                    if k != i and k != j:
                        rest |= cur[k]
                res = {d for d in (cur[i] | cur[j]) if d in out_idx or d in rest}
                cost = math.prod(size[d] for d in res)
                if best is None or cost < best[0]:
                    best = (cost, i, j, res)
        _, i, j, res = best
        path.append((i, j))
        cur = [c for k, c in enumerate(cur) if k != i and k != j] + [res]
    return path, None
