#!/usr/bin/env python3
"""Root-causing the "gradient of loc depends on host synchronisation" symptom of a replayed training graph
(tests/test_e2e_host.py::test_graphed_vi_step_does_not_depend_on_host_synchronisation).

Two changes went in together when the symptom was first seen: (a) long leading-dim torch reductions in the
outer-product producer's backward were split into short ones (dist._sum_leading), on the theory that the multi-block
reduction's semaphore memset node raced with its neighbours; (b) the capture moved onto the warm-up stream, so that
AccumulateGrad no longer ran on a parallel branch of the graph.  This probe reverts each one on its own (materialised
route, where that backward runs) and reports whether parameters after 4 replays depend on host synchronisation; with
--dot it also writes the captured graph for inspection of the memset nodes' edges.

    python3 tools/graph_race_probe.py [--dot DIR]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import alan_amd as alan
import bench
from alan_amd import dist as D

import ctypes as C
import collections
INSPECT = "--inspect" in sys.argv
NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
              7: "event_record", 10: "mem_alloc", 11: "mem_free"}


def inspect_graph(raw, label):
    """Node types and edge structure of a captured hipGraph_t: is the graph one chain, and what do the memset nodes
    (the reductions' semaphore clears) hang on?"""
    hip = C.CDLL("libamdhip64.so")
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    hip.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n))
    ne = C.c_size_t(0)
    hip.hipGraphGetEdges(C.c_void_p(raw), None, None, C.byref(ne))
    src, dst = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
    hip.hipGraphGetEdges(C.c_void_p(raw), src, dst, C.byref(ne))
    types = {}
    for nd in nodes:
        ty = C.c_int(0)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(ty))
        types[nd] = NODE_TYPES.get(ty.value, str(ty.value))
    ins, outs = collections.Counter(), collections.Counter()
    for a, b in zip(src, dst):
        outs[a] += 1
        ins[b] += 1
    roots = [nd for nd in nodes if ins[nd] == 0]
    forks = [nd for nd in nodes if outs[nd] > 1]
    joins = [nd for nd in nodes if ins[nd] > 1]
    print(f"[{label}] {n.value} nodes {dict(collections.Counter(types.values()))}, {ne.value} edges; roots {len(roots)} "
          f"({[types[r] for r in roots]}), forks {len(forks)}, joins {len(joins)}")
    ms = [nd for nd in nodes if types[nd] == "memset"]
    succ = collections.defaultdict(list)
    pred = collections.defaultdict(list)
    for a, b in zip(src, dst):
        succ[a].append(b)
        pred[b].append(a)
    for nd in ms:
        print(f"    memset node: preds {[types[p] for p in pred[nd]]}  succs {[types[q] for q in succ[nd]]}")
    return len(roots), len(forks), len(ms)
D.FUSE_PLATE_STEP = False              # the outer-product producer and its GEMM backward (dist._backward_outer)
real_sum_leading = D._sum_leading


def backward_outer_with_plain_torch_sums(ctx, G, v, l, s, v_shape, l_shape, s_shape, log_scale):
    """dist._FusedNormalLogProb._backward_outer as it stood when the symptom was seen (commit 9aca89c), with every
    reduction over a long leading dim as ONE torch reduction: d loc = T.view(9000, 540).sum(0), <G> = Gp.sum(0) over
    270,000 rows, the row-block sums of the scale gradient."""
    vd, ld, sd, od, _, _ = ctx.spec
    pos = {id(d): k for k, d in enumerate(od)}
    perm = [pos[id(d)] for d in (*vd, *ld, *sd)]
    nV, nL, nS, E = v.numel() // v.shape[-1], l.numel() // l.shape[-1], s.numel() // s.shape[-1], v.shape[-1]
    Gp = G.permute(*perm).reshape(nV * nL, nS)
    v2, l2, s2 = v.reshape(nV, 1, E), l.reshape(1, nL, E), s.reshape(nS, E)
    w2 = 1.0 / (s2 * s2)
    Dm = v2 - l2
    A = Gp @ w2
    T = Dm * A.view(nV, nL, E)
    gv = (-T.sum(1)).reshape(v_shape)
    gl = (T.view(nV, nL * E).sum(0) if "gl" in PLAIN else D._sum_leading(T.view(nV, nL * E))).reshape(l_shape)
    if "gl" in PLAIN and ERR is not None:
        # the same sum by the split route, compared on the device inside the captured iteration
        ERR.add_((gl.reshape(-1) - D._sum_leading(T.view(nV, nL * E)).reshape(-1)).abs().max() / gl.abs().max())
    rows = nV * nL
    blk = next((b for b in (1024, 1000, 900, 512, 500, 256, 250, 128, 100, 64, 50, 32, 30, 25, 16, 10, 8, 5, 4, 3, 2)
                if rows % b == 0), 1)
    D2 = (Dm * Dm).view(rows // blk, blk, E)
    S2 = t.bmm(Gp.view(rows // blk, blk, nS).transpose(1, 2), D2).view(rows // blk, nS * E)
    S2 = (S2.sum(0) if "S2" in PLAIN else D._sum_leading(S2)).view(nS, E)
    S0 = (Gp.sum(0) if "S0" in PLAIN else D._sum_leading(Gp)).unsqueeze(-1)
    gs = w2 * S2 - S0 if log_scale else (w2 * S2 - S0) / s2
    return gv, gl, gs.reshape(s_shape)


real_backward_outer = D._FusedNormalLogProb._backward_outer
PLAIN = {"gl", "S0", "S2"}
ERR = None


def run(sync_each, other_stream, tag=None):
    t.manual_seed(0)
    t.cuda.manual_seed_all(0)
    prob = bench.build_problem("cuda")
    opt = t.optim.Adam(prob.parameters(), lr=1e-2, capturable=True)
    step = alan.GraphedStep(prob, 30, opt, method="vi", capture_stream=t.cuda.Stream() if other_stream else None,
                            allow_memset_nodes=True)
    if INSPECT and tag:
        inspect_graph(step.graph.raw_cuda_graph(), tag)
    t.cuda.synchronize()
    for _ in range(4):
        step()
        if sync_each:
            t.cuda.synchronize()
    t.cuda.synchronize()
    return [p.detach().clone() for p in prob.parameters()]


def verdict(name, plain_sum, other_stream):
    D._FusedNormalLogProb._backward_outer = staticmethod(backward_outer_with_plain_torch_sums) if plain_sum \
        else real_backward_outer
    a = run(True, other_stream, name.replace(" ", "_"))
    worst, bad = 0.0, 0
    for trial in range(3):
        b = run(False, other_stream)
        for x, y in zip(a, b):
            if not t.equal(x, y):
                bad += 1
                worst = max(worst, float((x - y).abs().max()))
    print(f"{name:58s} mismatching parameter tensors over 3 trials: {bad:2d}  max |diff| {worst:.3e}")


verdict("as shipped (split sums, capture on warm-up stream)", False, False)
verdict("(a) reverted: plain long reductions", True, False)
verdict("(b) reverted: capture on a fresh stream", False, True)
verdict("(a) and (b) reverted", True, True)
for which in ("gl", "S0", "S2"):
    PLAIN = {which}
    verdict(f"only the {which} reduction as one torch sum", True, False)
PLAIN = set()
verdict("historic function, all three sums split (control)", True, False)

# which mode is wrong?  d loc by one torch sum against the split route, inside the same captured iteration
PLAIN = {"gl"}
D._FusedNormalLogProb._backward_outer = staticmethod(backward_outer_with_plain_torch_sums)
for sync_each in (True, False):
    ERR = t.zeros((), device="cuda")
    run(sync_each, False)
    print(f"one-sum d loc vs split-sum d loc, accumulated relative error over 4 replays, "
          f"{'synchronised' if sync_each else 'back-to-back'}: {float(ERR):.3e}  (also counts the warm-up and capture passes: 0 there)")
