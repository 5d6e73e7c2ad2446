#!/usr/bin/env python3
"""Does a bare torch multi-block reduction misbehave inside a replayed HIP graph -- no alan kernel involved?
(Follow-up of tools/graph_race_probe.py, which isolates the long torch reductions as the trigger.)
A graph holding   y = x.sum(0);  err += max|y - y_ref|   is replayed back to back (no host synchronisation) and with
a synchronisation after every replay; x.sum(0) over a long leading dim launches a multi-block reduce kernel behind a
hipMemsetAsync of its semaphore buffer (ATen/native/cuda/Reduce.cuh: the kernel never resets the semaphores itself)."""
import sys
import torch as t

def trial(rows, cols, neighbours, sync_each, replays=300, chain=0):
    g = t.Generator(device="cuda").manual_seed(0)
    x = t.randn(rows, cols, device="cuda", generator=g)
    w = t.randn(cols, 64, device="cuda", generator=g)
    y_ref = x.sum(0)
    err = t.zeros((), device="cuda")
    side = t.cuda.Stream()
    side.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(side):
        for _ in range(3):
            y = x.sum(0)
            if neighbours:
                z = (y @ w).relu().sum()
            err += (y - y_ref).abs().max()
    t.cuda.current_stream().wait_stream(side)
    t.cuda.synchronize()
    err.zero_()
    gr = t.cuda.CUDAGraph()
    pad = t.zeros(64, device="cuda")
    with t.cuda.graph(gr, stream=side):
        for _ in range(chain):
            pad.add_(1.0)                        # a long chain of small kernels ahead of the reduction
        if neighbours:
            x.mul_(1.0)                          # a kernel writing the reduction's input just before it
        y = x.sum(0)
        if neighbours:
            z = (y @ w).relu().sum()
        err += (y - y_ref).abs().max()
        for _ in range(chain):
            pad.add_(1.0)
    t.cuda.synchronize()
    for _ in range(replays):
        gr.replay()
        if sync_each:
            t.cuda.synchronize()
    t.cuda.synchronize()
    return float(err)

for rows, cols in ((9000, 540), (270000, 30), (270, 540)):
    for neighbours in (False, True):
        a = trial(rows, cols, neighbours, True)
        b = trial(rows, cols, neighbours, False)
        print(f"x[{rows}, {cols}].sum(0) {'with neighbours' if neighbours else 'alone          '}: accumulated |error| over 300 replays: "
              f"synchronised {a:.3e}   back-to-back {b:.3e}")

for chain in (20, 100):
    a = trial(9000, 540, True, True, chain=chain)
    b = trial(9000, 540, True, False, chain=chain)
    print(f"x[9000, 540].sum(0) inside a chain of {2 * chain} small kernels: synchronised {a:.3e}   back-to-back {b:.3e}")
# the producer of the reduction's input is itself a big kernel (as in the training graph: T = D * A over 4.9 M elements)
def trial_big(sync_each, replays=100):
    g = t.Generator(device="cuda").manual_seed(0)
    d = t.randn(9000, 30, 18, device="cuda", generator=g)
    a = t.randn(9000, 30, 18, device="cuda", generator=g)
    ref = (d * a).view(9000, 540).sum(0)
    err = t.zeros((), device="cuda")
    side = t.cuda.Stream()
    side.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(side):
        for _ in range(3):
            y = (d * a).view(9000, 540).sum(0)
            err += (y - ref).abs().max()
    t.cuda.current_stream().wait_stream(side)
    t.cuda.synchronize()
    err.zero_()
    gr = t.cuda.CUDAGraph()
    with t.cuda.graph(gr, stream=side):
        T = d * a
        y = T.view(9000, 540).sum(0)
        err += (y - ref).abs().max()
    t.cuda.synchronize()
    for _ in range(replays):
        gr.replay()
        if sync_each:
            t.cuda.synchronize()
    t.cuda.synchronize()
    return float(err)
print(f"(d * a).view(9000, 540).sum(0), temp input: synchronised {trial_big(True):.3e}   back-to-back {trial_big(False):.3e}")
