#!/usr/bin/env python3
"""Does a replayed HIP graph run independent branches concurrently?  4 chains of 12 small elementwise
kernels captured (a) on one stream, (b) forked over 4 side streams and joined."""
import time
import torch as t

dev = "cuda"
xs = [t.randn(5000, device=dev) for _ in range(4)]
big = t.randn(8_000_000, device=dev)


def chain(x, n=12):
    for _ in range(n):
        x = x * 1.0001 + 0.5
    return x


def work_serial():
    return [chain(x) for x in xs], big.exp()


side = [t.cuda.Stream() for _ in range(4)]


def work_forked():
    main = t.cuda.current_stream()
    outs = []
    for s, x in zip(side, xs):
        s.wait_stream(main)
        with t.cuda.stream(s):
            outs.append(chain(x))
    b = big.exp()
    for s in side:
        main.wait_stream(s)
    return outs, b


def capture(fn):
    s = t.cuda.Stream()
    s.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(s):
        fn()
    t.cuda.current_stream().wait_stream(s)
    t.cuda.synchronize()
    g = t.cuda.CUDAGraph()
    with t.cuda.graph(g, capture_error_mode="thread_local"):
        out = fn()
    return g, out


for name, fn in (("serial", work_serial), ("forked", work_forked)):
    g, out = capture(fn)
    for _ in range(5):
        g.replay()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        g.replay()
    t.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e6:.1f} us per replay (49 kernels)", flush=True)
    ref = [chain(x) for x in xs]
    for a, b in zip(out[0], ref):
        assert t.allclose(a, b)
