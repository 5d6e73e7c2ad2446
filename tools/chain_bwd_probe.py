#!/usr/bin/env python3
"""The timeseries chain's backward (alan_chain_logmmexp_backward_batched), timed by HIP events around N back-to-back calls:
the one-launch tree backward (round 4, default) against a launch per round (ALAN_CHAIN_BWD_TREE=0), in child processes so
that each reads its own environment.   python3 tools/chain_bwd_probe.py [T] [K ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(T, Ks):
    import torch as t
    from alan_amd import native as N
    for K in Ks:
        g = t.Generator(device="cuda").manual_seed(K)
        ms = t.randn(T, K, K, device="cuda", generator=g) * 2.0 - 3.0
        vec, _, tree = N.chain_logmmexp(ms)
        gv = t.randn(K, device="cuda", generator=g)
        run = lambda: N.chain_logmmexp_backward(ms, tree, out_vec=vec, grad_vec=gv)
        ref = run()
        for _ in range(5):
            run()
        n = 50
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        t.cuda.synchronize()
        a.record()
        for _ in range(n):
            out = run()
        b.record()
        t.cuda.synchronize()
        fa, fb = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        fa.record()
        for _ in range(n):
            N.chain_logmmexp(ms)
        fb.record()
        t.cuda.synchronize()
        print(f"T={T} K={K} tree={os.environ.get('ALAN_CHAIN_BWD_TREE', '1')}: backward {a.elapsed_time(b) / n * 1e3:.1f} us per call, "
              f"forward {fa.elapsed_time(fb) / n * 1e3:.1f}; finite {bool(t.isfinite(out).all())}, repeat-equal {bool(t.equal(out, ref))}, "
              f"|grad| sum {float(out.abs().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), [int(x) for x in sys.argv[3:]])
    else:
        T = sys.argv[1] if len(sys.argv) > 1 else "1000"
        Ks = sys.argv[2:] or ["30", "100"]
        for knob in ("1", "0"):
            env = dict(os.environ, ALAN_CHAIN_BWD_TREE=knob)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", T, *Ks], env=env, check=False, timeout=300)
