"""The user's lambda in movielens is `z @ x` ([300,30,18] x [300,5,18] -> [300,30,5]): a tiny batched GEMM that
hipBLASLt runs in ~10 us.  Which formulation / BLAS backend is fastest?  Usage: python tools/small_bmm_probe.py"""
import time
import torch as t

M, K, N, D = 300, 30, 5, 18
z = t.randn(M, K, D, device="cuda")
x = t.randn(M, N, D, device="cuda")


def bench(name, fn):
    g = t.cuda.CUDAGraph()
    s = t.cuda.Stream()
    s.wait_stream(t.cuda.current_stream())
    with t.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        with t.cuda.graph(g, stream=s):
            for _ in range(20):
                out = fn()
    t.cuda.synchronize()
    for _ in range(3):
        g.replay()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    t.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / 50 / 20 * 1e6:7.2f} us per call (graph of 20)", flush=True)


for lib in ("default", "hipblaslt", "hipblas"):
    if lib != "default":
        try:
            t.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable:", e)
            continue
    print("preferred_blas_library:", t.backends.cuda.preferred_blas_library())
    bench("bmm(z, x^T)", lambda: t.bmm(z, x.transpose(1, 2)))
    bench("matmul(z[:, :, None, :], x[:, None, :, :, None])", lambda: t.matmul(z[:, :, None, None, :], x[:, None, :, :, None]))
    bench("einsum mkd,mnd->mkn", lambda: t.einsum("mkd,mnd->mkn", z, x))
bench("(z[:, :, None, :] * x[:, None, :, :]).sum(-1)", lambda: (z[:, :, None, :] * x[:, None, :, :]).sum(-1))
vm = t.vmap(t.vmap(t.vmap(lambda a, b: a @ b, in_dims=(None, 0)), in_dims=(0, None)), in_dims=(0, 0))
bench("nested vmap of a @ b (what the model does)", lambda: vm(z, x))
