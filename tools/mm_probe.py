"""The movielens logits lambda `z @ x` under (a) torchdim with the Dim creation order of a Split eval and
(b) nested torch.vmap with an explicit nesting order: which GEMM runs, and how long does it take?"""
import torch as t, time
from functorch.dim import Dim
def bench(f, n=30):
    for _ in range(5): f()
    t.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    t.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
fn = lambda z, x: z @ x
M,K,N=38,100,5
dn=Dim('plate_2',N); dk=Dim('K_z',K); dm=Dim('plate_1_split_3',M)    # creation order as in a Split eval
z=t.randn(M,K,18,device='cuda'); x=t.randn(M,N,18,device='cuda')
zd=z[dm,dk]; xd=x[dm,dn]
print("torchdim (split-order dims)", bench(lambda: fn(zd, xd)), "us", flush=True)
def vm(z, x):
    f = t.vmap(fn, in_dims=(None, 0))          # plate_2 (x only), innermost
    f = t.vmap(f, in_dims=(0, None))           # K_z (z only)
    f = t.vmap(f, in_dims=(0, 0))              # plate_1 (both), outermost
    return f(z, x)
print("nested vmap [m,k,n]        ", bench(lambda: vm(z, x)), "us", flush=True)
ref = (z[:, :, None, :] * x[:, None, :, :]).sum(-1)
print("max err", float((vm(z, x) - ref).abs().max()), float((fn(zd, xd).order(dm, dk, dn) - ref).abs().max()))
