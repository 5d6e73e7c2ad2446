"""Debug probe: gradients of the nested-timeseries ELBO on the GPU against the CPU oracle backend, with individual
HIP pieces swapped for the oracle to localise a disagreement.  Usage: python tools/nested_ts_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as t
import alan_amd as alan
from alan_amd import native as N
from oracle import backend
import test_timeseries as TT

K, T = 30, 25


def grads(prob, sample):
    for p in prob.parameters():
        p.grad = None
    sample.elbo_rws(alan.no_checkpoint).backward()
    return {n: p.grad.detach().cpu().clone() for n, p in prob.Q.named_parameters()}


prob, _ = TT.nested_problem(T, opt=True)
prob.to("cuda")
t.manual_seed(2)
sample = prob.sample(K, reparam=False)
cpu_prob, _ = TT.nested_problem(T, opt=True)
with backend.installed():
    cs = TT._same_sample_on_cpu(sample, cpu_prob, K)
    want = grads(cpu_prob, cs)


def report(tag, got):
    print(tag)
    for n in want:
        d = (got[n] - want[n]).abs().max().item()
        print(f"   {n:28s} max|diff| {d:.3e}   max|want| {want[n].abs().max().item():.3e}")


report("HIP as is", grads(prob, sample))

saved = N.chain_logmmexp_backward
N.chain_logmmexp_backward = lambda ms, vec, g: backend.oracle_chain_backward(ms.cpu(), vec.cpu(), g.cpu()).to(ms.device)
report("chain backward from the oracle", grads(prob, sample))
N.chain_logmmexp_backward = saved

saved = N.run_reduce_backward
N.run_reduce_backward = lambda desc, device: False
report("one-pass rows backward disabled", grads(prob, sample))
N.run_reduce_backward = saved

from alan_amd import dist as D
for flag in ("FUSE_NORMAL", "OUTER_BACKWARD"):
    old = getattr(D, flag)
    setattr(D, flag, False)
    report(f"dist.{flag} = False", grads(prob, sample))
    setattr(D, flag, old)

# ---- capture the chain backward's actual operands and compare kernel variants on them
cap = {}
saved = N.chain_logmmexp_backward


def spy(ms, vec, g):
    cap.update(ms=ms, vec=vec, g=g)
    return saved(ms, vec, g)


N.chain_logmmexp_backward = spy
grads(prob, sample)
N.chain_logmmexp_backward = saved
ms, vec, g = cap["ms"], cap["vec"], cap["g"]
print("ms", tuple(ms.shape), ms.stride(), "vec", tuple(vec.shape), vec.stride(), "g", tuple(g.shape), g.stride(), g.dtype)
print("g range", g.min().item(), g.max().item(), "rows all-zero:", int((g == 0).all(-1).sum()))
want_g = backend.oracle_chain_backward(ms.cpu().double(), vec.cpu().double(), g.cpu().double())
a = saved(ms, vec, g).cpu().double()
b = saved(ms.contiguous(), vec.contiguous(), g.contiguous()).cpu().double()
c = t.stack([saved(ms[i].contiguous(), vec[i].contiguous(), g[i].contiguous()) for i in range(ms.shape[0])], 0).cpu().double()
for tag, x in (("batched as given", a), ("batched contiguous", b), ("loop of unbatched", c)):
    d = (x - want_g).abs()
    bad = (d.reshape(d.shape[0], -1).max(-1).values > 1e-3 * want_g.abs().max()).nonzero().flatten().tolist()
    print(f"{tag:22s} max|diff| {d.max().item():.3e} of {want_g.abs().max().item():.3e}; bad batch elements: {bad[:20]}")
# recompute vec from ms with the oracle: is the saved vec the forward's?
v2, _ = backend.oracle_chain(ms.cpu().double())
print("saved vec vs oracle forward", (v2 - vec.cpu().double()).abs().max().item())
