"""Localise a disagreement of the VI gradients: fused route (variants) against autograd through torch.distributions."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch as t
import alan_amd as alan
from alan_amd import dist as D
from conftest import load_golden
import models

fx = load_golden("e2e_movielens_K10.pt")


def grads(**flags):
    old = {k: getattr(D, k) for k in flags}
    for k, v in flags.items():
        setattr(D, k, v)
    try:
        prob = models.BUILDERS["movielens"](fx).to("cuda")
        t.manual_seed(11)
        t.cuda.manual_seed_all(11)
        sample = prob.sample(5, reparam=True)
        elbo = sample.elbo_vi(alan.no_checkpoint)
        elbo.backward()
        return float(elbo), {n: p.grad.detach().cpu().double().clone() for n, p in prob.named_parameters() if p.grad is not None}
    finally:
        for k, v in old.items():
            setattr(D, k, v)


e0, g0 = grads(FUSE_NORMAL=False)
for name, fl in (("all fused", {}), ("HIP_PRODUCER_BACKWARD off", dict(HIP_PRODUCER_BACKWARD=False)),
                 ("OUTER_BACKWARD off", dict(OUTER_BACKWARD=False)),
                 ("both off", dict(HIP_PRODUCER_BACKWARD=False, OUTER_BACKWARD=False))):
    e1, g1 = grads(**fl)
    print(f"{name}: elbo {e1:.4f} vs {e0:.4f}")
    for n in g0:
        d = (g1[n] - g0[n]).abs().max().item()
        print(f"    {n:32s} max|diff| {d:.3e}   max|ref| {g0[n].abs().max().item():.3e}")

# ---- what do the producer_grads calls of this model look like?
from alan_amd import engine as E
real = E.producer_grads


def spy(G, out_dims, args, wanted, kinds, log_scale=False, scale=1.0):
    res = real(G, out_dims, args, wanted, kinds, log_scale=log_scale, scale=scale)
    print("producer_grads: G", tuple(G.shape), G.stride(), G.dtype, "od", [str(d) for d in out_dims], "log_scale", log_scale,
          "scale", scale, "wanted", tuple(wanted), "->", None if res is None else [None if r is None else tuple(r.shape) for r in res])
    for x, d in args:
        print("      arg", tuple(x.shape), x.stride(), x.dtype, [str(k) for k in d])
    if len(args) == 2:
        y, x = args[0][0].double(), args[1][0].double()
        want = scale * G.double()[:, :, None] * (y[:, None, :] - t.sigmoid(x))
        print("      d logits max|diff|", (res[1].double() - want).abs().max().item(), "max|ref|", want.abs().max().item())
    else:
        leaves = [a[0].detach().double().clone().requires_grad_(True) for a in args]
        sc = leaves[2].exp() if log_scale else leaves[2]
        lp = t.distributions.Normal(leaves[1][:, None, :], sc[:, None, :]).log_prob(leaves[0]).sum(-1)
        want = t.autograd.grad((scale * lp * G.double()).sum(), leaves)
        for nm, a, b in zip(("value", "loc", "scale"), res, want):
            print(f"      d {nm} max|diff|", (a.double() - b).abs().max().item(), "max|ref|", b.abs().max().item())
    return res


E.producer_grads = spy
grads()
