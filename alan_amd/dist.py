"""
Distribution layer (host side, stays on PyTorch-ROCm as north_star specifies): wraps
``torch.distributions`` so parameters and samples may be torchdim tensors, and provides the
user-facing ``alan.Normal(...)``-style specs.  Mirrors dist.py / TorchDimDist.py of the reference in
behaviour; the implementation differs in one respect that matters to the HIP path:
``TorchDimDist.log_prob`` lays the resulting factor out with the dims in a caller-chosen order, and
the plate recursion asks for  [plates..., parent Ks..., own K]  so that the K dim which reduce_Ks
eliminates is the CONTIGUOUS dim of the big factor -- the layout the rows kernel streams at HBM rate.
"""
import inspect
import math
import os
import types

import torch as t
import torch.distributions as td
import torch.nn as nn

from .dims import PT, ExpPT, LazyNormalPT, LinearPT, PendingPT, ReparamPT, ScaledPT, ShiftPT, is_tensor, pt_align, pt_order

Number = (int, float)

# torch >= 2.10 made arg_constraints an instance property for these
_EVENT_NDIM_FALLBACK = {
    "Uniform": {"low": 0, "high": 0},
    "Wishart": {"df": 0, "covariance_matrix": 2, "precision_matrix": 2, "scale_tril": 2},
}


def _arg_event_ndim(dist_cls, name):
    ac = dist_cls.__dict__.get("arg_constraints", getattr(dist_cls, "arg_constraints", None))
    if isinstance(ac, dict) and name in ac:
        return ac[name].event_dim
    return _EVENT_NDIM_FALLBACK.get(dist_cls.__name__, {}).get(name, 0)


def _arg_is_discrete(dist_cls, name):
    ac = getattr(dist_cls, "arg_constraints", None)
    if isinstance(ac, dict) and name in ac:
        return bool(ac[name].is_discrete)
    return False


VALIDATE_ARGS = False
"""torch.distributions argument validation.  Off by default: on a GPU every validated construction
costs a device synchronisation (``constraint.check(value).all()``), which dominates small ELBO evals.
Set ``alan_amd.dist.VALIDATE_ARGS = True`` to get the reference's behaviour (exceptions on invalid
parameters / samples)."""


class _FusedNormalLogProb(t.autograd.Function):
    """sum_event log N(value; loc, scale) over the cross product of the arguments' first-class dims.

    forward : ONE HIP launch (alan_reduce mode NORMAL) -- the [.., K, K, K, event] broadcast of
              TorchDimDist.py:127-162 is never materialised.
    backward: closed-form contractions of the upstream gradient G (shaped like the output) with the
              small arguments, as ``torch.einsum`` calls -- per-factor log-prob gradients stay on PyTorch,
              but still without any event-sized broadcast:
                d value = -(value * <G, w> - <G, loc*w>),  w = 1/scale^2
                d loc   =  <G, value*w> - loc * <G, w>
                d scale =  (<G, value^2> - 2 <G, value*loc> + <G, loc^2>) / scale^3 - <G> / scale
              where <G, a*b> sums over every dim the result lacks."""

    @staticmethod
    def forward(ctx, spec, value, loc, scale):
        """spec = (value dims, loc dims, scale dims, out dims, log_scale, (a, b)): with ``log_scale`` the third
        tensor is log(scale) (and its gradient is returned wrt that); the result is a * log_prob + b."""
        from . import engine as E
        vd, ld, sd, od, log_scale, affine = spec
        out = E.normal_logprob((value.detach(), vd), (loc.detach(), ld), (scale.detach(), sd), od,
                               log_scale=log_scale, affine=affine)
        ctx.spec = spec
        ctx.save_for_backward(value, loc, scale)
        return out

    @staticmethod
    def backward(ctx, G):
        vd, ld, sd, od, log_scale, affine = ctx.spec
        value, loc, scale = ctx.saved_tensors
        ids = lambda ds: {id(d) for d in ds}
        # (the GEMM formulation is for the big cross-product factor: loc and scale both carry dims of their own; a prior
        # with constant parameters is a small factor and takes the producer-gradient kernels below)
        outer = OUTER_BACKWARD and ids(ld) and ids(sd) and not (ids(vd) & ids(ld)) and not (ids(vd) & ids(sd)) \
            and not (ids(ld) & ids(sd)) and ids(od) == ids(vd) | ids(ld) | ids(sd)
        if HIP_PRODUCER_BACKWARD and G.is_cuda and not outer:
            from . import engine as E
            from . import native as N
            res = E.producer_grads(G, od, [(value, vd), (loc, ld), (scale, sd)], ctx.needs_input_grad[1:4],
                                   (N.GRAD_VALUE, N.GRAD_LOC, N.GRAD_SCALE), log_scale=log_scale, scale=affine[0])
            if res is not None:
                return (None, *res)
        if affine[0] != 1.0:
            G = G * affine[0]
        raw_scale = scale
        if log_scale:
            scale = scale.detach().exp()
        letters = {}

        def sub(dims, event=True):
            return "".join(letters.setdefault(id(d), chr(ord("a") + len(letters))) for d in dims) + \
                ("Z" if event else "")

        nev = value.ndim - len(vd)
        flat = lambda x, d: x.reshape(*x.shape[: len(d)], -1) if nev else x.unsqueeze(-1)
        v, l, s = flat(value.detach(), vd), flat(loc.detach(), ld), flat(scale.detach(), sd)
        if outer:
            return (None, *_FusedNormalLogProb._backward_outer(ctx, G, v, l, s, value.shape, loc.shape,
                                                                raw_scale.shape, log_scale))
        g, V, L, S = sub(od, False), sub(vd), sub(ld), sub(sd)
        w = 1.0 / (s * s)
        es = lambda expr, *ops: t.einsum(expr, *ops)
        gv = gl = gs = None
        if ctx.needs_input_grad[1]:
            gv = -(v * es(f"{g},{S}->{V}", G, w) - es(f"{g},{L},{S}->{V}", G, l, w))
            gv = gv.reshape(value.shape)
        if ctx.needs_input_grad[2]:
            gl = es(f"{g},{V},{S}->{L}", G, v, w) - l * es(f"{g},{S}->{L}", G, w)
            gl = gl.reshape(loc.shape)
        if ctx.needs_input_grad[3]:
            q = es(f"{g},{V}->{S}", G, v * v) - 2 * es(f"{g},{V},{L}->{S}", G, v, l) + es(f"{g},{L}->{S}", G, l * l)
            gs = q / (s * s * s) - es(f"{g}->{sub(sd, False)}", G).unsqueeze(-1) / s
            if log_scale:
                gs = gs * s                                  # d/d log(scale)
            gs = gs.reshape(raw_scale.shape)
        return None, gv, gl, gs

    @staticmethod
    def _backward_outer(ctx, G, v, l, s, v_shape, l_shape, s_shape, log_scale):
        """value / loc / scale on pairwise DISJOINT dims (the big K-cross-product factor, e.g. movielens
        z[plate_1,K_z] ~ N(mu_z[K_mu], exp(psi_z)[K_psi])): one permuted copy of G, two GEMMs and a handful of
        passes over [n_value, n_loc, event]-sized tensors -- with d = value - loc, w = 1 / (2 scale^2):
            A = G @ 2w,   d value = -sum_loc d*A,   d loc = sum_value d*A,
            d scale = (2w/scale) * <G, d^2> - <G>/scale          (d log scale = 2w <G, d^2> - <G>)."""
        vd, ld, sd, od, _, _ = ctx.spec
        pos = {id(d): k for k, d in enumerate(od)}
        perm = [pos[id(d)] for d in (*vd, *ld, *sd)]
        nV, nL, nS, E = v.numel() // v.shape[-1], l.numel() // l.shape[-1], s.numel() // s.shape[-1], v.shape[-1]
        Gp = G.permute(*perm).reshape(nV * nL, nS)                      # one copy of G
        v2, l2, s2 = v.reshape(nV, 1, E), l.reshape(1, nL, E), s.reshape(nS, E)
        w2 = 1.0 / (s2 * s2)                                            # = 2w
        gv = gl = gs = None
        need_v, need_l, need_s = ctx.needs_input_grad[1:4]
        D = v2 - l2                                                     # [nV, nL, E]
        if need_v or need_l:
            with _blas("hipblaslt"):
                A = Gp @ w2
            T = D * A.view(nV, nL, E)
            if need_v:
                gv = (-T.sum(1)).reshape(v_shape)
            if need_l:
                from . import engine as E_
                gsum, gd = E_.reduce_factors([(T, ("v", "l", "e"))], plate=("v",))    # two launches (long dim split)
                gl = (gsum if gd == ("l", "e") else gsum.t()).reshape(l_shape)
        if need_s:
            # [nS, E] = Gp^T @ d^2 with K = n_value * n_loc (270,000 at movielens K=30): a batched product over row
            # blocks, then the blocks summed by alan_reduce (no K = 270,000 GEMM, no multi-block torch reduction)
            rows = nV * nL
            blk = next((b for b in (1024, 1000, 900, 512, 500, 256, 250, 128, 100, 64, 50, 32, 30, 25, 16, 10, 8, 5, 4, 3, 2)
                        if rows % b == 0), 1)
            D2 = (D * D).view(rows // blk, blk, E)
            S2 = _sum_leading(t.bmm(Gp.view(rows // blk, blk, nS).transpose(1, 2), D2).view(rows // blk, nS * E)).view(nS, E)
            # <G> over everything but the scale dims: one alan_reduce (which takes few-outputs / huge-reduce sums in
            # two launches), laid out like the scale rows
            from . import engine as E
            sd_ids = {id(d) for d in sd}
            s0, s0d = E.reduce_factors([(G.detach(), od)], plate=tuple(d for d in od if id(d) not in sd_ids))
            S0 = pt_align(PT(s0, s0d), tuple(id(d) for d in sd)).reshape(nS, 1)
            gs = w2 * S2 - S0 if log_scale else (w2 * S2 - S0) / s2
            gs = gs.reshape(s_shape)
        return gv, gl, gs


class _blas:
    """``with _blas("hipblaslt"):`` -- torch's BLAS preference for the enclosed GEMMs only, restored on exit.  Model
    lambdas run under rocBLAS (LAMBDA_BLAS: tiny batched GEMMs, 3.3 us against 9.3); the tall-skinny products of the
    outer-product producer's backward ([270000, 30] x [30, 18]) are the opposite case: 47 us with hipBLASLt, 80 with
    rocBLAS."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        try:
            self.old = t.backends.cuda.preferred_blas_library()
            t.backends.cuda.preferred_blas_library(self.name)
        except Exception:
            self.old = None

    def __exit__(self, *exc):
        if self.old is not None:
            t.backends.cuda.preferred_blas_library(self.old)
        return False


def _sum_leading(x, blk=64):
    """x[n, m].sum(0) as two short reductions (blocks of ``blk`` rows, then the blocks).  A single torch reduction
    over a long leading dim is a multi-block kernel with a semaphore buffer cleared by a memset node; replayed
    from a HIP graph onto an idle GPU that node was observed racing with its neighbours (the gradient of loc
    changed with whether the host synchronised between replays).  Short reductions take the single-block path."""
    n, m = x.shape
    if n <= 4 * blk:
        return x.sum(0)
    main = (n // blk) * blk
    part = x[:main].view(n // blk, blk, m).sum(1)
    if main < n:
        part = t.cat([part, x[main:].sum(0, keepdim=True)])
    return _sum_leading(part, blk)


class _BernoulliLinear(t.autograd.Function):
    """sum log Bernoulli(value; logits = a . b + further terms) with the logits computed by the producer launch itself
    (alan_reduce mode BERNOULLI_LINEAR) and the gradient with respect to ``a`` -- the only differentiable input -- by
    the library's gradient launch (mode BERNOULLI_LINEAR_GRAD): no batched GEMM, no logits tensor, in either direction
    (TorchDimDist.py:127-162 + the model lambda under autograd)."""

    @staticmethod
    def forward(ctx, spec, value, a, *rest):
        from . import engine as E
        vd, term_dims, od, affine = spec
        flat = [a.detach(), *[r.detach() for r in rest]]
        terms, i = [], 0
        for td_ in term_dims:
            terms.append(tuple((flat[i + j], d) for j, d in enumerate(td_)))
            i += len(td_)
        out = E.bernoulli_linear_logprob((value.detach(), vd), terms, od, affine)
        ctx.spec, ctx.terms, ctx.value = spec, terms, value.detach()
        if out is None:
            # (a shape the producer does not take: the same value through torch, laid out like od)
            l, ids = _BernoulliLinear._logits(terms, (value.detach(), vd))
            lp = t.nn.functional.logsigmoid(l) - (1.0 - pt_align(PT(value.detach(), vd), ids)) * l
            keep = {id(d) for d in od}
            red = [k for k, i in enumerate(ids) if i not in keep]
            lp = lp.sum(red) if red else lp
            left = [i for i in ids if i in keep]
            lp = lp.permute(*[left.index(id(d)) for d in od]) if od else lp
            out = t.add(affine[1], lp, alpha=affine[0]) if affine != (1.0, 0.0) else lp
        return out

    @staticmethod
    def _logits(terms, value):
        dims, ids = pt_order([PT(*value), *[PT(x, d) for tm in terms for x, d in tm]])
        l = 0
        for tm in terms:
            if len(tm) == 1:
                l = l + pt_align(PT(tm[0][0], tm[0][1]), ids)
            else:
                l = l + (pt_align(PT(tm[0][0], tm[0][1]), ids) * pt_align(PT(tm[1][0], tm[1][1]), ids)).sum(-1)
        return l, ids

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, G):
        from . import engine as E
        vd, term_dims, od, affine = ctx.spec
        g = E.bernoulli_linear_grad(G, (ctx.value, vd), ctx.terms, od, scale=affine[0])
        if g is None:
            # (shapes the gradient kernel does not take: the same formula through torch)
            (a, ad), (b, bd) = ctx.terms[0]
            l, ids = _BernoulliLinear._logits(ctx.terms, (ctx.value, vd))
            c = pt_align(PT(G, od), ids) * affine[0] * (pt_align(PT(ctx.value, vd), ids) - t.sigmoid(l))
            full = c.unsqueeze(-1) * pt_align(PT(b, bd), ids)
            a_ids = {id(d) for d in ad}
            red = [k for k, i in enumerate(ids) if i not in a_ids]
            full = full.sum(red) if red else full
            keep = [i for i in ids if i in a_ids]
            perm = [keep.index(id(d)) for d in ad]
            g = full.permute(*perm, len(perm)).expand(a.shape).contiguous()
        return (None, None, g, *([None] * (len(ctx.needs_input_grad) - 3)))


class _FusedBernoulliLogProb(t.autograd.Function):
    """sum log Bernoulli(value; logits) over the event dims and every first-class dim missing from the
    output (ONE HIP launch, alan_reduce mode BERNOULLI).  backward: d logits = G * (value - sigmoid(logits))
    summed over the dims logits lacks (plain torch ops on the broadcast, which is only data-sized)."""

    @staticmethod
    def forward(ctx, spec, value, logits):
        from . import engine as E
        vd, ld, od, affine = spec
        out = E.bernoulli_logprob((value.detach(), vd), (logits.detach(), ld), od, affine)
        ctx.spec = spec
        ctx.save_for_backward(value, logits)
        return out

    @staticmethod
    def backward(ctx, G):
        vd, ld, od, affine = ctx.spec
        value, logits = ctx.saved_tensors
        if not ctx.needs_input_grad[2]:
            return None, None, None
        if HIP_PRODUCER_BACKWARD and G.is_cuda:
            from . import engine as E
            from . import native as N
            res = E.producer_grads(G, od, [(value, vd), (logits, ld)], (False, True), (0.0, N.GRAD_LOGITS),
                                   scale=affine[0])
            if res is not None:
                return None, None, res[1]
        if affine[0] != 1.0:
            G = G * affine[0]
        dims, ids = pt_order((PT(value, vd), PT(logits, ld)))
        nv, nl = value.ndim - len(vd), logits.ndim - len(ld)
        nev = max(nv, nl)
        v = pt_align(PT(value.detach(), vd), ids, nev - nv)
        l = pt_align(PT(logits.detach(), ld), ids, nev - nl)
        g = pt_align(PT(G, od), ids)[(...,) + (None,) * nev]
        full = g * (v - t.sigmoid(l))
        full = full.expand([max(a, b) for a, b in zip(full.shape, l.shape)])
        red = [k for k, (a, b) in enumerate(zip(full.shape, l.shape)) if b == 1 and a != 1]
        gl = full.sum(red, keepdim=True) if red else full           # aligned like l
        ld_ids = {id(d) for d in ld}
        drop = [k for k, i in enumerate(ids) if i not in ld_ids] + list(range(len(ids), len(ids) + nev - nl))
        if drop:
            gl = gl.squeeze(drop)
        cur = {i: k for k, i in enumerate(i for i in ids if i in ld_ids)}
        perm = [cur[id(d)] for d in ld]
        gl = gl.permute(*perm, *range(len(ld), gl.ndim))
        return None, None, gl.reshape(logits.shape).to(logits.dtype)


FUSE_NORMAL = True
"""Route Normal / Bernoulli(logits) log-probs on the GPU to the fused HIP producer kernels."""

FUSE_PLATE_STEP = True
"""A Normal factor on disjoint dims (the [plate, K, K, K] tensor of a hierarchical model) goes to the fused
plate-step kernel (alan_normal_lse: producer + log-sum-exp + plate sum in one launch, the factor never materialised;
with gradients, alan_normal_lse_backward) when the plate's contraction has that shape.  Off: the factor is
materialised by the producer kernel and streamed by the HBM-bound reduce_Ks kernel (rows.hip) -- the route every other
shape takes."""

HIP_PRODUCER_BACKWARD = True
"""Gradients of the (small) fused producers by alan_reduce mode PRODUCER_GRAD -- one multi-problem launch per
producer -- instead of a dozen einsum / elementwise torch kernels each."""

OUTER_BACKWARD = True
"""Use the two-GEMM backward of the Normal producer when value / loc / scale carry disjoint dims."""


FUSE_REPARAM = True
"""Reparameterised Normal samples (problem.sample(K, reparam=True)) through one autograd node whose backward is two
small library launches (the sums of G and of G * eps over the broadcast dims, alan_reduce modes SUM / DOT, issued
together) instead of the five or six torch kernels autograd derives from exp, addcmul and the broadcasts."""


def _sum_to(G, other, shape, log_third=None, plus=None, G2=None, noise=None):
    """sum of G (or of G * other, or of G * other * exp(log_third) [+ c * plus[0] with plus = (tensor broadcastable to G,
    c)]) over the dims along which a parameter of shape ``shape`` was broadcast to G's.  ``G2`` (fp32 on the GPU, with
    log_third where other is given): G is G + G2, added inside the launch.  ``noise``: ``other`` is the placeholder of
    noise the launch generates (_NoiseSpec.backward_args)."""
    pad = G.ndim - len(shape)
    keep = [i for i in range(G.ndim) if i >= pad and shape[i - pad] != 1]
    if noise is not None:
        from . import engine as E
        from . import native as N
        axes = tuple(range(G.ndim))
        facs = [(G, axes), (other.expand_as(G), axes)]
        scales = [1.0, 1.0]
        if log_third is not None or plus is not None or G2 is not None:
            facs.append(((log_third if log_third is not None else G).expand_as(G), axes))
            scales.append(2.0 if log_third is not None else 1.0)
            assert log_third is not None
        if plus is not None or G2 is not None:
            facs.append(((plus[0] if plus is not None else G).expand_as(G), axes))
            scales.append(float(plus[1]) if plus is not None else 0.0)
        if G2 is not None:
            facs.append((G2, axes))
            scales.append(1.0)
        with N.may_defer():
            out = E._produce(N.MODE_DOT, facs, tuple(keep), scales=scales, noise=noise)
        if out is not None:
            return out.reshape(shape)
        # (the library does not take this reduction with generated noise -- more than two summed dims that do not merge:
        # the noise is written out by a launch that does, x = 0 + eps * 1 as the forward drew it, and read from memory)
        other = _materialise_noise(other, noise)
    if G2 is not None and other is None:
        from . import engine as E
        from . import native as N
        axes = tuple(range(G.ndim))
        with N.may_defer():
            out = E._produce(N.MODE_SUM, [(G, axes), (G2, axes)], tuple(keep))
        return out.reshape(shape)
    if plus is not None or G2 is not None:
        assert other is not None and log_third is not None
        if G.is_cuda and G.dtype == t.float32 and other.dtype == t.float32 and (plus is None or plus[0].dtype == t.float32):
            from . import engine as E
            from . import native as N
            axes = tuple(range(G.ndim))
            facs = [(G, axes), (other.expand_as(G), axes), (log_third.expand_as(G), axes),
                    ((plus[0] if plus is not None else G).expand_as(G), axes)]
            scales = [1.0, 1.0, 2.0, float(plus[1]) if plus is not None else 0.0]
            if G2 is not None:
                facs.append((G2, axes))
                scales.append(1.0)
            with N.may_defer():
                out = E._produce(N.MODE_DOT, facs, tuple(keep), scales=scales)
            return out.reshape(shape)
        v = (G if G2 is None else G + G2) * other * log_third.exp()
        if plus is not None:
            v = v + plus[1] * plus[0]
        red = [i for i in range(G.ndim) if i not in keep]
        return (v.sum(red) if red else v).reshape(shape)
    if G.is_cuda and G.dtype == t.float32 and (other is None or other.dtype == t.float32) and \
            (len(keep) < G.ndim or log_third is not None):
        from . import engine as E
        from . import native as N
        axes = tuple(range(G.ndim))
        with N.may_defer():
            if other is None:
                out = E._produce(N.MODE_SUM, [(G, axes)], tuple(keep))
            elif log_third is None:
                out = E._produce(N.MODE_DOT, [(G, axes), (other.expand_as(G), axes)], tuple(keep))
            else:
                out = E._produce(N.MODE_DOT, [(G, axes), (other.expand_as(G), axes), (log_third.expand_as(G), axes)],
                                 tuple(keep), scales=[1.0, 1.0, 2.0])
        return out.reshape(shape)
    v = G if other is None else G * other
    if log_third is not None:
        v = v * log_third.exp()
    if len(keep) == G.ndim:
        return v.reshape(shape)
    red = [i for i in range(G.ndim) if i not in keep]
    return v.sum(red).reshape(shape)


class _ReparamNormal(t.autograd.Function):
    """x = loc + eps * scale with eps ~ N(0, 1) drawn inside (TorchDimDist.py: d.rsample(), i.e. torch's
    Normal.rsample).  ``raw`` given: scale = exp(raw), the OptParam(transformation=torch.exp) idiom, and the gradient is
    returned with respect to raw.  backward: d loc = sum_b G, d scale = sum_b G * eps (d raw = that times scale), b = the
    dims the parameter was broadcast along -- two reductions issued as one multi-problem launch."""

    @staticmethod
    def forward(ctx, loc, scale, raw, shape, perm=None):
        """``perm``: the noise is drawn with shape ``shape`` (the order torch's rsample would use, so the particles are
        the same ones) and read through ``.permute(perm)``: the sample is written contiguously in that order."""
        s = raw.exp() if raw is not None else scale
        eps = t.empty(shape, dtype=loc.dtype, device=loc.device).normal_()
        if perm is not None:
            eps = eps.permute(perm)
        ctx.save_for_backward(eps, s)
        ctx.shapes = (tuple(loc.shape), tuple((raw if raw is not None else scale).shape))
        ctx.is_log = raw is not None
        return t.addcmul(loc, eps, s, out=t.empty(eps.shape, dtype=loc.dtype, device=loc.device))

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, G):
        from . import native as N
        eps, s = ctx.saved_tensors
        G = G.contiguous()
        gl = gs = None
        with N.deferring():
            if ctx.needs_input_grad[0]:
                gl = _sum_to(G, None, ctx.shapes[0])
            if ctx.needs_input_grad[2 if ctx.is_log else 1]:
                gs = _sum_to(G, eps, ctx.shapes[1])
        if gs is not None and ctx.is_log:
            gs = gs * s                                       # d/d raw = d/d scale * exp(raw)
        return gl, (None if ctx.is_log else gs), (gs if ctx.is_log else None), None, None


BATCH_DRAWS = os.environ.get("ALAN_AMD_BATCH_DRAWS", "1") != "0"
"""Problem.sample: the Normal draws of the ancestral pass whose parameters do not depend on a sample still waiting are
collected and issued together -- the affine maps x = loc + eps * exp(raw) of all of them as ONE library launch
(alan_reduce_batch of ALAN_MODE_AFFINE problems, written straight in the sample's layout) behind the noise, and, under
autograd, one node whose backward issues every variable's two reductions together (d loc = sum G, d raw = sum G * eps *
exp(raw): no torch multiply, no exp).  A movielens VI iteration: 2 launches for the draws (their noise generated inside:
DEVICE_NOISE) where there were 9, 2 for their backward where there were 6.  False: a draw is issued where the model
meets it."""

BATCH_NOISE = os.environ.get("ALAN_AMD_BATCH_NOISE", "1") != "0"
"""The noise of a batch of draws from ONE normal_ call over one buffer (one kernel instead of one per variable).  The
particles then differ from the ones torch's rsample would draw variable by variable under the same seed (same
distribution); False keeps them identical."""

_DRAW_BATCH = [None]


class _DrawBatch:
    def __init__(self):
        self.jobs = []

    def add(self, la, sa, is_log, full, perm, dims, src, reparam, holder=None):
        shape = tuple(full[i] for i in perm) if perm is not None else tuple(full)
        pt = PendingPT(self, dims, src, shape, la.dtype, la.device)
        self.jobs.append((la, sa, is_log, t.Size(full), perm, pt, reparam, holder))
        return pt

    def flush(self):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        for dev, dt in {(j[0].device, j[0].dtype) for j in jobs}:
            mine = [j for j in jobs if j[0].device == dev and j[0].dtype == dt]
            specs = [None] * len(mine)
            with t.no_grad():
                if BATCH_NOISE:
                    total = sum(j[3].numel() for j in mine)
                    src = _noise_source(dev, dt, total, any(j[6] for j in mine)) if DEVICE_NOISE else None
                    flat = t.empty(total, dtype=dt, device=dev)
                    if src is None:
                        flat.normal_()
                    eps, o = [], 0
                    for k, j in enumerate(mine):
                        eps.append(flat[o:o + j[3].numel()].view(j[3]))
                        if src is not None:                # (flat stays unwritten: a placeholder whose offsets number the noise)
                            specs[k] = _NoiseSpec(src, o)
                        o += j[3].numel()
                else:
                    eps = [t.empty(j[3], dtype=dt, device=dev).normal_() for j in mine]
                eps = [e if j[4] is None else e.permute(j[4]) for e, j in zip(eps, mine)]
            rp = [k for k, j in enumerate(mine) if j[6]]
            outs = [None] * len(mine)
            if rp:
                meta = tuple((mine[k][2], eps[k], mine[k][7], specs[k]) for k in rp)
                res = _ReparamNormalBatch.apply(meta, *[x for k in rp for x in (mine[k][0], mine[k][1])])
                for i, k in enumerate(rp):
                    outs[k] = res[i]
                    if SAMPLE_ALIAS:
                        mine[k][5].x2 = res[len(rp) + i]
            rest = [k for k in range(len(mine)) if not mine[k][6]]
            if rest:
                with t.no_grad():
                    res = _affine_batch([(mine[k][0], mine[k][1], mine[k][2], eps[k], specs[k]) for k in rest])
                for k, x in zip(rest, res):
                    outs[k] = x
            if BATCH_NOISE:
                _release_unused_slot(src)
            for j, x in zip(mine, outs):
                j[5]._val = x


def _affine_batch(jobs):
    """[(loc, scale or log-scale, is_log, eps)] -> [loc + eps * scale], each written contiguously with eps's shape; on the
    GPU all of them as one multi-problem launch (ALAN_MODE_AFFINE)."""
    from . import engine as E
    from . import native as N
    if not all(e.is_cuda and e.dtype == t.float32 and l.dtype == s_.dtype == t.float32 for l, s_, _, e, _ in jobs):
        assert all(z is None for *_, z in jobs)
        return [t.addcmul(l, e, s_.exp() if lg else s_) for l, s_, lg, e, _ in jobs]
    outs = []
    with N.deferring(), N.may_defer():
        for i, (l, s_, lg, e, z) in enumerate(jobs):
            axes = tuple(range(e.ndim))
            args = [(l.expand_as(e), axes), (e, axes), (s_.expand_as(e), axes)]
            # (the last job with generated noise advances a captured graph's counter: the library finds it by itself)
            x = E._produce(N.MODE_AFFINE, args, axes, scales=[1.0, 1.0, 2.0 if lg else 1.0],
                           noise=z.forward_args() if z is not None else None)
            if x is None:                                 # (not a problem the library generates noise for: torch's, in memory)
                z.declined = True
                e.normal_()
                x = E._produce(N.MODE_AFFINE, args, axes, scales=[1.0, 1.0, 2.0 if lg else 1.0])
            elif z is not None:
                z.src.used = True
            outs.append(x)
    N.flush()                                             # (inside someone else's deferring() the exit above issues nothing)
    return outs


DEVICE_NOISE = os.environ.get("ALAN_AMD_DEVICE_NOISE", "1") != "0"
"""The standard-normal noise of a batch of draws generated INSIDE the launch that uses it (alan_noise_t: Philox4x32-10 +
Box-Muller keyed by torch's generator -- its seed, its offset, which the draw advances as torch's own kernels would) and,
for a reparameterised sample, again inside the launch of its gradient: no noise kernel and no noise in memory.  In a
graph captured by GraphedStep / GraphedEval the counter lives on the device (native.GraphNoise): no generator-state fills
in front of a replay either.  Same distribution as torch's normal_, different particles for a given seed (as BATCH_NOISE).
False: the noise is torch's, drawn into memory."""


class _NoiseSource:
    """Where one batch of draws gets its noise: (seed, offset) by value from torch's generator, or a captured graph's
    device-side state."""
    __slots__ = ("seed", "offset", "cell", "receipt", "advance", "advance_by", "state", "used")


def _noise_source(dev, dtype, total, reparam):
    from . import native as N
    if dtype != t.float32 or dev.type != "cuda" or total < 1:
        return None
    inc = 4 * ((total + 3) // 4)
    src = _NoiseSource()
    src.advance_by, src.state, src.used = inc, None, False
    if t.cuda.is_current_stream_capturing():
        st = N.graph_noise(dev)
        if st is None:                                    # (someone else's capture: torch's generator knows how to be captured)
            return None
        slot = st.next_launch()
        if slot is None:
            return None
        src.seed, src.offset, (src.cell, src.advance) = 0, 0, slot
        src.receipt = t.empty(2, dtype=t.int64, device=dev) if reparam else None
        st.per_replay += inc
        src.state = st                                    # (given back by _release_unused_slot if no launch takes it)
        return src
    gen = t.cuda.default_generators[dev.index if dev.index is not None else t.cuda.current_device()]
    src.seed, src.offset, src.cell, src.receipt, src.advance = gen.initial_seed(), gen.get_offset(), None, None, None
    gen.set_offset(src.offset + inc)
    return src


def _release_unused_slot(src):
    """A captured graph's ring slot that NO launch of its batch took (the library declined every job: the draws fell back
    to torch's normal_): given back, so that the ring the capture closes holds only slots some launch reads and writes --
    a reserved slot nobody writes would leave every later drawing launch reading {counter 0, seed 0} on every replay."""
    if src is not None and src.state is not None and not src.used:
        src.state.release_last(src.advance_by)
        src.state = None


class _NoiseSpec:
    """One variable's share of a batch's noise: the elements of its placeholder, numbered from ``base``."""
    __slots__ = ("src", "base", "declined")

    def __init__(self, src, base):
        self.src, self.base, self.declined = src, base, False

    def forward_args(self):
        s = self.src
        return (s.seed, s.offset + self.base, s.cell, s.receipt, s.advance, s.advance_by)

    def backward_args(self):
        s = self.src
        if self.declined:
            return None
        if s.cell is not None:                            # (replayed: the forward's receipt says which numbers it used)
            return (0, self.base, s.receipt, None, None, 0)
        return (s.seed, s.offset + self.base, None, None, None, 0)


def _materialise_noise(placeholder, noise):
    """The noise ``noise`` names, written into its placeholder's memory (a launch x = 0 + eps * 1)."""
    from . import engine as E
    from . import native as N
    axes = tuple(range(placeholder.ndim))
    zero = t.zeros((), dtype=placeholder.dtype, device=placeholder.device)
    one = t.ones((), dtype=placeholder.dtype, device=placeholder.device)
    out = E._produce(N.MODE_AFFINE, [(zero.expand_as(placeholder), axes), (placeholder, axes),
                                     (one.expand_as(placeholder), axes)], axes, scales=[1.0, 1.0, 1.0], noise=noise)
    if out is None:
        raise N.NativeError("alan_amd: the library declined to regenerate a draw's noise (it generated it in the forward)")
    N.flush()
    return out


class _ReparamNormalBatch(t.autograd.Function):
    """x_i = loc_i + eps_i * scale_i for a batch of Normal variables (what torch's Normal.rsample computes one variable
    at a time, TorchDimDist.py:88-125), the noise given: one launch forward, and the backward of all of them -- d loc_i =
    sum_b G_i, d scale_i = sum_b G_i eps_i, or d raw_i = sum_b G_i eps_i exp(raw_i) where the scale is exp(raw_i) -- as
    one multi-problem launch per four reductions."""

    @staticmethod
    def forward(ctx, meta, *params):
        jobs = [(params[2 * i], params[2 * i + 1], meta[i][0], meta[i][1], meta[i][3]) for i in range(len(meta))]
        outs = _affine_batch(jobs)
        ctx.meta = meta
        for m in meta:                                     # (this node will add log q's share of d raw itself: OWN_LOGQ_FOLD)
            if m[2] is not None and m[0]:
                m[2]["node"] = True
        ctx.shapes = [(tuple(l.shape), tuple(s_.shape)) for l, s_, _, _, _ in jobs]
        ctx.save_for_backward(*[j[1] for j in jobs])
        # every sample twice: the second output aliases the first (dims.ReparamPT.x2) -- two consumers, two gradient slots
        return tuple(outs) + tuple(o.view_as(o) for o in outs)

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, *Gs):
        from . import native as N
        grads = [None]
        with N.deferring():
            n = len(ctx.meta)
            for i, ((is_log, eps, holder, spec), G) in enumerate(zip(ctx.meta, Gs[:n])):
                gl = gs = None
                G2 = Gs[n + i]                            # (the gradient that arrived through the sample's second output)
                if G is None:
                    G, G2 = G2, None
                own = holder.pop("own", None) if holder is not None else None
                if G is None and own is not None:         # (nobody but its own log q used the sample)
                    G = t.zeros(eps.shape, dtype=eps.dtype, device=eps.device)
                if G is not None:
                    G = G.contiguous()
                    if G2 is not None:
                        G2 = G2.contiguous()
                        if not (G.is_cuda and G.dtype == G2.dtype == t.float32 and is_log):
                            G, G2 = G + G2, None          # (the launches below take the pair only in this form)
                    if ctx.needs_input_grad[1 + 2 * i]:
                        gl = _sum_to(G, None, ctx.shapes[i][0], G2=G2)
                    if ctx.needs_input_grad[2 + 2 * i]:
                        plus = None
                        if own is not None:
                            # log q's upstream gradient, laid out over the sample's leading dims and broadcast over the
                            # event dims: d raw gets coef * its sum over the same broadcast dims, in the same launch
                            Gq, coef = own
                            plus = (Gq.reshape(tuple(Gq.shape) + (1,) * (G.ndim - Gq.ndim)), coef)
                        gs = _sum_to(G, eps, ctx.shapes[i][1], log_third=ctx.saved_tensors[i] if is_log else None, plus=plus,
                                     G2=G2, noise=spec.backward_args() if spec is not None else None)
                grads += [gl, gs]
        return tuple(grads)


def _tkey(x):
    """Identity of a tensor's contents for as long as nobody writes to it: address, layout, version counter."""
    return (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x._version)


class _OwnSampleLogProb(t.autograd.Function):
    """log N(x; loc, scale) (times a, plus b) for x = loc + eps * scale drawn from THIS distribution (_ReparamNormal):
    as a function of the parameters it is -eps^2 / 2 - log scale - const, so its total derivative is -1 with respect to
    log scale and nothing with respect to loc -- what autograd finds by sending -eps / scale back through x and adding
    the explicit partials (TorchDimDist.py:127-162 under elbo_vi), three producer-gradient launches and as many
    accumulations per variable.  forward: the ordinary producer launch; backward: one small sum of G."""

    @staticmethod
    def forward(ctx, spec, value, loc, scale):
        from . import engine as E
        vd, ld, sd, od, log_scale, affine = spec
        out = E.normal_logprob((value.detach(), vd), (loc.detach(), ld), (scale.detach(), sd), od,
                               log_scale=log_scale, affine=affine)
        ctx.spec = spec
        ctx.save_for_backward(scale)
        return out

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, G):
        from . import engine as E
        from . import native as N
        vd, ld, sd, od, log_scale, affine = ctx.spec
        (scale,) = ctx.saved_tensors
        if not ctx.needs_input_grad[3]:
            return None, None, None, None
        g = E._produce(N.MODE_SUM, [(G.contiguous(), tuple(id(d) for d in od))], tuple(id(d) for d in sd))
        g = g.reshape(tuple(g.shape) + (1,) * (scale.ndim - len(sd))).expand(scale.shape)      # every event element alike
        a = float(affine[0])
        if log_scale:
            return None, None, None, (g if a == -1.0 else g * (-a))
        return None, None, None, g * (-a) / scale


SAMPLE_ALIAS = os.environ.get("ALAN_AMD_SAMPLE_ALIAS", "1") != "0"
"""The node that draws a batch of samples returns every sample twice -- the second output aliases the first -- and the
variable's own log P term takes the second (logpq.logPQ_group): the two gradients a sample receives (from its prior term
and from whatever it parameterises) reach the node in separate slots and are added inside its backward's launches
(two-factor SUM, five-factor DOT) instead of by an add kernel of autograd's: three launches fewer per VI iteration."""

OWN_LOGQ_FOLD = os.environ.get("ALAN_AMD_OWN_FOLD", "1") != "0"
"""The log-scale gradient that a variable's own log q contributes (-1 per unit of upstream gradient, _OwnSampleLogProb) is
handed to the node that drew the sample (_ReparamNormalBatch), whose backward writes the parameter's WHOLE gradient with
one launch (four-factor ALAN_MODE_DOT) -- instead of a sum launch here and an add kernel where autograd joins the two
contributions: two launches fewer per latent variable of a VI iteration."""


class _OwnSampleLogProbFolded(t.autograd.Function):
    """_OwnSampleLogProb whose backward does not return the log-scale's gradient but leaves it with the sample's node
    (``holder``), which runs after this one -- the sample is an input here, so autograd orders them."""

    @staticmethod
    def forward(ctx, spec, value, loc, scale, holder):
        from . import engine as E
        vd, ld, sd, od, log_scale, affine = spec
        out = E.normal_logprob((value.detach(), vd), (loc.detach(), ld), (scale.detach(), sd), od,
                               log_scale=log_scale, affine=affine)
        ctx.spec, ctx.holder = spec, holder
        return out

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, G):
        vd, ld, sd, od, log_scale, affine = ctx.spec
        # G is laid out over od = the sample's leading dims, in the sample's order (checked at the call site)
        prev = ctx.holder.get("own")
        G = G.contiguous()
        ctx.holder["own"] = (G if prev is None else prev[0] + G, -float(affine[0]))
        return None, None, None, None, None


def _peek(p):
    """A tensor with the device, dtype, autograd state and shape of ``p.x`` that does not make a lazy PT evaluate."""
    if isinstance(p, ShiftPT) and not p.materialised:
        return p.rest
    if isinstance(p, ScaledPT) and not p.materialised:
        return _peek(p.src) if p._raw is None else p._raw
    if isinstance(p, ExpPT) and not p.materialised:
        return p.raw
    return p.x


def _lazy_shift(loc):
    """The un-concatenated previous state behind a transition's location (a ShiftPT, or a constant multiple of one), or
    None.  -> (ShiftPT, multiplier)"""
    if isinstance(loc, ShiftPT) and not loc.materialised:
        return loc, 1.0
    if isinstance(loc, ScaledPT) and not loc.materialised and loc._raw is None and isinstance(loc.src, ShiftPT) \
            and not loc.src.materialised:
        return loc.src, loc.mul
    return None


class TorchDimDist:
    """A torch.distributions distribution whose parameters are torchdim tensors (or PTs)."""

    def __init__(self, dist, **kwargs):
        self.dist = dist
        self.kwargs = {k: PT.of(v) for k, v in kwargs.items()}
        self.all_arg_dims, self.all_arg_ids = pt_order(self.kwargs.values())
        self.sample_event_ndim = dist.support.event_dim
        self.arg_event_ndim = {k: _arg_event_ndim(dist, k) for k in self.kwargs}
        self.arg_batch_ndim = {k: v.n_pos - self.arg_event_ndim[k] for k, v in self.kwargs.items()}
        self.sample_batch_ndim = max(self.arg_batch_ndim.values())

    def _build(self, ids, n_sample_pad=0):
        """Instantiate the torch distribution with every parameter laid out
        [ids (1 where absent)..., 1*n_sample_pad, batch (left-padded)..., event...]."""
        args = {}
        for k, v in self.kwargs.items():
            pad = n_sample_pad + self.sample_batch_ndim - self.arg_batch_ndim[k]
            args[k] = pt_align(v, ids, pad)
        return self.dist(**args, validate_args=VALIDATE_ARGS)

    def sample(self, reparam, sample_dims, sample_shape=()):
        """Sample with exactly the dims ``sample_dims`` (a superset of the parameters' dims) and the
        positional shape [*sample_shape, *batch, *event].  Returns a torchdim tensor."""
        return self.sample_pt(reparam, sample_dims, sample_shape).dim()

    def sample_pt(self, reparam, sample_dims, sample_shape=()):
        """As ``sample`` but returns a PT (first-class dims leading: extra dims, then parameter dims)."""
        sample_dims = list(sample_dims)
        ids = [id(d) for d in sample_dims]
        if len(set(ids)) != len(ids):
            raise Exception("Non-unique elements in sample_dims")
        assert set(self.all_arg_ids).issubset(ids)
        if reparam and not self.dist.has_rsample:
            raise Exception(f"Trying to do reparameterised sampling of {self.dist}, which is not implemented "
                            "by PyTorch (likely because it is a distribution over discrete random variables).")
        have = set(self.all_arg_ids)
        extra = [d for d in sample_dims if id(d) not in have]
        shape = t.Size([*sample_shape, *[e.size for e in extra]])
        if reparam and FUSE_REPARAM and self.dist is td.Normal and set(self.kwargs) == {"loc", "scale"} and t.is_grad_enabled():
            loc, scale = self.kwargs["loc"], self.kwargs["scale"]
            lazy = isinstance(scale, ExpPT) and not scale.materialised
            sv = PT(scale.raw, scale.dims) if lazy else scale
            if loc.x.is_cuda and loc.x.dtype == sv.x.dtype == t.float32 and (loc.x.requires_grad or sv.x.requires_grad):
                # one autograd node: exp of the raw scale, the noise and the affine map inside; its backward is two
                # small library reductions (see _ReparamNormal)
                # (the tensors themselves are kept: while the sample lives nothing else can take their addresses)
                src = (_tkey(loc.x), loc.ids, _tkey(sv.x), sv.ids, lazy, loc.x, sv.x, {})
                pl = self.sample_batch_ndim - self.arg_batch_ndim["loc"]
                ps = self.sample_batch_ndim - self.arg_batch_ndim["scale"]
                if not sample_shape:
                    # drawn directly in the caller's dim order (plates outermost, K innermost): no re-layout copy after
                    la, sa = pt_align(loc, ids, pl), pt_align(sv, ids, ps)
                    n = len(ids)
                    rest = t.broadcast_shapes(la.shape[n:], sa.shape[n:])
                    drawn = [*extra, *self.all_arg_dims]                       # the order rsample draws in
                    full = t.Size([*[d.size for d in drawn], *rest])
                    pos = {id(d): k for k, d in enumerate(drawn)}
                    perm = [pos[i] for i in ids] + list(range(n, n + len(rest)))
                    perm = None if perm == list(range(len(perm))) else perm
                    if _DRAW_BATCH[0] is not None:
                        return _DRAW_BATCH[0].add(la, sa, lazy, full, perm, sample_dims, src, True, holder=src[7])
                    x = _ReparamNormal.apply(la, None if lazy else sa, sa if lazy else None, full, perm)
                    return ReparamPT(x, sample_dims, src)
                la, sa = pt_align(loc, self.all_arg_ids, pl), pt_align(sv, self.all_arg_ids, ps)
                full = t.Size([*shape, *t.broadcast_shapes(la.shape, sa.shape)])
                x = _ReparamNormal.apply(la, None if lazy else sa, sa if lazy else None, full)
                ns, nd = len(sample_shape), len(extra) + len(self.all_arg_dims)
                if ns and nd:
                    x = x.permute(*range(ns, ns + nd), *range(ns), *range(ns + nd, x.ndim))
                return ReparamPT(x, (*extra, *self.all_arg_dims), src)
        if self.dist is td.Normal and set(self.kwargs) == {"loc", "scale"} and not sample_shape \
                and not (reparam and t.is_grad_enabled()) \
                and self.kwargs["loc"].x.dtype == _peek(self.kwargs["scale"]).dtype \
                and self.kwargs["loc"].x.device == _peek(self.kwargs["scale"]).device:
            # gradient-free Normal draw, written directly in the caller's dim order (plates outermost, K innermost) from
            # noise drawn in rsample's order -- the same particles, no re-layout copy afterwards
            loc, scale = self.kwargs["loc"], self.kwargs["scale"]
            with t.no_grad():
                la = pt_align(loc, ids, self.sample_batch_ndim - self.arg_batch_ndim["loc"])
                # (a learned log-scale -- OptParam(transformation=exp) -- stays unevaluated for the batched draw: the exp
                # happens in its launch, and the log-prob producers then take the raw parameter too)
                lazy = _DRAW_BATCH[0] is not None and la.is_cuda and la.dtype == t.float32 and isinstance(scale, ExpPT) \
                    and not scale.materialised
                sa = pt_align(PT(scale.raw, scale.dims) if lazy else scale, ids,
                              self.sample_batch_ndim - self.arg_batch_ndim["scale"])
                n = len(ids)
                rest = t.broadcast_shapes(la.shape[n:], sa.shape[n:])
                drawn = [*extra, *self.all_arg_dims]
                pos = {id(d_): k for k, d_ in enumerate(drawn)}
                perm = [pos[i] for i in ids] + list(range(n, n + len(rest)))
                if _DRAW_BATCH[0] is not None and la.is_cuda and la.dtype == t.float32:
                    return _DRAW_BATCH[0].add(la, sa, lazy, [*[d_.size for d_ in drawn], *rest],
                                              None if perm == list(range(len(perm))) else perm, sample_dims, None, False)
                eps = t.empty([*[d_.size for d_ in drawn], *rest], dtype=la.dtype, device=la.device).normal_().permute(perm)
                x = t.addcmul(la, eps, sa, out=t.empty(eps.shape, dtype=la.dtype, device=la.device))
            return PT(x, sample_dims)
        d = self._build(self.all_arg_ids)

        def draw():
            if self.dist is td.Normal:
                # loc + eps * scale as ONE kernel (and one backward node) instead of rsample's two: a training
                # iteration is bound by its count of small launches
                eps = t.empty(d._extended_shape(shape), dtype=d.loc.dtype, device=d.loc.device).normal_()
                return t.addcmul(d.loc, eps, d.scale)
            return d.rsample(shape)

        if reparam:
            x = draw()
        elif self.dist.has_rsample:
            # same distribution as .sample(), but e.g. Normal.sample() calls torch.normal(mean, std), whose
            # std >= 0 check synchronises with the device and cannot be captured into a HIP graph
            with t.no_grad():
                x = draw()
        else:
            x = d.sample(shape)
        ns, nd = len(sample_shape), len(extra) + len(self.all_arg_dims)
        if ns and nd:      # [sample_shape, dims, batch, event] -> [dims, sample_shape, batch, event]
            x = x.permute(*range(ns, ns + nd), *range(ns), *range(ns + nd, x.ndim))
        return PT(x, (*extra, *self.all_arg_dims))

    def log_prob_pt(self, x, dim_order=None, sum_dims=(), affine=None, unevaluated_ok=False):
        """log p(x) as a PT over (x's dims) U (parameter dims); positional sample/batch dims are summed
        out (utils.py:147-152).  ``dim_order = (lead, last)`` fixes the storage order of the result:
        ``lead`` dims outermost, ``last`` dims innermost, any others in between.  ``sum_dims``: first-class
        dims to sum out as well (a data-only plate's sum, logpq.py:149, fused into the producer).
        ``affine = (a, b, allowed ids)``: if every dim of the result is in ``allowed`` the result is
        a * log_prob + b instead (the caller re-checks the same condition on the returned dims)."""
        x = PT.of(x)
        lead, last = dim_order if dim_order is not None else ((), ())
        dims, ids = pt_order((x, *self.kwargs.values()), lead, last)
        drop = {id(d) for d in sum_dims}
        if not drop <= set(ids):
            raise Exception("log_prob: a dim to sum out is on neither the value nor the parameters")
        out_dims = tuple(d for d in dims if id(d) not in drop)
        n_sample = x.n_pos - self.sample_batch_ndim - self.sample_event_ndim
        assert n_sample >= 0
        ab = (1.0, 0.0)
        if affine is not None and {id(d) for d in out_dims} <= set(affine[2]):
            ab = (float(affine[0]), float(affine[1]))
        kind = self._fusable(x)
        if kind == "normal":
            loc, scale = self.kwargs["loc"], self.kwargs["scale"]
            lazy = isinstance(scale, ExpPT) and not scale.materialised
            spec = (x.dims, loc.dims, scale.dims, out_dims, lazy, ab)
            sx = scale.raw if lazy else scale.x
            shift = _lazy_shift(loc)
            if shift is not None and unevaluated_ok and LAZY_TRANSITION and ab == (1.0, 0.0) and not drop \
                    and not (t.is_grad_enabled() and (x.x.requires_grad or sx.requires_grad)) \
                    and x.x.dtype == shift[0].rest.dtype == sx.dtype == t.float32:
                # a timeseries transition whose previous state is still in its two pieces: the factor stays unevaluated,
                # the pieces go to the chain's first round as they are (logpq._chain_of_terms)
                return LazyNormalPT(PT(x.x.detach(), x.dims), shift[0], PT(sx.detach(), scale.dims), lazy, out_dims,
                                    loc_mul=shift[1])
            lx = loc.raw if isinstance(loc, ScaledPT) and not loc.materialised else loc.x
            nograd = not (t.is_grad_enabled() and (x.x.requires_grad or lx.requires_grad or sx.requires_grad))
            vi, li, si = set(x.ids), set(loc.ids), set(scale.ids)
            if FUSE_PLATE_STEP and ab == (1.0, 0.0) and not drop and li and si \
                    and not (vi & li) and not (vi & si) and not (li & si):
                # the big K-cross-product factor: leave it unevaluated -- the plate recursion may fuse it into the
                # log-sum-exp + plate sum that consumes it (logpq._contract: alan_normal_lse, and with gradients to
                # record its one-pass backward); anyone else reading .x gets it made
                if nograd:
                    return LazyNormalPT(PT(x.x.detach(), x.dims), PT(loc.x.detach(), loc.dims),
                                        PT(sx.detach(), scale.dims), lazy, out_dims)
                return LazyNormalPT(PT(x.x, x.dims), PT(loc.x, loc.dims), PT(sx, scale.dims), lazy, out_dims, grad=True)
            if nograd and unevaluated_ok and LAZY_TRANSITION and ab == (1.0, 0.0) and not drop \
                    and x.x.dtype == lx.dtype == sx.dtype == t.float32:
                # asked for by the caller (a timeseries transition, whose consumer is the chain's first round,
                # logpq._chain_of_terms): unevaluated; anyone else reading .x gets it made
                mul = loc.mul if isinstance(loc, ScaledPT) and not loc.materialised else 1.0
                return LazyNormalPT(PT(x.x.detach(), x.dims), PT(lx.detach(), loc.dims), PT(sx.detach(), scale.dims),
                                    lazy, out_dims, loc_mul=mul)
            if nograd:
                from . import engine as E           # nothing to record: skip the autograd.Function round trip
                if isinstance(loc, ScaledPT) and not loc.materialised:      # loc = c * raw: the producer multiplies
                    return PT(E.normal_logprob((x.x.detach(), x.dims), (loc.raw.detach(), loc.dims), (sx.detach(), scale.dims),
                                               out_dims, log_scale=lazy, affine=ab, loc_scale=loc.mul), out_dims)
                return PT(E.normal_logprob((x.x.detach(), x.dims), (loc.x.detach(), loc.dims), (sx.detach(), scale.dims),
                                           out_dims, log_scale=lazy, affine=ab), out_dims)
            if FUSE_REPARAM and isinstance(x, ReparamPT) and not drop \
                    and x.src[:5] == (_tkey(loc.x), loc.ids, _tkey(sx), scale.ids, lazy):
                # x is this distribution's own reparameterised sample: the log-prob's total gradient reaches the
                # (log) scale only
                holder = x.src[7] if len(x.src) > 7 else None
                if OWN_LOGQ_FOLD and lazy and holder is not None and holder.get("node") and x.x.requires_grad \
                        and tuple(id(d_) for d_ in out_dims) == tuple(x.ids) and x.x.dtype == t.float32 \
                        and float(ab[0]) == float(holder.setdefault("coef", float(ab[0]))):
                    return PT(_OwnSampleLogProbFolded.apply(spec, x.x, loc.x, sx, holder), out_dims)
                return PT(_OwnSampleLogProb.apply(spec, x.x, loc.x, sx), out_dims)
            return PT(_FusedNormalLogProb.apply(spec, x.x, loc.x, sx), out_dims)
        if kind == "bernoulli":
            logits = self.kwargs["logits"]
            if isinstance(logits, LinearPT) and not logits.materialised and x.n_pos == 0 \
                    and not (t.is_grad_enabled() and x.x.requires_grad):
                # logits = a sum of arguments and dot products of arguments: the producer computes them itself
                from . import engine as E
                if logits.grad is None:
                    out = E.bernoulli_linear_logprob((x.x, x.dims), _producer_terms(x, logits.terms), out_dims, ab)
                    if out is not None:
                        return PT(out, out_dims)
                else:
                    # one operand carries a gradient: its term first, forward and backward both library launches
                    terms = [logits.terms[logits.grad], *[tm for i, tm in enumerate(logits.terms) if i != logits.grad]]
                    prod = _producer_terms(x, terms)
                    target = terms[0][0]
                    if len(prod[0]) == 2 and set(target.ids) == {id(d) for d in out_dims}:
                        spec = (x.dims, tuple(tuple(d for _, d in tm) for tm in prod), out_dims, ab)
                        flat = [tensor for tm in prod for tensor, _ in tm]
                        out = _BernoulliLinear.apply(spec, x.x, target.x, *flat[1:])
                        if out is not None:
                            return PT(out, out_dims)
            spec = (x.dims, logits.dims, out_dims, ab)
            if not (t.is_grad_enabled() and (x.x.requires_grad or logits.x.requires_grad)):
                from . import engine as E
                return PT(E.bernoulli_logprob((x.x.detach(), x.dims), (logits.x.detach(), logits.dims), out_dims, ab),
                          out_dims)
            return PT(_FusedBernoulliLogProb.apply(spec, x.x, logits.x), out_dims)
        d = self._build(ids, n_sample)
        lp = d.log_prob(pt_align(x, ids))
        axes = [i for i, d_ in enumerate(dims) if id(d_) in drop] + list(range(len(ids), lp.ndim))
        if axes:
            lp = lp.sum(tuple(axes))
        if ab != (1.0, 0.0):
            lp = t.add(ab[1], lp, alpha=ab[0])
        return PT(lp, out_dims)

    def log_prob(self, x, dim_order=None):
        assert is_tensor(x)
        return self.log_prob_pt(x, dim_order).dim()

    def _normal_args(self):
        loc, scale = self.kwargs["loc"], self.kwargs["scale"]
        lazy = isinstance(scale, ExpPT) and not scale.materialised
        return loc, scale, lazy

    @staticmethod
    def log_p_minus_q(P, Q, x, dim_order, allowed, logK):
        """log P(x) - log Q(x) - log K as ONE launch, or None when that does not apply: both must be fusable
        Normals, nothing may need a gradient, and the result may only carry dims in ``allowed`` (the
        group's own K and the active plates -- then reduce_logQ is the identity, Sampler.py:118-134)."""
        x = PT.of(x)
        if P._fusable(x) != "normal" or Q._fusable(x) != "normal":
            return None
        args = [x, *[a for D in (P, Q) for a in D._normal_args()[:2]]]
        if t.is_grad_enabled() and any((a.raw if isinstance(a, ExpPT) and not a.materialised else a.x).requires_grad
                                       for a in args):
            return None
        lead, last = dim_order
        dims, ids = pt_order(args, lead, last)
        if not set(ids) <= set(allowed):
            return None
        from . import engine as E
        terms = []
        for D in (P, Q):
            loc, scale, lazy = D._normal_args()
            terms.append(((loc.x, loc.dims), ((scale.raw if lazy else scale.x), scale.dims), lazy))
        out = E.normal_logprob_pq((x.x, x.dims), terms[0], terms[1], tuple(dims), affine=(1.0, -logK))
        return PT(out, dims)

    def _fusable(self, x):
        """Normal and Bernoulli(logits) log-probs on the GPU go to the fused HIP producers (alan_reduce
        modes NORMAL / BERNOULLI); everything else stays on torch.distributions.  -> "normal" | "bernoulli" | None"""
        if not FUSE_NORMAL:
            return None
        if self.dist is td.Normal and set(self.kwargs) == {"loc", "scale"}:
            kind, args = "normal", (x, self.kwargs["loc"], self.kwargs["scale"])
            sc = args[2]
            if isinstance(sc, ExpPT) and not sc.materialised:        # look at the raw parameter instead
                args = (x, args[1], PT(sc.raw, sc.dims))
            if _lazy_shift(args[1]) is not None:                     # (an un-concatenated previous state: look at its series)
                args = (x, PT(_peek(args[1]), args[1].dims), args[2])
            elif isinstance(args[1], ScaledPT) and not args[1].materialised:
                args = (x, PT(args[1].raw, args[1].dims), args[2])
        elif self.dist is td.Bernoulli and set(self.kwargs) == {"logits"}:
            kind, args = "bernoulli", (x, self.kwargs["logits"])
            lg = args[1]
            if isinstance(lg, LinearPT) and not lg.materialised:     # look at its operands instead (fp32, on the GPU)
                args = (x, *[p for term in lg.terms for p in term])
        else:
            return None
        ts = [p.x for p in args]
        if not all(v.is_cuda and v.dtype in (t.float32, t.float64) for v in ts):
            return None
        if kind == "normal" and t.is_grad_enabled() and any(v.requires_grad for v in ts):
            # the backward (einsum contractions) needs one common event shape
            ev = [tuple(v.shape[len(p.dims):]) for v, p in zip(ts, args)]
            return kind if ev[0] == ev[1] == ev[2] else None
        return kind


# --------------------------------------------------------------------------------------------
def function_arguments(f):
    """Argument names of a model lambda; alan insists on plain positional signatures (utils.py:17-43)."""
    spec = inspect.getfullargspec(f)
    if spec.varargs is not None:
        raise Exception("In Alan, functions may not have *args")
    if spec.varkw is not None:
        raise Exception("In Alan, functions may not have **kwargs")
    if spec.defaults is not None or spec.kwonlydefaults is not None:
        raise Exception("In Alan, functions may not have defaults")
    if spec.kwonlyargs:
        raise Exception("In Alan, functions may not have keyword only arguments")
    if spec.annotations:
        raise Exception("In Alan, functions may not have type annotations")
    return spec.args


class Param:
    """Marker base for parameters declared inline in a distribution."""


def _as_init(init):
    if isinstance(init, Number):
        init = t.tensor(float(init))
    assert isinstance(init, t.Tensor)
    return init


class OptParam(Param):
    """A parameter learned by gradient descent (Param.py:18-25)."""

    def __init__(self, init, transformation=None, ignore_platenames=(), name=None):
        self.init = _as_init(init)
        self.trans = transformation if transformation is not None else (lambda x: x)
        self.ignore_platenames = ignore_platenames
        self.name = name


class QEMParam(Param):
    """A parameter of a QEM-updated distribution (Param.py:27-32).  alan_amd keeps the parameter
    (so sampling / log-probs work) but does not implement the QEM optimiser (out of scope)."""

    def __init__(self, init, ignore_platenames=(), name=None):
        self.init = _as_init(init)
        self.trans = lambda x: x
        self.ignore_platenames = ignore_platenames
        self.name = name


class Dist(nn.Module):
    """A named random variable: torch distribution class + how to get each argument from scope."""

    is_timeseries = False

    def __init__(self, varname, dist, args, kwargs, sample_shape=t.Size([])):
        super().__init__()
        self.varname = varname
        self.dist = dist
        self.sample_shape = t.Size(sample_shape)
        self.register_buffer("_device_tensor", t.zeros(()))
        bound = inspect.signature(dist).bind(*args, **kwargs).arguments

        n_qem = sum(isinstance(v, QEMParam) for v in bound.values())
        n_opt = sum(isinstance(v, OptParam) for v in bound.values())
        self.qem_dist, self.opt_dist = n_qem > 0, n_opt > 0
        if (self.qem_dist or self.opt_dist) and len(self.sample_shape) > 0:
            raise Exception("You can't use sample_shape with QEM or Opt parameters")
        if self.qem_dist and n_qem != len(bound):
            raise Exception("If one parameter on a distribution is a QEMParam, then all parameters on that "
                            "distribution should be QEM distributions")

        self.opt_qem_params = {}          # paramname -> (distargname, Param)
        self._names, self._funcs, self._consts = {}, {}, {}
        used = []
        for argname, v in bound.items():
            if isinstance(v, Param):
                if varname is None:
                    raise Exception("You can't use QEMParam / OptParam in a timeseries at present")
                pname = v.name if v.name is not None else f"{varname}_{argname}"
                self.opt_qem_params[pname] = (argname, v)
                v = pname
            if isinstance(v, str):
                self._names[argname] = v
                used.append(v)
            elif isinstance(v, types.FunctionType):
                self._funcs[argname] = (v, function_arguments(v))
                used.extend(self._funcs[argname][1])
            elif isinstance(v, t.Tensor):
                self.register_buffer(f"_const_{argname}", v.rename(None))
                self._consts[argname] = None
            else:
                assert isinstance(v, Number), f"unsupported argument {v!r} for {varname}"
                val = v if _arg_is_discrete(dist, argname) else float(v)
                self.register_buffer(f"_const_{argname}", t.tensor(val))
                self._consts[argname] = None
        self.all_args = list(dict.fromkeys(used))

    @property
    def device(self):
        return self._device_tensor.device

    def tdd(self, scope, dimcache=None):
        """Bind the distribution's arguments from ``scope`` (values: PT or torchdim tensors).  Model
        lambdas receive torchdim tensors; ``dimcache`` memoises the PT -> torchdim wrapping per eval."""
        kw = {}
        for a in self._consts:
            kw[a] = PT(getattr(self, f"_const_{a}"), ())
        for a, name in self._names.items():
            kw[a] = scope[name]
        for a, (fn, argnames) in self._funcs.items():
            kw[a] = call_model_lambda(fn, [(n, scope[n]) for n in argnames], dimcache)
        return TorchDimDist(self.dist, **kw)

    def sample(self, scope, reparam, active_platedims, K_dim, timeseries_perm=None, dimcache=None):
        """-> PT with dims [active plates..., K_dim], stored contiguously in that order (so a Split
        chunk of the plate is a contiguous slice).  scope values may be PTs or torchdim tensors."""
        want = [*active_platedims, K_dim]
        p = self.tdd(scope, dimcache).sample_pt(reparam, want, self.sample_shape)
        ids = tuple(id(d) for d in want)
        if p.ids != ids:
            x = pt_align(p, ids).contiguous()
            p = ReparamPT(x, want, p.src) if isinstance(p, ReparamPT) else PT(x, want)
        return p

    def log_prob(self, x, scope, T_dim=None, K_dim=None, dim_order=None, dimcache=None, sum_dims=(), affine=None,
                 unevaluated_ok=False):
        """-> (PT, None)   [the None mirrors Timeseries.log_prob's K_init slot]"""
        return self.tdd(scope, dimcache).log_prob_pt(x, dim_order=dim_order, sum_dims=sum_dims, affine=affine,
                                                     unevaluated_ok=unevaluated_ok), None


LAMBDA_BACKEND = "vmap"
"""How model lambdas (``lambda z, x: z @ x``) see their arguments.

"torchdim": functorch.dim tensors, as in the reference.  "vmap" (default): the same semantics --
the lambda sees only the POSITIONAL dims of each argument, every first-class dim is mapped over --
implemented as nested ``torch.vmap`` with an explicit nesting order (dims shared by most arguments
outermost).  Besides skipping torchdim's Python dispatch, this makes the lowering independent of
Dim creation order: under a Split, torchdim lowers movielens' ``z @ x`` to 19,000 separate 1x18
dot products (300 us at K=100), nested vmap to one batched GEMM (10 us)."""


_PLAIN_EXP = {}      # code object -> (code object, bool)


def _is_plain_exp(fn):
    """Is ``fn`` exactly ``lambda v: v.exp()`` / ``torch.exp(v)`` (the usual way to write a positive scale,
    e.g. movielens' ``lambda psi_z: psi_z.exp()``)?  Decided once per code object by symbolic tracing."""
    code = getattr(fn, "__code__", None)
    if code is None or code.co_argcount != 1 or getattr(fn, "__closure__", None):
        return False
    hit = _PLAIN_EXP.get(id(code))
    if hit is not None and hit[0] is code:
        return hit[1]
    ok = False
    try:
        import torch.fx
        nodes = list(torch.fx.symbolic_trace(fn).graph.nodes)
        if len(nodes) == 3 and nodes[0].op == "placeholder" and nodes[2].op == "output":
            n = nodes[1]
            is_exp = (n.op == "call_method" and n.target == "exp") or (n.op == "call_function" and n.target is t.exp)
            ok = bool(is_exp and tuple(n.args) == (nodes[0],) and not n.kwargs and nodes[2].args == (n,))
    except Exception:
        ok = False
    _PLAIN_EXP[id(code)] = (code, ok)
    return ok


LAZY_TRANSITION = True
"""The Normal transition factor of a timeseries stays unevaluated on gradient-free evaluations (dims.LazyNormalPT): the
chain's first round computes it on load (alan_chain_logmmexp_terms_normal), the [T, K_init, K] tensor is never written."""

LAZY_SCALED = True
"""``lambda v: c * v`` stays unevaluated where no gradient is wanted (dims.ScaledPT); a fused Normal producer folds the
constant into its location operand."""

_SCALED = {}         # id(code object) -> (code object, constant or None)


def _scaled_form(fn):
    """The constant c when ``fn`` is exactly ``lambda v: c * v`` / ``v * c`` with a numeric literal c, else None.
    Decided once per code object by symbolic tracing."""
    code = getattr(fn, "__code__", None)
    if code is None or code.co_argcount != 1 or getattr(fn, "__closure__", None) or code.co_names:
        return None                    # (co_names: a global the constant could come from -- and change under us)
    hit = _SCALED.get(id(code))
    if hit is not None and hit[0] is code:
        return hit[1]
    c = None
    try:
        import operator
        import torch.fx
        nodes = list(torch.fx.symbolic_trace(fn).graph.nodes)
        if len(nodes) == 3 and nodes[0].op == "placeholder" and nodes[2].op == "output" and nodes[2].args == (nodes[1],):
            n = nodes[1]
            is_mul = (n.op == "call_function" and n.target in (operator.mul, t.mul)) or \
                     (n.op == "call_method" and n.target == "mul")
            if is_mul and not n.kwargs and len(n.args) == 2:
                others = [a for a in n.args if a is not nodes[0]]
                if len(others) == 1 and type(others[0]) in (int, float):
                    c = float(others[0])
    except Exception:
        c = None
    _SCALED[id(code)] = (code, c)
    return c


LINEAR_LOGITS_GRAD = True
"""The same where ONE operand of the lambda carries a gradient (elbo_vi): the forward is the producer launch, the backward
the library's gradient launch (dist._BernoulliLinear) instead of torch's batched GEMMs and elementwise kernels."""

LINEAR_LOGITS = True
"""A model lambda that is a sum of its arguments and of dot products of them (``z @ x``,
``alpha + phi @ bus_company_name + psi @ run_type``) stays unevaluated where no gradient is wanted (dims.LinearPT): a
Bernoulli(logits=...) log-prob then computes the logits inside its producer launch (alan_reduce mode BERNOULLI_LINEAR)."""

_LINEAR_FORM = {}    # id(code object) -> (code object, terms or None)


def _linear_form(fn):
    """``fn``'s value as a sum of terms, each ("arg", i) or ("dot", i, j) over its positional arguments -- or None when
    it is anything else.  Decided once per code object by symbolic tracing (only ``+`` and ``@`` / torch.matmul between
    arguments are recognised; constants, closures, other operators: None)."""
    code = getattr(fn, "__code__", None)
    if code is None or code.co_argcount < 2 or getattr(fn, "__closure__", None):
        return None
    hit = _LINEAR_FORM.get(id(code))
    if hit is not None and hit[0] is code:
        return hit[1]
    terms = None
    try:
        import operator
        import torch.fx
        nodes = list(torch.fx.symbolic_trace(fn).graph.nodes)
        arg = {n: i for i, n in enumerate(n for n in nodes if n.op == "placeholder")}

        def expr(n):
            if n in arg:
                return [("arg", arg[n])]
            if not isinstance(n, torch.fx.Node) or n.kwargs or len(n.args) != 2:
                raise ValueError
            fnc, meth = n.op == "call_function", n.op == "call_method"
            if (fnc and n.target in (operator.add, t.add)) or (meth and n.target == "add"):
                return expr(n.args[0]) + expr(n.args[1])
            if (fnc and n.target in (operator.matmul, t.matmul)) or (meth and n.target == "matmul"):
                a, b = n.args
                if a in arg and b in arg:
                    return [("dot", arg[a], arg[b])]
            raise ValueError

        out = nodes[-1]
        if out.op == "output" and len(out.args) == 1 and isinstance(out.args[0], torch.fx.Node):
            found = expr(out.args[0])
            if any(k[0] == "dot" for k in found):
                terms = tuple(found)
    except Exception:
        terms = None
    _LINEAR_FORM[id(code)] = (code, terms)
    return terms


def _linear_pt(fn, named_args, form, dimcache):
    """The LinearPT of ``fn(*args)`` when the arguments have the shapes its form needs (dot operands: exactly one
    positional dim of the same length; plain summands: none), else None."""
    vals = [v for _, v in named_args]
    terms = []
    for k in form:
        ops = tuple(vals[i] for i in k[1:])
        if k[0] == "dot":
            a, b = ops
            if a.n_pos != 1 or b.n_pos != 1 or a.x.shape[-1] != b.x.shape[-1]:
                return None
        elif ops[0].n_pos != 0:
            return None
        terms.append(ops)
    used = {i for k in form for i in k[1:]}
    if used != set(range(len(vals))):
        return None                               # (an unused argument still contributes its dims: leave that to torch)
    grad = None
    for i, v in enumerate(vals):
        if not (v.x.is_cuda and v.x.dtype == t.float32):
            return None
        if t.is_grad_enabled() and v.x.requires_grad:
            # one operand may be attached to the autograd graph: the first operand of a dot term, used once (movielens'
            # z in `z @ x` under elbo_vi) -- its gradient is the library's own launch (mode BERNOULLI_LINEAR_GRAD)
            where = [ti for ti, k in enumerate(form) if k[0] == "dot" and k[1] == i]
            uses = sum(k[1:].count(i) for k in form)
            if not LINEAR_LOGITS_GRAD or grad is not None or len(where) != 1 or uses != 1:
                return None
            grad = where[0]
    seen, count = {}, {}
    for p in vals:
        for d, i in zip(p.dims, p.ids):
            seen.setdefault(i, d)
            count[i] = count.get(i, 0) + 1
    order = sorted(seen, key=lambda i: -count[i])                  # as _call_lambda_vmap lays its result out
    return LinearPT(terms, [seen[i] for i in order], lambda: _call_lambda_vmap(fn, named_args, dimcache).x, grad=grad)


def _dot_pt(a, b):
    """sum over the single positional dim of a * b, broadcast over first-class dims: one launch of the library's own
    (mode DOT -- it may sit in the queue of small producer launches), or a (batched) GEMM for dtypes it does not take."""
    dims, ids = pt_order((a, b))
    n_out = math.prod(d.size for d in dims)
    # (small ones ride in a queued multi-problem launch; a big one is a GEMM's job: 90,000 outputs take the library's
    # generic kernel 9.4 us against rocBLAS's 4.7)
    if a.x.dtype == t.float32 and b.x.dtype == t.float32 and n_out <= 32768:
        from . import engine as E
        from . import native as N
        with N.may_defer():
            return PT(E.dot_sum((a.x, a.dims), (b.x, b.dims), dims), dims)
    letters = {}
    sub = lambda p: "".join(letters.setdefault(i, chr(ord("a") + len(letters))) for i in p.ids)
    sa, sb = sub(a), sub(b)
    expr = f"{sa}Z,{sb}Z->{''.join(letters[i] for i in ids)}"
    from . import native as N_
    N_.trace("lambda", f"a dot term of a linear-logits lambda with {n_out} outputs (its operands lack some dim of the likelihood's "
             "index space, so it is evaluated once instead of inside the producer)", torch=True,
             route="a (batched) GEMM of torch's (rocBLAS): the library's own dot launch takes up to 32768 outputs")
    if LAMBDA_BLAS is None:
        return PT(t.einsum(expr, a.x, b.x), dims)
    with _blas(LAMBDA_BLAS):                          # (the lambda's own product: its backend, see LAMBDA_BLAS)
        return PT(t.einsum(expr, a.x, b.x), dims)


def _producer_terms(value, terms):
    """The (tensor, dims) terms handed to engine.bernoulli_linear_logprob.  The producer computes a dot product once
    per element of the WHOLE index space; a product whose operands lack some of the dims (bus_breakdown's
    ``phi @ bus_company_name`` has no K_alpha) would be recomputed for every index of those -- such a term is evaluated
    once, the usual way (a batched GEMM), and enters as a plain summand: the launch still saves the adds."""
    sizes = {}
    for p in (value, *[p for term in terms for p in term]):
        for d, i in zip(p.dims, p.ids):
            sizes[i] = d.size
    total = math.prod(sizes.values())
    out, flush = [], False
    for term in terms:
        if len(term) == 2:
            own = {i for p in term for i in p.ids}
            if math.prod(sizes[i] for i in own) < total:
                term = (_dot_pt(*term),)
                flush = True
        out.append(tuple((p.x, p.dims) for p in term))
    if flush:
        from . import native as N
        N.flush()              # the producer launch that reads these terms may be queued too: they go out first
    return out


def _lambda_name(fn, named_args):
    try:
        src = inspect.getsource(fn).strip().replace("\n", " ")
        src = src[src.index("lambda"):][:90] if "lambda" in src else getattr(fn, "__name__", "lambda")
    except Exception:
        src = getattr(fn, "__name__", "lambda")
    return f"{src}   (arguments: {', '.join(n for n, _ in named_args)})"


def call_model_lambda(fn, named_args, dimcache=None):
    from . import native as N_
    vals = [v for _, v in named_args]
    if len(vals) == 1 and type(vals[0]) in (PT, ReparamPT) and vals[0].x.is_floating_point() and _is_plain_exp(fn):
        # exp of one variable: keep it lazy (dims.ExpPT) -- a fused Normal producer then takes the log-scale as it is
        # (alan_reduce mode NORMAL_LOGSCALE) and the exp launch never happens; anyone else reading .x gets exp(raw)
        N_.trace("lambda", _lambda_name(fn, named_args), route="exp of one variable: left unevaluated, the consumer's launch takes the log-scale")
        return ExpPT(vals[0].x, vals[0].dims)
    if LAZY_SCALED and len(vals) == 1 and isinstance(vals[0], ShiftPT) and not vals[0].materialised \
            and vals[0].rest.dtype == t.float32:
        c = _scaled_form(fn)
        if c is not None:                         # c * (the un-concatenated previous state): both stay as they are
            N_.trace("lambda", _lambda_name(fn, named_args), route=f"{c} x the previous state: left unevaluated, the chain's first round multiplies")
            return ScaledPT(None, c, vals[0].dims, src=vals[0])
    if LAZY_SCALED and len(vals) == 1 and type(vals[0]) is PT and vals[0].x.is_cuda and vals[0].x.dtype == t.float32 \
            and not (t.is_grad_enabled() and vals[0].x.requires_grad):
        c = _scaled_form(fn)
        if c is not None:
            # a constant multiple of one variable: lazy (dims.ScaledPT) -- a fused Normal producer takes it as the
            # location's scale field and the multiply launch never happens; anyone else reading .x gets c * v
            N_.trace("lambda", _lambda_name(fn, named_args), route=f"{c} x one variable: left unevaluated, the Normal producer multiplies its location")
            return ScaledPT(vals[0].x, c, vals[0].dims)
    if LINEAR_LOGITS and LAMBDA_BACKEND == "vmap" and len(vals) >= 2 and all(type(v) in (PT, ReparamPT) for v in vals):
        form = _linear_form(fn)
        if form is not None:
            lin = _linear_pt(fn, named_args, form, dimcache)
            if lin is not None:
                N_.trace("lambda", _lambda_name(fn, named_args), route="sum of arguments / dot products: left unevaluated, the Bernoulli "
                         "producer computes the logits in its launch (a gradient-carrying use evaluates it: library launches too)")
                return lin
    N_.trace("lambda", _lambda_name(fn, named_args), route="RUNS AS WRITTEN through torch (vmap over the K / plate dims): its kernels are "
             "torch's, so a captured evaluation replays as a HIP graph, not from the library's launch list", torch=True)
    return _call_lambda_vmap(fn, named_args, dimcache)


LAMBDA_BLAS = None if os.environ.get("ALAN_AMD_KEEP_BLAS") == "1" else "hipblas"
"""torch's BLAS backend WHILE A MODEL LAMBDA RUNS (None: torch's own choice).  A lambda such as movielens' ``z @ x`` that
runs as written is a tiny batched GEMM ([300,30,18] x [300,18,5]): torch's default backend on ROCm (hipBLASLt) takes 9.3 us
for it, rocBLAS 3.3 us (tools/small_bmm_probe.py).  Scoped to the lambda's call -- the host application's other GEMMs
keep torch's preference (until round 3 the package set it process-wide at import)."""


def _call_lambda_vmap(fn, named_args, dimcache=None):
    if LAMBDA_BLAS is None or not t.cuda.is_available():
        return _call_lambda_vmap_(fn, named_args, dimcache)
    with _blas(LAMBDA_BLAS):
        return _call_lambda_vmap_(fn, named_args, dimcache)


def _call_lambda_vmap_(fn, named_args, dimcache=None):
    vals = [v for _, v in named_args]
    if LAMBDA_BACKEND != "vmap" or not all(isinstance(v, PT) for v in vals):
        val = fn(*[_as_dim(v, n, dimcache) for n, v in named_args])
        if not is_tensor(val):
            raise Exception("Lambda on a distribution returned a non-Tensor")
        return val
    # Within ONE evaluation the same lambda on the same argument objects (a parent-level variable seen again by
    # every chunk of a Split) is evaluated once.  Only without autograd: under torch.utils.checkpoint the
    # forward and the recomputation must build identical graphs.
    memo = key = None
    if dimcache is not None and not t.is_grad_enabled():
        memo = dimcache.setdefault("__lambda__", {})
        key = (id(fn), *[id(v) for v in vals])
        hit = memo.get(key)
        if hit is not None and hit[0] is fn and all(a is b for a, b in zip(hit[1], vals)):
            return hit[2]
    seen, count = {}, {}
    for p in vals:
        for d, i in zip(p.dims, p.ids):
            seen.setdefault(i, d)
            count[i] = count.get(i, 0) + 1
    order = sorted(seen, key=lambda i: -count[i])                  # stable: ties keep first appearance
    args = []
    for p in vals:                                                 # present dims leading, in nesting order
        pos = {i: k for k, i in enumerate(p.ids)}
        perm = [pos[i] for i in order if i in pos]
        x = p.x
        if perm != list(range(len(perm))):
            x = x.permute(*perm, *range(len(perm), x.ndim))
        args.append(x)
    val = _nested_vmap(fn, args, [p.ids for p in vals], order, {i: d.size for i, d in seen.items()})
    if not isinstance(val, t.Tensor):
        raise Exception("Lambda on a distribution returned a non-Tensor")
    out = PT(val, [seen[i] for i in order])
    if memo is not None:
        memo[key] = (fn, vals, out)
    return out


try:    # the functorch primitives torch.vmap is built from: 5-10 us per level instead of ~100 us of checks and pytrees
    from torch._C._functorch import (_add_batch_dim, _remove_batch_dim, _vmap_decrement_nesting,
                                     _vmap_increment_nesting)
except ImportError:                                                        # pragma: no cover
    _add_batch_dim = None


def _nested_vmap(fn, args, arg_ids, order, sizes):
    """fn mapped over the dims ``order`` (outermost first); every arg carries its present dims leading, in that
    order.  Same semantics as nested ``torch.vmap(..., in_dims=0/None, out_dims=0, randomness="error")``."""
    if _add_batch_dim is None:
        f = fn
        for i in reversed(order):
            f = t.vmap(f, in_dims=tuple(0 if i in ids else None for ids in arg_ids))
        return f(*args)
    args = list(args)
    levels = []
    try:
        for i in order:
            lvl = _vmap_increment_nesting(sizes[i], "error")
            levels.append((lvl, sizes[i]))
            for k, ids in enumerate(arg_ids):
                if i in ids:
                    args[k] = _add_batch_dim(args[k], 0, lvl)
        out = fn(*args)
        if not isinstance(out, t.Tensor):
            raise Exception("Lambda on a distribution returned a non-Tensor")
        while levels:
            lvl, size = levels.pop()
            out = _remove_batch_dim(out, lvl, size, 0)
            _vmap_decrement_nesting()
        return out
    finally:
        for _ in levels:                    # only non-empty when fn raised
            _vmap_decrement_nesting()


def _as_dim(v, name, cache):
    if not isinstance(v, PT):
        return v
    if cache is None:
        return v.dim()
    key = (name, id(v))
    hit = cache.get(key)
    if hit is None or hit[0] is not v:       # id() can be recycled once a PT dies: check identity
        hit = (v, v.dim())
        cache[key] = hit
    return hit[1]


class _DistSpec:
    """What the user writes, e.g. ``Normal('a', 1.)``; becomes a Dist once its variable name is known."""
    dist = None
    nargs = None

    def __init__(self, *args, sample_shape=t.Size([]), **kwargs):
        if len(args) + len(kwargs) != self.nargs:
            raise Exception(f"Wrong number of arguments provided to {type(self)}")
        self.args, self.kwargs, self.sample_shape = args, kwargs, sample_shape

    def finalize(self, varname):
        return Dist(varname, self.dist, self.args, self.kwargs, self.sample_shape)


_REGISTRY = {
    "Bernoulli": 1, "Beta": 2, "Binomial": 2, "Categorical": 1, "Cauchy": 2, "Chi2": 1,
    "ContinuousBernoulli": 1, "Dirichlet": 1, "Exponential": 1, "FisherSnedecor": 2, "Gamma": 2,
    "Geometric": 1, "Gumbel": 2, "HalfCauchy": 1, "HalfNormal": 1, "Kumaraswamy": 2, "LKJCholesky": 2,
    "Laplace": 2, "LogNormal": 2, "LowRankMultivariateNormal": 3, "Multinomial": 2,
    "MultivariateNormal": 2, "NegativeBinomial": 2, "Normal": 2, "OneHotCategorical": 1, "Pareto": 2,
    "Poisson": 1, "RelaxedBernoulli": 2, "LogitRelaxedBernoulli": 2, "RelaxedOneHotCategorical": 2,
    "StudentT": 3, "Uniform": 2, "VonMises": 2, "Weibull": 2, "Wishart": 2,
}

__all__ = ["TorchDimDist", "Dist", "OptParam", "QEMParam", "new_dist"]


def new_dist(name, dist, nargs):
    """Register a distribution class under ``alan_amd.<name>`` (dist.py:357-366)."""
    cls = type(name, (_DistSpec,), {"dist": dist, "nargs": nargs})
    globals()[name] = cls
    if name not in __all__:
        __all__.append(name)
    return cls


for _name, _nargs in _REGISTRY.items():
    if hasattr(td, _name):
        new_dist(_name, getattr(td, _name), _nargs)
