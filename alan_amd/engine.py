"""
Positional-tensor front end of the HIP contraction engine.

A *factor* here is ``(tensor, dims)``: a plain (positional) device tensor plus one hashable key
per positional dim (functorch ``Dim`` objects in production, strings in tests).  This layer
  * builds ``alan_reduce`` descriptors (include/alan_mi355.h) from factors,
  * wraps each launch in a ``torch.autograd.Function`` whose backward is again ``alan_reduce``
    (mode WEXPSUM: grad_f = sum_{dims not in f} grad_out * exp(sum_f lp_f - lse)), which is what
    autograd derives from the reference's op sequence (utils.py:218-220) up to the eps term,
  * plans the order in which K dims are eliminated (replaces opt_einsum.contract_path,
    reduce_Ks.py:264-265, which only ever chose an order).
"""
import contextlib
import math

import torch as t

from . import native as N
from .dims import Dim


# --------------------------------------------------------------------------- descriptor building
def _space(factors, extra=()):
    """Ordered union of dim keys with sizes."""
    sizes = {}
    for x, dims in list(factors) + list(extra):
        if x.ndim != len(dims):
            raise AssertionError(f"factor has {x.ndim} positional dims but {len(dims)} dim keys")
        for d, s in zip(dims, x.shape):
            if sizes.setdefault(d, s) != s:
                raise Exception(f"size mismatch for dim {d}: {sizes[d]} vs {s}")
    return sizes


def _strides(x, dims, space):
    """Element stride of ``x`` along each dim of the space (0 where x lacks the dim or it has size 1)."""
    st = {d: (s if n > 1 else 0) for d, s, n in zip(dims, x.stride(), x.shape)}
    return [st.get(d, 0) for d in space]


def _out_order(factors, keep, sizes):
    """Keep dims ordered like the largest factor stores them (so the kernel's stores coalesce)."""
    dom = max(factors, key=lambda f: f[0].numel())
    st = dict(zip(dom[1], dom[0].stride()))
    big = 1 << 62
    return sorted(keep, key=lambda d: -(st.get(d, big) or big))


def _use_count(x):
    return t._C._storage_Use_Count(x.untyped_storage()._cdata)


class ResultRing:
    """Where the LAST launch of a graphed evaluation delivers its scalar (alan_reduce_desc_t.ring_*): SLOTS one-element
    tensors, each with its own storage, whose addresses sit in a device table; a device counter says which one the next
    replay writes and the launch itself advances it.  A replayed evaluation can then hand its caller the slot tensor
    instead of a copy of a fixed output buffer (one copy kernel, 4.6 us of a 42 us evaluation at the headline config).
    A slot comes round again after SLOTS replays: if anything still references its storage by then (the result, a view
    or a detach() of it), that storage is left to its holders and the slot gets a fresh tensor."""
    SLOTS = 64

    def __init__(self, device):
        self.device = device
        self.n = self.SLOTS
        self.slots = [t.empty((), dtype=t.float32, device=device) for _ in range(self.n)]
        self.table = t.tensor([s.data_ptr() for s in self.slots], dtype=t.int64, device=device)
        self.counter = t.zeros((), dtype=t.int32, device=device)
        self.placeholder = t.empty((), dtype=t.float32, device=device)     # stands for "the current slot" while tracing
        self.idle = _use_count(self.slots[0])      # references to a slot's storage when only the ring holds it
        self.pos = 0
        self.taken = 0          # launches that wrote through the ring since reset()
        self.declined = False   # the library refused the shape: stop offering

    @classmethod
    def create(cls, device):
        try:
            return cls(device)
        except AttributeError:          # no storage use count in this torch: results are copied out instead
            return None

    def sync_position(self):
        """Read the device counter (once, after capture: warm-ups and the capture pass advanced it)."""
        self.pos = int(self.counter.item())

    def settle(self, out):
        """After the capture pass (``taken`` was zeroed before it) that returned ``out``: "ring" -- the evaluation's
        result IS the ring's value, replays deliver through it; "copy" -- the last launch did not take the ring, the
        graph's output buffer is copied out as usual; "recapture" -- a launch took it but the result was computed
        from it afterwards: the graph must be captured again without a ring."""
        if self.taken == 0:
            return "copy"
        if self.taken == 1 and out is self.placeholder:
            self.sync_position()
            return "ring"
        return "recapture"

    def claim(self):
        """The tensor the next replay writes.  Call exactly once per replay, before it."""
        i = self.pos
        self.pos = i + 1 if i + 1 < self.n else 0
        s = self.slots[i]
        if _use_count(s) > self.idle:
            s = self.slots[i] = t.empty((), dtype=t.float32, device=self.device)
            self.table[i] = s.data_ptr()           # (a one-element fill on the current stream, ahead of the replay)
        return s


class ResultStrip(ResultRing):
    """A result ring whose slots are elements of ONE buffer, ``stride`` apart from ``offset`` on: for a caller that issues
    many replays before it reads any result (sample.EvalPipeline, whose lanes interleave their slots so that evaluation i
    of the whole pipeline lands in element i) -- replay k of the evaluation delivers to element offset + stride (k % n),
    nothing is claimed or recycled, and a run of results is a view."""

    def __init__(self, device, n, buf=None, offset=0, stride=1):
        self.device, self.n = device, int(n)
        self.buf = buf if buf is not None else t.empty(self.n * stride, dtype=t.float32, device=device)
        self.slots = None
        self.table = (t.arange(self.n, dtype=t.int64, device=device) * stride + offset) * 4 + self.buf.data_ptr()
        self.counter = t.zeros((), dtype=t.int32, device=device)
        self.placeholder = t.empty((), dtype=t.float32, device=device)
        self.pos, self.taken, self.declined = 0, 0, False

    def sync_position(self):
        self.counter.zero_()                       # (warm-ups advanced it: replays start at the first element)
        self.pos = 0

    def claim(self):
        raise RuntimeError("a ResultStrip's elements are read in runs, not claimed one by one")


FP64_SMALL_FACTORS = "fused"
"""What the fused plate step does with an fp64 small factor (the likelihood of fp64 observations, which the reference's
sum of factors promotes to fp64 before its log-sum-exp, utils.py:218-220).  "fused" (default): the factor enters the fp32
kernel converted and the result is returned as fp64 -- inside 1e-6 of the reference's value on the BASELINE
configurations, three launches per evaluation.  "exact": the plate step declines such a problem and the materialised
route runs it -- fp32 factor produced as the reference produces it, factors added and reduced in fp64 by the generic
kernel -- at roughly twice the time."""


_RING = [None]          # set by sample._GraphedELBO around warm-up + capture of one evaluation
_MIRROR_RING = [None]   # a mirror ring on its way into _Reduce.forward (training.GraphedStep: the ELBO kept AND delivered)


class TooLargeForMergedSplit(Exception):
    """A tensor the engine was about to allocate exceeds split.MERGE_MAX_BYTES while a Split plate is being evaluated
    as one merged slice: the plate falls back to the reference's per-chunk loop (logpq.logPQ_plate)."""


_ALLOC_LIMIT = [None]   # bytes; set by logpq.logPQ_plate around a merged Split evaluation


def _empty(shape, dtype, device):
    """torch.empty for the engine's factor-sized outputs, under the merged-Split memory bound."""
    lim = _ALLOC_LIMIT[0]
    if lim is not None:
        n = math.prod(shape) * (8 if dtype == t.float64 else 4)
        if n > lim:
            raise TooLargeForMergedSplit(f"{n} bytes")
    return t.empty(shape, dtype=dtype, device=device)


def _launch(mode, factors, sizes, roles, out, out_dims, weight=None, lse_out=None, add_const=0.0,
            scales=None, out_scale=1.0, ring=None, noise=None, ring_and_out=False):
    space = list(sizes)
    if len(space) > N.MAX_DIMS:
        raise N.NativeError(f"alan_amd: {len(space)} dims in one contraction step (max {N.MAX_DIMS})")
    if len(factors) > N.MAX_FACTORS:
        raise N.NativeError(f"alan_amd: {len(factors)} factors in one contraction step (max {N.MAX_FACTORS})")
    if FP64_SMALL_FACTORS == "exact" and out.dtype == t.float64 and mode in (N.MODE_LSE, N.MODE_SUM, N.MODE_WEXPSUM) \
            and any(x.dtype == t.float32 for x, _ in factors):
        # mixed fp32 / fp64 factors, exactly as the reference's promotion computes them: every factor in fp64 (the
        # library's streaming kernel would otherwise run such a problem at the big fp32 factor's precision, rows.hip)
        factors = [(x.double() if x.dtype == t.float32 else x, d) for x, d in factors]
    desc = N.ReduceDesc()
    desc.mode = mode
    desc.ndim = len(space)
    for i, d in enumerate(space):
        desc.size[i] = sizes[d]
        desc.role[i] = roles[d]
    desc.n_factors = len(factors)
    device = out.device
    for i, (x, dims) in enumerate(factors):
        N.require_device(x, "log-prob factor")
        if x.device != device:
            raise N.NativeError("alan_amd: factors live on different devices")
        N.fill_tensor(desc.factor[i], x, _strides(x, list(dims), space), 1.0 if scales is None else scales[i])
    if weight is not None:
        N.fill_tensor(desc.weight, weight[0], _strides(weight[0], list(weight[1]), space))
    N.fill_tensor(desc.out, out, _strides(out, list(out_dims), space), out_scale)
    if lse_out is not None:
        N.fill_tensor(desc.lse_out, lse_out[0], _strides(lse_out[0], list(lse_out[1]), space))
    desc.add_const = add_const
    # algorithmic bytes of this call: every distinct input element once + the output once
    algo = sum(x.numel() * x.element_size() for x, _ in factors) + out.numel() * out.element_size()
    if weight is not None:
        algo += weight[0].numel() * weight[0].element_size()
    if ring is not None:
        desc.ring_slots, desc.ring_counter, desc.ring_n = ring.table.data_ptr(), ring.counter.data_ptr(), ring.n
        desc.ring_and_out = int(bool(ring_and_out))
    if noise is not None:
        # factor 1 is generated inside the launch (alan_noise_t); noise = (seed, offset, cell, receipt, advance, advance_by),
        # the three in the middle int64 device tensors or None
        seed, offset, cell, receipt, advance, advance_by = noise
        z = desc.noise
        z.on, z.seed, z.offset, z.advance_by = 1, seed, offset, advance_by
        z.cell = cell.data_ptr() if cell is not None else None
        z.receipt = receipt.data_ptr() if receipt is not None else None
        z.advance = advance.data_ptr() if advance is not None else None
    return N.run_reduce(desc, device, algo, keepalive=([x for x, _ in factors], out, weight, lse_out, noise))


def _result_dtype(tensors):
    dt = tensors[0].dtype
    for x in tensors[1:]:
        dt = t.promote_types(dt, x.dtype)
    if dt not in (t.float32, t.float64):
        raise N.NativeError(f"alan_amd: unsupported factor dtype {dt}")
    return dt


# --------------------------------------------------------------------------- autograd function
def sum_slices(parts):
    """[slices, ...] -> the sum over the leading dim, as one SUM launch (dims.PartialSumPT.x)."""
    N.flush()
    toks = tuple(range(parts.ndim))
    out, _ = _reduce_factors([(parts, toks)], plate=(0,))
    return out


def _reduce_forward(spec, tensors, need_grad, ring=None, presum=()):
    """The launch behind _Reduce.forward.  Returns (out, out_dims, lse or None, sizes).  ``ring``: a ResultRing the
    value may be delivered through (then ``out`` is the ring's placeholder).  ``presum``: dims carried by one factor
    each -- the slices of a dims.PartialSumPT -- that are summed as that factor is loaded (role PRESUM); where the
    library declines, the factor is summed by a launch of its own first."""
    dimlists, reduce, plate, add_const = spec
    if presum:
        assert not need_grad and len(presum) == 1
        mode = N.MODE_LSE if reduce else N.MODE_SUM
        res = None if plate else _reduce_presum(mode, tensors, dimlists, reduce, add_const, ring, presum[0])
        if res is not None:
            return res
        tensors, dimlists = list(tensors), list(dimlists)
        for i, d in enumerate(dimlists):
            if presum[0] in d:
                k = d.index(presum[0])
                x = tensors[i].movedim(k, 0) if k else tensors[i]
                tensors[i], dimlists[i] = sum_slices(x), tuple(dd for dd in d if dd != presum[0])
        spec = (tuple(dimlists), reduce, plate, add_const)
    factors = [(x.detach(), d) for x, d in zip(tensors, dimlists)]
    sizes = _space(factors)
    for d in (*reduce, *plate):
        if d not in sizes:
            raise Exception(f"dim {d} to reduce is not on any factor")
    keep = [d for d in sizes if d not in reduce and d not in plate]
    dtype = _result_dtype(list(tensors))
    device = tensors[0].device
    out_dims = _out_order(factors, keep, sizes)
    mode = N.MODE_LSE if reduce else N.MODE_SUM
    if (ring is not None and not ring.declined and reduce and not out_dims and not plate and not need_grad
            and dtype == t.float32 and ring.device == device):
        roles = {d: N.REDUCE for d in sizes}
        if _launch(mode, factors, sizes, roles, ring.placeholder, out_dims, add_const=add_const, ring=ring):
            ring.taken += 1
            return ring.placeholder, out_dims, None, sizes
        ring.declined = True
    out = _empty([sizes[d] for d in out_dims], dtype, device)
    lse = None
    if (ring is not None and getattr(ring, "mirror", False) and not ring.declined and reduce and not out_dims and not plate
            and need_grad and dtype == t.float32 and ring.device == device):
        # a training iteration's ELBO: `out` is kept for the backward, the ring's slot receives the same value (what
        # training.GraphedStep hands its caller: no copy of the graph's output buffer after a replay)
        roles = {d: N.REDUCE for d in sizes}
        if _launch(mode, factors, sizes, roles, out, out_dims, add_const=add_const, ring=ring, ring_and_out=True):
            ring.taken += 1
            return out, out_dims, (out, out_dims), sizes
        ring.declined = True
    if reduce:
        roles = {d: (N.REDUCE if d in reduce else N.PLATE if d in plate else N.KEEP) for d in sizes}
        if plate and need_grad:
            lse_dims = _out_order(factors, keep + list(plate), sizes)
            lse = (_empty([sizes[d] for d in lse_dims], dtype, device), lse_dims)
        _launch(N.MODE_LSE, factors, sizes, roles, out, out_dims, lse_out=lse, add_const=add_const)
        if lse is None:
            lse = (out, out_dims)   # add_const is 0 whenever a backward is needed through here
    else:
        roles = {d: (N.REDUCE if d in plate else N.KEEP) for d in sizes}
        _launch(N.MODE_SUM, factors, sizes, roles, out, out_dims, add_const=add_const)
    return out, out_dims, lse, sizes


def _reduce_presum(mode, tensors, dimlists, reduce, add_const, ring, pdim):
    """One launch that adds the slices of the factor carrying ``pdim`` as it loads it.  None: the library declines."""
    factors = [(x.detach(), d) for x, d in zip(tensors, dimlists)]
    sizes = _space(factors)
    keep = [d for d in sizes if d not in reduce and d != pdim]
    dtype = _result_dtype(list(tensors))
    device = tensors[0].device
    if dtype != t.float32:
        return None
    out_dims = _out_order(factors, keep, sizes)
    roles = {d: (N.PRESUM if d == pdim else N.REDUCE if d in reduce else N.KEEP) for d in sizes}
    if ring is not None and not ring.declined and reduce and not out_dims and ring.device == device:
        if _launch(mode, factors, sizes, roles, ring.placeholder, out_dims, add_const=add_const, ring=ring):
            ring.taken += 1
            return ring.placeholder, out_dims, None, sizes
        ring.declined = True
    out = t.empty([sizes[d] for d in out_dims], dtype=dtype, device=device)
    if not _launch(mode, factors, sizes, roles, out, out_dims, add_const=add_const):
        return None
    return out, out_dims, None, sizes


class _Reduce(t.autograd.Function):
    """out[keep] = sum_plate  LSE_reduce( sum_f factor_f )  (+ add_const);  mode SUM when no reduce dims."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        dimlists, reduce, plate, add_const = spec
        ring, _MIRROR_RING[0] = _MIRROR_RING[0], None
        out, out_dims, lse, sizes = _reduce_forward(spec, tensors, any(x.requires_grad for x in tensors), ring)
        ctx.spec = (dimlists, tuple(reduce), tuple(plate), add_const, tuple(out_dims), sizes)
        ctx.lse_dims = None if lse is None else tuple(lse[1])
        ctx.has_lse = bool(reduce)
        saved = list(tensors) + ([lse[0]] if reduce else [])
        ctx.save_for_backward(*saved)
        ctx.out_dims = tuple(out_dims)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dimlists, reduce, plate, add_const, out_dims, sizes = ctx.spec
        saved = ctx.saved_tensors
        nf = len(dimlists)
        tensors = saved[:nf]
        factors = [(x.detach(), d) for x, d in zip(tensors, dimlists)]
        grad_out = grad_out.detach()
        need = [ctx.needs_input_grad[i + 1] for i in range(nf)]
        if ctx.has_lse and any(need):
            fused = _fused_backward(factors, need, sizes, reduce, plate, saved[nf], ctx.lse_dims, grad_out, out_dims)
            if fused is not None:
                return (None, *fused)
        grads = []
        # the wanted gradients are independent problems: small ones leave as ONE multi-problem launch
        # (alan_reduce_batch; the queue is flushed when the block ends, before anyone can read a gradient)
        import contextlib
        batch = contextlib.nullcontext() if t.is_grad_enabled() else N.deferring()
        with batch, N.may_defer():
            for i, (x, dims) in enumerate(factors):
                if not need[i]:
                    grads.append(None)
                    continue
                g = t.empty(x.shape, dtype=x.dtype, device=x.device)
                roles = {d: (N.KEEP if d in dims else N.REDUCE) for d in sizes}
                if ctx.has_lse:
                    lse = saved[nf]
                    _launch(N.MODE_WEXPSUM, factors + [(lse, ctx.lse_dims)], sizes, roles, g, dims,
                            weight=(grad_out, out_dims), scales=[1.0] * nf + [-1.0])
                else:
                    _launch(N.MODE_SUM, [(grad_out, out_dims)], sizes, roles, g, dims)
                grads.append(g)
        return (None, *grads)


def _fused_backward(factors, need, sizes, reduce, plate, lse, lse_dims, grad_out, out_dims):
    """Every wanted gradient of an LSE(+plate) call from ONE pass over the largest factor
    (alan_reduce_backward); None when the library declines the shape (the caller then issues one WEXPSUM
    launch per factor)."""
    if not all(x.dtype == t.float32 for x, _ in factors) or lse.dtype != t.float32 or grad_out.dtype != t.float32:
        return None
    space = list(sizes)
    if len(space) > N.MAX_DIMS or len(factors) > N.MAX_FACTORS:
        return None
    bd = N.BackwardDesc()
    desc = bd.fwd
    desc.mode = N.MODE_LSE
    desc.ndim = len(space)
    for i, d in enumerate(space):
        desc.size[i] = sizes[d]
        desc.role[i] = N.REDUCE if d in reduce else N.PLATE if d in plate else N.KEEP
    desc.n_factors = len(factors)
    device = factors[0][0].device
    grads = []
    for i, (x, dims) in enumerate(factors):
        N.require_device(x, "log-prob factor")
        N.fill_tensor(desc.factor[i], x, _strides(x, list(dims), space))
        g = None
        if need[i]:
            g = t.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=device)
            if g.untyped_storage().nbytes() != x.numel() * x.element_size():
                return None                      # a factor with gaps / overlaps in memory: not this path
            N.fill_tensor(bd.grad[i], g, _strides(g, list(dims), space))
        grads.append(g)
    go = grad_out.contiguous()
    N.fill_tensor(desc.weight, go, _strides(go, list(out_dims), space))
    N.fill_tensor(desc.lse_out, lse, _strides(lse, list(lse_dims), space))
    N.fill_tensor(desc.out, go, _strides(go, list(out_dims), space))      # unused by the library
    return grads if N.run_reduce_backward(bd, device) else None


class _Tokens:
    """Dim keys -> small ints and back.  functorch ``Dim`` overloads ``==`` (it builds a tensor), so
    keys may only ever be compared through a dict; internally the engine works on ints."""

    def __init__(self):
        self.fwd, self.back = {}, []

    @staticmethod
    def _k(key):
        # hash(Dim) goes through torchdim's Python dispatcher (~40 us); identity is what we mean anyway
        return key if isinstance(key, (str, int, tuple)) else id(key)

    def __call__(self, key):
        k = self._k(key)
        if k not in self.fwd:
            self.fwd[k] = len(self.back)
            self.back.append(key)
        return self.fwd[k]

    def __contains__(self, key):
        return self._k(key) in self.fwd

    def many(self, keys):
        return tuple(self(k) for k in keys)

    def keys(self, toks):
        return tuple(self.back[i] for i in toks)


def reduce_factors(factors, reduce=(), plate=(), add_const=0.0):
    """Fused ``sum(factors)`` -> log-sum-exp over ``reduce`` -> sum over ``plate``.
    Returns (tensor, dims)."""
    N.flush()          # consumers read what queued producer launches write (native.deferring)
    tok = _Tokens()
    factors = [(x, tok.many(d)) for x, d in factors]
    for d in (*reduce, *plate):
        if d not in tok:
            raise Exception(f"dim {d} to reduce is not on any factor")
    out, dims = _reduce_factors(factors, tok.many(reduce), tok.many(plate), add_const)
    return out, tok.keys(dims)


def _reduce_factors(factors, reduce=(), plate=(), add_const=0.0, ring=None, presum=()):
    factors = [(x, tuple(d)) for x, d in factors]
    reduce, plate = tuple(reduce), tuple(plate)
    if presum:
        assert not (t.is_grad_enabled() and any(x.requires_grad for x, _ in factors)), "partial sums: gradient-free only"
        spec = (tuple(d for _, d in factors), reduce, plate, float(add_const))
        out, out_dims, _, _ = _reduce_forward(spec, [x for x, _ in factors], False, ring, tuple(presum))
        return out, tuple(out_dims)
    if add_const != 0.0 and any(x.requires_grad for x, _ in factors) and reduce and not plate:
        # keep the saved log-sum-exp free of the constant: apply it outside
        out, dims = _reduce_factors(factors, reduce, plate, 0.0)
        return out + add_const, dims
    spec = (tuple(d for _, d in factors), reduce, plate, float(add_const))
    tensors = [x for x, _ in factors]
    if not (t.is_grad_enabled() and any(x.requires_grad for x in tensors)):
        out, out_dims, _, _ = _reduce_forward(spec, tensors, False, ring)   # nothing to record: no autograd.Function
        return out, tuple(out_dims)
    _MIRROR_RING[0] = ring if (ring is not None and getattr(ring, "mirror", False)) else None
    out = _Reduce.apply(spec, *tensors)
    _MIRROR_RING[0] = None
    sizes = _space(factors)
    keep = [d for d in sizes if d not in reduce and d not in plate]
    return out, tuple(_out_order([(x.detach(), d) for x, d in factors], keep, sizes))


# --------------------------------------------------------------------------- fused factor producer
def _produce(mode, args, out_dims, affine=(1.0, 0.0), scales=None, noise=None):
    """One producer launch: ``args`` are (tensor, leading first-class dims) pairs whose trailing positional
    dims are right-aligned; every dim not in ``out_dims`` is summed out; out = affine[0] * sum + affine[1].
    ``noise`` (modes AFFINE / DOT): the second argument is standard-normal noise generated inside the launch (_launch);
    None is returned -- nothing launched -- when the library does not take such a problem."""
    tok = _Tokens()
    npos = max(x.ndim - len(d) for x, d in args)
    factors = []
    for x, d in args:
        k = x.ndim - len(d)
        keys = tok.many(d) + tuple(tok(("_e", npos - k + j)) for j in range(k))
        factors.append((x.detach(), keys))
    sizes = _space(factors)
    odims = tok.many(out_dims)
    roles = {d: (N.KEEP if d in odims else N.REDUCE) for d in sizes}
    dtype = _result_dtype([x for x, _ in factors])
    out = _empty([sizes[d] for d in odims], dtype, args[0][0].device)
    ok = _launch(mode, factors, sizes, roles, out, odims, out_scale=float(affine[0]), add_const=float(affine[1]),
                 scales=scales, noise=noise)
    if noise is not None and not ok:
        return None
    return out


def normal_logprob(value, loc, scale, out_dims, log_scale=False, affine=(1.0, 0.0), loc_scale=1.0):
    """log N(value; loc, scale) summed over every positional (sample/batch/event) dim -- and over any
    first-class dim missing from ``out_dims`` (a data-only plate's sum, logpq.py:149) -- as ONE launch
    (alan_reduce mode NORMAL): the [.., K, K, K, d] broadcast the reference materialises
    (TorchDimDist.py:127-162) never exists.  Each argument is (tensor, leading first-class dims);
    trailing positional dims are right-aligned.  Returns a tensor laid out as ``out_dims``.
    ``log_scale``: the third argument holds log(scale) (a learned scale's raw parameter, Param.py:18-25);
    ``affine = (a, b)``: the launch writes a * log_prob + b (the -(log Q + log K) of logpq.py:234-235)."""
    mode = N.MODE_NORMAL_LOGSCALE if log_scale else N.MODE_NORMAL
    scales = None if loc_scale == 1.0 else [1.0, float(loc_scale), 1.0]     # (``loc_scale``: the location is c * loc)
    return _produce(mode, (value, loc, scale), out_dims, affine, scales)


def normal_logprob_pq(value, p, q, out_dims, affine=(1.0, 0.0)):
    """affine[0] * (log N(value; p) - log N(value; q)) + affine[1] in ONE launch: the "log P - log Q - log K"
    of a latent whose prior and approximate posterior are both Normal and carry the same dims
    (logpq.py:221-235).  p, q = (loc, scale, scale_is_log) with loc / scale as (tensor, dims) pairs."""
    (pl, ps, plog), (ql, qs, qlog) = p, q
    scales = [1.0, 1.0, 2.0 if plog else 1.0, -1.0, 1.0, 2.0 if qlog else 1.0]
    return _produce(N.MODE_NORMAL, (value, pl, ps, value, ql, qs), out_dims, affine, scales)


def _normal_lse_args(value, loc, scale, smalls, plate, K):
    """Shape / stride bookkeeping shared by the fused plate step and its backward.  -> dict or None (not this path)."""
    (xv, dv), (xl, dl), (xs, ds) = value, loc, scale
    tensors = [xv, xl, xs, *[x for x, _ in smalls]]
    if not all(x.is_cuda for x in tensors) or len(smalls) > 4:
        return None
    if not all(x.dtype == t.float32 for x in (xv, xl, xs)) or \
            not all(x.dtype in (t.float32, t.float64) for x, _ in smalls):
        return None
    wide = any(x.dtype == t.float64 for x, _ in smalls)
    if wide and FP64_SMALL_FACTORS == "exact":
        return None                                   # (the materialised route adds the factors and takes the log-sum-exp in fp64)
    # (an fp64 small factor -- the likelihood of fp64 observations -- enters the fp32 kernel converted; the result is
    # returned as fp64, the dtype torch's promotion gives the reference's sum of factors)
    smalls = [(x.float() if x.dtype == t.float64 else x, d) for x, d in smalls]
    if len(dv) != 2 or len(dl) != 1 or len(ds) != 1:
        return None
    nev = xv.ndim - 2
    if xl.ndim - 1 != nev or xs.ndim - 1 != nev:
        return None
    if nev == 0:
        xv, xl, xs = xv.unsqueeze(-1), xl.unsqueeze(-1), xs.unsqueeze(-1)
    elif nev > 1:
        if xv.shape[2:] != xl.shape[1:] or xv.shape[2:] != xs.shape[1:]:
            return None
        xv, xl, xs = xv.flatten(2), xl.flatten(1), xs.flatten(1)
    if not (xv.shape[-1] == xl.shape[-1] == xs.shape[-1]):
        return None                                   # broadcasting event shapes: not this path
    ip = 0 if dv[0] is plate else 1
    if dv[ip] is not plate or dv[1 - ip] is not K:
        return None
    sm = []
    for x, dims in smalls:
        if x.ndim != len(dims) or any(id(dd) not in (id(plate), id(K)) for dd in dims):
            return None
        st = {id(dd): (x.stride(j) if x.shape[j] > 1 else 0) for j, dd in enumerate(dims)}
        sm.append((x, st.get(id(plate), 0), st.get(id(K), 0)))
    return dict(xv=xv, xl=xl, xs=xs, ip=ip, smalls=sm, dl=dl[0], ds=ds[0], wide=wide)


def _normal_lse_desc(a, log_scale, d=None):
    d = d if d is not None else N.NormalLseDesc()
    xv, xl, xs, ip = a["xv"], a["xl"], a["xs"], a["ip"]
    d.value, d.v_sm, d.v_sk, d.v_se = xv.data_ptr(), xv.stride(ip), xv.stride(1 - ip), xv.stride(2)
    d.loc, d.l_sl, d.l_se = xl.data_ptr(), xl.stride(0), xl.stride(1)
    d.scale, d.s_ss, d.s_se = xs.data_ptr(), xs.stride(0), xs.stride(1)
    d.log_scale, d.n_small = int(bool(log_scale)), len(a["smalls"])
    for i, (x, s_m, s_k) in enumerate(a["smalls"]):
        d.small[i], d.small_sm[i], d.small_sk[i] = x.data_ptr(), s_m, s_k
    d.M, d.NK, d.NL, d.NS, d.E = xv.shape[ip], xv.shape[1 - ip], xl.shape[0], xs.shape[0], xv.shape[2]
    return d


SCALE_TABLE = True      # the plate step's scale table built by the producers' launch when there is one (ALAN_MODE_NORMAL_TABLE)


def _ride_scale_table(a, d, log_scale, device):
    """With producers queued for the launch in front of this plate step (a gradient-free evaluation), one more problem joins
    them: the step's matrix operand built from ``scale`` (alan_normal_lse_desc_t.scale_table) -- its waves then load it instead
    of building it behind a workgroup barrier.  Same bits either way.  -> the table (kept alive by the caller) or None."""
    import ctypes as C
    nbytes = int(N.lib().alan_normal_lse_table_bytes(C.byref(d)))
    timed = N._TIMER[0] is not None and N.queue_active() and not t.is_grad_enabled()
    if nbytes == 0 or (N.n_pending() == 0 and not timed):
        return None
    xs = a["xs"]
    table = t.empty(nbytes, dtype=t.uint8, device=device)
    rd = N.ReduceDesc()
    rd.mode, rd.ndim, rd.n_factors = N.MODE_NORMAL_TABLE, 2, 1
    rd.size[0], rd.size[1] = xs.shape[0], xs.shape[1]
    rd.role[0], rd.role[1] = N.KEEP, N.REDUCE
    N.fill_tensor(rd.factor[0], xs, (xs.stride(0), xs.stride(1)), 2.0 if log_scale else 1.0)
    rd.out.data, rd.out.dtype, rd.out.scale = table.data_ptr(), N.dtype_code(t.float32), 1.0
    if timed:
        # under profiling.KernelTimer nothing is queued (every launch carries its own events): the table gets a launch of its
        # own there, so that the plate step that is timed is the kernel an evaluation runs
        N.check(N.lib().alan_reduce(C.byref(rd), None, 0, N.current_stream(device)), "alan_reduce(NORMAL_TABLE)")
    elif not N.ride_along(rd, device, keepalive=(xs, table)):
        return None
    d.scale_table = table.data_ptr()
    return table


def _normal_lse_forward(a, log_scale, want_lse, partials=False):
    """One alan_normal_lse launch.  -> (out [NL, NS], lse [M, NL, NS] or None), or None when the library declines.
    ``partials``: out is [slices, NL, NS], the launch's partial sums left for the consumer to add (keep_partials)."""
    d = _normal_lse_desc(a, log_scale)
    device = a["xv"].device
    n_parts = 0
    if partials:
        import ctypes as C
        d.out = d.value                                  # (planning only: any non-null pointer)
        n_parts = int(N.lib().alan_normal_lse_n_partials(C.byref(d)))
    if n_parts > 1:
        out = t.empty(n_parts, d.NL, d.NS, dtype=t.float32, device=device)
        d.out, d.o_sl, d.o_ss, d.add_const, d.keep_partials = out.data_ptr(), d.NS, 1, 0.0, 1
    else:
        out = t.empty(d.NL, d.NS, dtype=t.float32, device=device)
        d.out, d.o_sl, d.o_ss, d.add_const = out.data_ptr(), out.stride(0), out.stride(1), 0.0
    lse = _empty([d.M, d.NL, d.NS], t.float32, device) if want_lse else None
    d.lse_out = lse.data_ptr() if want_lse else None
    table = _ride_scale_table(a, d, log_scale, device) if SCALE_TABLE else None
    if not N.run_normal_lse(d, device, keepalive=(a, out, lse, table)):
        return None
    if a["wide"]:
        N.flush()                                        # (the conversion below reads the launch's output)
    return (out.double() if a["wide"] else out), lse


class _NormalLse(t.autograd.Function):
    """The fused plate step (alan_normal_lse) with its one-pass backward (alan_normal_lse_backward): the
    [plate, l, s, K] factor exists neither in the forward nor in the backward."""

    @staticmethod
    def forward(ctx, spec, value, loc, scale, *smalls):
        vd, ld, sd, small_dims, plate, K, log_scale = spec
        a = _normal_lse_args((value.detach(), vd), (loc.detach(), ld), (scale.detach(), sd),
                             [(x.detach(), d) for x, d in zip(smalls, small_dims)], plate, K)
        res = _normal_lse_forward(a, log_scale, True) if a is not None else None
        if res is None:
            raise N.NativeError("alan_normal_lse declined a shape normal_lse_supported() accepted")
        out, lse = res
        ctx.spec = spec
        ctx.save_for_backward(value, loc, scale, lse, *smalls)
        return out

    @staticmethod
    @t.autograd.function.once_differentiable
    def backward(ctx, G):
        vd, ld, sd, small_dims, plate, K, log_scale = ctx.spec
        value, loc, scale, lse, *smalls = ctx.saved_tensors
        a = _normal_lse_args((value.detach(), vd), (loc.detach(), ld), (scale.detach(), sd),
                             [(x.detach(), d) for x, d in zip(smalls, small_dims)], plate, K)
        need_v, need_l, need_s = ctx.needs_input_grad[1:4]
        need_small = list(ctx.needs_input_grad[4:])
        bd = N.NormalLseBackwardDesc()
        _normal_lse_desc(a, log_scale, bd.fwd)
        device = value.device
        M, NK, NL, NS, E = bd.fwd.M, bd.fwd.NK, bd.fwd.NL, bd.fwd.NS, bd.fwd.E
        Gc = G.detach().to(t.float32).contiguous()
        bd.lse, bd.grad_out, bd.g_sl, bd.g_ss = lse.data_ptr(), Gc.data_ptr(), Gc.stride(0), Gc.stride(1)
        gv = t.empty(M, NK, E, dtype=t.float32, device=device) if need_v else None
        gls = t.empty(NL * E + NS * E, dtype=t.float32, device=device) if (need_l or need_s) else None
        gsm = t.empty(M, NK, dtype=t.float32, device=device) if any(need_small) else None
        bd.grad_value = gv.data_ptr() if need_v else None
        bd.grad_loc = gls.data_ptr() if need_l else None
        bd.grad_scale = gls[NL * E:].data_ptr() if need_s else None
        bd.grad_small = gsm.data_ptr() if gsm is not None else None
        if not N.run_normal_lse_backward(bd, device):
            raise N.NativeError("alan_normal_lse_backward declined the shape its forward accepted")
        grads = [None, None, None]
        if need_v:
            g = gv if a["ip"] == 0 else gv.transpose(0, 1)
            grads[0] = g.reshape(value.shape)
        if need_l:
            grads[1] = gls[: NL * E].view(loc.shape)
        if need_s:
            grads[2] = gls[NL * E:].view(scale.shape)
        out_small = []
        for x, dims, need in zip(smalls, small_dims, need_small):
            if not need:
                out_small.append(None)
                continue
            # d small_f = d (sum of smalls) summed over the dims (and broadcast size-1 axes) that factor lacks
            g = gsm                                                       # [plate, K]
            axes = {id(plate): 0, id(K): 1}
            have = [axes[id(dd)] for dd in dims]
            drop = [ax for ax in (0, 1) if ax not in have]
            if drop:
                g = g.sum(drop)
            if len(have) == 2 and have[0] > have[1]:
                g = g.t()
            for j, n in enumerate(x.shape):
                if n == 1 and g.shape[j] != 1:
                    g = g.sum(j, keepdim=True)
            out_small.append(g.to(x.dtype))
        return (None, *grads, *out_small)


def normal_lse(value, loc, scale, smalls, plate, K, log_scale=False, partials=False):
    """out[l, s] = sum_plate LSE_K( log N(value[plate,K,:]; loc[l,:], scale[s,:]) + sum small[plate,K] ) in one
    launch, the [plate, l, s, K] factor never materialised (alan_normal_lse).  value = (tensor, (two dims: the plate
    and K, any order)); loc / scale = (tensor, (one dim,)); smalls = [(tensor, dims within {plate, K})].
    Differentiable in every argument (alan_normal_lse_backward).  Returns (out, (loc dim, scale dim)) or None when
    the library declines.  ``partials`` (gradient-free calls only): out may be [slices, l, s] -- the launch's per-slice
    partial sums, for a consumer that adds them on load (dims.PartialSumPT)."""
    tensors = [value[0], loc[0], scale[0], *[x for x, _ in smalls]]
    recording = t.is_grad_enabled() and any(x.requires_grad for x in tensors)
    if any(x.dtype != t.float32 for x in tensors):
        N.flush()      # a dtype conversion below reads what queued producer launches write
    # (with gradients recorded the step runs inside _NormalLse.forward -- grad mode off in there, like the producers' own
    # forwards that queued the launches -- and autograd's bookkeeping reads no tensor data)
    # (else the queue stays as it is until native.run_normal_lse issues it in front of this launch -- nothing below reads a
    # tensor -- and the step's scale table can still join it: _ride_scale_table)
    a = _normal_lse_args(value, loc, scale, smalls, plate, K)
    if a is None:
        N.flush()
        return None
    if recording:
        d = _normal_lse_desc(a, log_scale)
        d.out = d.value                                  # (planning only: any non-null pointer)
        L = N.lib()
        import ctypes as C
        bd = N.NormalLseBackwardDesc()
        _normal_lse_desc(a, log_scale, bd.fwd)
        bd.lse = bd.grad_out = bd.grad_small = bd.fwd.value
        if L.alan_normal_lse_workspace_bytes(C.byref(d)) == 0 or \
                L.alan_normal_lse_backward_workspace_bytes(C.byref(bd)) == 0:
            return None                                  # forward or backward would decline: the materialised route
        spec = (tuple(value[1]), tuple(loc[1]), tuple(scale[1]), tuple(tuple(d_) for _, d_ in smalls), plate, K,
                bool(log_scale))
        out = _NormalLse.apply(spec, value[0], loc[0], scale[0], *[x for x, _ in smalls])
        return out, (a["dl"], a["ds"])
    res = _normal_lse_forward(a, log_scale, False, partials)       # (partials: out may come back [slices, l, s])
    if res is None:
        N.flush()
        return None
    return res[0], (a["dl"], a["ds"])


def producer_grads(G, out_dims, args, wanted, kinds, log_scale=False, scale=1.0):
    """Backward of a fused producer (alan_reduce mode PRODUCER_GRAD): ``args`` = the producer's arguments as (tensor,
    leading first-class dims) pairs -- (value, loc, scale) or (value, logits) --, ``G`` the upstream gradient laid out
    like ``out_dims``; returns the gradient of every argument i with wanted[i] (shaped like the argument), else None.
    kinds[i] = N.GRAD_*.  The wanted gradients are independent problems: they leave as ONE multi-problem launch when
    they are small (alan_reduce_batch).  Returns None when the shapes do not fit (broadcast event dims)."""
    tok = _Tokens()
    npos = max(x.ndim - len(d) for x, d in args)
    facs = []
    for x, d in args:
        k = x.ndim - len(d)
        keys = tok.many(d) + tuple(tok(("_e", npos - k + j)) for j in range(k))
        facs.append((x.detach(), keys))
    gfac = (G.detach(), tok.many(out_dims))
    try:
        sizes = _space([gfac, *facs])
    except Exception:
        return None                                  # a size-1 event dim broadcasting against a longer one
    outs = [None] * len(args)
    if t.is_grad_enabled():
        return None                                  # (double backward: the torch formulas)
    with N.deferring(), N.may_defer():
        for i, ((x, keys), want) in enumerate(zip(facs, wanted)):
            if not want:
                continue
            roles = {d: (N.KEEP if d in keys else N.REDUCE) for d in sizes}
            out = t.empty([sizes[d] for d in keys], dtype=_result_dtype([G, *[y for y, _ in facs]]), device=G.device)
            scales = [kinds[i]] + [1.0] * len(facs)
            if log_scale and len(facs) == 3:
                scales[3] = 2.0
            _launch(N.MODE_PRODUCER_GRAD, [gfac, *facs], sizes, roles, out, keys, scales=scales, out_scale=float(scale))
            outs[i] = out
    # (casts only now: inside the block the launches are still queued and `out` unwritten)
    return [o if o is None or o.dtype == a[0].dtype else o.to(a[0].dtype) for o, a in zip(outs, args)]


def bernoulli_logprob(value, logits, out_dims, affine=(1.0, 0.0)):
    """log Bernoulli(value; logits=logits), summed like ``normal_logprob`` (alan_reduce mode BERNOULLI)."""
    return _produce(N.MODE_BERNOULLI, (value, logits), out_dims, affine)


def dot_sum(a, b, out_dims):
    """sum over the trailing positional dim(s) -- and any first-class dim missing from ``out_dims`` -- of a * b, as one
    launch (alan_reduce mode DOT; small ones ride in a queued multi-problem launch): a term ``phi @ bus_company_name`` of
    a model lambda, evaluated once.  Each argument is (tensor, leading first-class dims)."""
    return _produce(N.MODE_DOT, (a, b), out_dims)


def bernoulli_linear_logprob(value, terms, out_dims, affine=(1.0, 0.0)):
    """log Bernoulli(value; logits = sum of ``terms``) summed over every first-class dim missing from ``out_dims``, as
    ONE launch that computes the logits itself (alan_reduce mode BERNOULLI_LINEAR): what the reference evaluates as the
    model's lambda (a batched matmul and adds through torchdim) followed by TorchDimDist.py:127-162.  ``value``:
    (tensor, dims) with no positional dims; a term: ((tensor, dims),) -- a plain summand, no positional dims -- or
    ((a, dims), (b, dims)) -- contracted over their one trailing positional dim.  Returns the tensor laid out as
    ``out_dims``, or None when the shape is not one the kernel takes (the caller then evaluates the lambda)."""
    tok = _Tokens()
    factors, scales, dots = [(value[0].detach(), tok.many(value[1]))], [1.0], []
    for ti, term in enumerate(terms):
        if len(term) == 1:
            (x, d), = term
            factors.append((x.detach(), tok.many(d)))
        else:
            e = tok(("_dot", ti))
            dots.append(e)
            for x, d in term:
                factors.append((x.detach(), tok.many(d) + (e,)))
        scales += [float(ti + 1)] * len(term)
    if len(factors) > N.MAX_FACTORS or any(x.dtype != t.float32 for x, _ in factors):
        return None
    try:
        sizes = _space(factors)
    except Exception:               # operands whose sizes do not line up: the lambda's own error message is better
        return None
    if len(sizes) > N.MAX_DIMS:
        return None
    odims = tok.many(out_dims)
    roles = {d: (N.DOT if d in dots else N.KEEP if d in odims else N.REDUCE) for d in sizes}
    out = _empty([sizes[d] for d in odims], t.float32, value[0].device)
    ok = _launch(N.MODE_BERNOULLI_LINEAR, factors, sizes, roles, out, odims, out_scale=float(affine[0]),
                 add_const=float(affine[1]), scales=scales)
    return out if ok else None


def _linear_factors(value, terms):
    """(factors, scales, dot tokens, tok) of a linear-logits launch: the value, then every term's operands tagged with
    their term number (factor.scale), a dot term's two operands sharing one extra DOT dim."""
    tok = _Tokens()
    factors, scales, dots = [(value[0].detach(), tok.many(value[1]))], [1.0], []
    for ti, term in enumerate(terms):
        if len(term) == 1:
            (x, d), = term
            factors.append((x.detach(), tok.many(d)))
        else:
            e = tok(("_dot", ti))
            dots.append(e)
            for x, d in term:
                factors.append((x.detach(), tok.many(d) + (e,)))
        scales += [float(ti + 1)] * len(term)
    return factors, scales, dots, tok


def bernoulli_linear_grad(G, value, terms, out_dims, scale=1.0):
    """Gradient of ``bernoulli_linear_logprob`` with respect to the FIRST operand of its FIRST term, which must be a dot
    product (movielens' z in ``z @ x``): sum over the summed dims of G * (value - sigmoid(logits)) * b, one launch that
    recomputes the logits (alan_reduce mode BERNOULLI_LINEAR_GRAD) -- what autograd derives from the lambda's batched
    matmul and Bernoulli.log_prob.  Returns the gradient shaped like that operand, or None when the library declines."""
    if len(terms[0]) != 2:
        return None
    factors, scales, dots, tok = _linear_factors(value, terms)
    a, a_keys = factors[1]
    if len(factors) > N.MAX_FACTORS or any(x.dtype != t.float32 for x, _ in factors) or G.dtype != t.float32:
        return None
    try:
        sizes = _space(factors)
    except Exception:
        return None
    if len(sizes) > N.MAX_DIMS:
        return None
    odims = tok.many(out_dims)
    if set(a_keys[:-1]) != set(odims):
        return None                                   # the operand must carry exactly the kept dims
    roles = {d: (N.DOT if d in dots else N.KEEP if d in odims else N.REDUCE) for d in sizes}
    out = t.empty(a.shape, dtype=t.float32, device=a.device)             # contiguous, dims ordered like the operand
    ok = _launch(N.MODE_BERNOULLI_LINEAR_GRAD, factors, sizes, roles, out, a_keys, weight=(G.detach().contiguous(), odims),
                 out_scale=float(scale), scales=scales)
    return out if ok else None


PRESUM_DIMS = {}         # slices -> the Dim that stands for "slice of a partial sum" (kept alive: contract() knows them by identity)


def presum_dim(n):
    d = PRESUM_DIMS.get(n)
    if d is None:
        d = PRESUM_DIMS[n] = Dim("partial_sum_slices", n)
    return d


# --------------------------------------------------------------------------- elimination planner
LONG_SINGLE_REDUCTION = {True: 16384, False: 2048}
"""Elements of a reduction with (almost) nothing kept above which one K is peeled per step: fp32 problems have the
1024-thread kernel for one long output (reduce.hip, reduce_wide_kernel: K=100's 10^4-element top level in two rounds of
loads), anything else a 256-thread workgroup that would walk it serially (38 us at K=100)."""


def plan_elimination(dimsets, sizes, Ks, fp32=False):
    """Order of K eliminations.  Returns a list of steps ``(factor_ids, Ks_now)`` over a growing list of
    factors (each step appends its result).  Greedy smallest-intermediate-first variable elimination;
    a step absorbs every other K whose factors fit inside the step's index space, so e.g. the
    movielens top level (a[Ka], b[Kb], T[Ka,Kb]) is ONE launch rather than the reference's two
    pairwise steps (reduce_Ks.py:270-281)."""
    Ks = [k for k in Ks]
    live = {i: set(ds) for i, ds in enumerate(dimsets)}
    stored = {i: tuple(ds) for i, ds in enumerate(dimsets)}      # storage order of the input factors
    steps = []
    nxt = len(dimsets)
    remaining = [k for k in Ks if any(k in ds for ds in live.values())]
    while remaining:
        best = None
        for k in remaining:
            grp = [i for i, ds in live.items() if k in ds]
            union = set().union(*[live[i] for i in grp])
            cost = math.prod(sizes[d] for d in union)
            if best is None or cost < best[0]:
                best = (cost, k, grp, union)
        _, k, grp, union = best
        grp = list(grp)
        # absorb factors of other Ks that live entirely inside this step's index space
        changed = True
        while changed:
            changed = False
            for k2 in remaining:
                g2 = [i for i, ds in live.items() if k2 in ds]
                if all(live[i] <= union for i in g2) and not set(g2) <= set(grp) \
                        and len(set(grp) | set(g2)) <= N.MAX_FACTORS - 1:
                    grp = sorted(set(grp) | set(g2))
                    changed = True
        if len(grp) > N.MAX_FACTORS - 1:
            # too many factors for one launch: pre-add the smallest ones (plain broadcast sum), at most MAX_FACTORS of
            # them per step -- the loop comes back here until the group fits (with gradients every variable of a
            # Group contributes log P and -log Q separately: 6 Normals on one K are 12 factors)
            grp_sorted = sorted(grp, key=lambda i: math.prod(sizes[d] for d in live[i]))
            pre = grp_sorted[: min(N.MAX_FACTORS, len(grp) - (N.MAX_FACTORS - 2))]
            steps.append((tuple(pre), ()))
            live[nxt] = set().union(*[live[i] for i in pre])
            for i in pre:
                del live[i]
            nxt += 1
            continue
        others = set().union(*[ds for i, ds in live.items() if i not in grp]) if len(live) > len(grp) else set()
        now = tuple(kk for kk in remaining if kk in union and kk not in others)
        if len(now) > 1:
            # several Ks at once with (almost) nothing kept is one long serial reduction in a single
            # workgroup (K=100: 10^4 elements, 38 us): peel off one K per step instead -- the innermost
            # one of the largest factor, so that step streams it with coalesced loads
            red = math.prod(sizes[kk] for kk in now)
            kept = math.prod(sizes[dd] for dd in union if dd not in now)
            if red > LONG_SINGLE_REDUCTION[bool(fp32)] and kept < 256:
                big = max(grp, key=lambda i: math.prod(sizes[dd] for dd in live[i]))
                pos = {dd: j for j, dd in enumerate(stored.get(big, ()))}
                now = (max(now, key=lambda kk: pos.get(kk, -1)),)
        steps.append((tuple(grp), now))
        live[nxt] = {d for d in union if d not in now}
        for i in grp:
            del live[i]
        nxt += 1
        remaining = [kk for kk in remaining if kk not in now]
    if len(live) > 1:
        ids = sorted(live)
        while len(ids) > N.MAX_FACTORS:
            steps.append((tuple(ids[: N.MAX_FACTORS]), ()))
            ids = [nxt] + ids[N.MAX_FACTORS:]
            nxt += 1
        steps.append((tuple(ids), ()))
    return steps


def contract(factors, Ks, plate=(), final=False):
    """reduce_Ks on positional factors, with the trailing plate sum fused into the last launch.
    Returns (result, dims, per-step record) -- the record is what sample_Ks-style consumers need.
    ``final``: this contraction ends an evaluation (its result is THE scalar): inside a graph capture its last launch
    may deliver through the ResultRing."""
    # consumers read what queued producer launches write (native.deferring) -- except the launches of an evaluation's
    # FINAL contraction behind a queued fused plate step, which may join that launch as its tail (native.tail_attach):
    # the first launch that cannot flushes the queue as usual
    chain = final and N.fused_pending() and not t.is_grad_enabled()
    if not chain:
        N.flush()
    with (N.tail_attach() if chain else contextlib.nullcontext()):
        return _contract(factors, Ks, plate, final)


def _contract(factors, Ks, plate, final):
    tok = _Tokens()
    factors = [(x, tok.many(d)) for x, d in factors]
    sizes = _space(factors)
    for k in (*Ks, *plate):
        if k not in tok:
            raise Exception(f"dim {k} to sum is not on any factor")
    Ks = tok.many(Ks)
    plate = tok.many(plate)
    # (the slice dim of a partial-sum factor, dims.PartialSumPT: invisible to the planner, summed as its step loads it)
    presum = {tok(d) for d in PRESUM_DIMS.values() if d in tok}
    steps = plan_elimination([tuple(dd for dd in d if dd not in presum) for _, d in factors], sizes, Ks,
                             fp32=all(x.dtype == t.float32 and x.is_cuda for x, _ in factors) and not t.is_grad_enabled())
    if not steps:
        steps = [(tuple(range(len(factors))), ())] if (plate or len(factors) > 1) else []
    pool = list(factors)
    record = []
    for si, (ids, now) in enumerate(steps):
        last = si == len(steps) - 1
        group = [pool[i] for i in ids]
        record.append(([(x, tok.keys(d)) for x, d in group], tok.keys(now)))
        out, dims = _reduce_factors(group, reduce=now, plate=plate if last else (),
                                    ring=_RING[0] if (final and last) else None,
                                    presum=tuple(p for p in presum if any(p in d for _, d in group)))
        pool.append((out, dims))
    if not steps:
        return factors[0][0], tok.keys(factors[0][1]), record
    out, dims = pool[-1]
    return out, tok.keys(dims), record
