"""
Small helpers over functorch.dim ("torchdim") first-class dimensions.

alan represents every sample / log-prob as a torchdim tensor whose K- and plate-dims are first
class ``Dim`` objects.  The HIP engine works on positional strided tensors, so this module is the
bridge: ``unwrap`` turns a torchdim tensor into (positional view, dims) without copying, ``wrap``
goes back.  (Role of the reference's generic_dims/generic_order/generic_getitem, utils.py:229-273.)
"""
import functorch.dim
import torch as t
from functorch.dim import Dim

DimTensor = functorch.dim.Tensor
AnyTensor = (t.Tensor, DimTensor)


def is_dimtensor(x):
    return isinstance(x, DimTensor)


def is_tensor(x):
    return isinstance(x, AnyTensor)


def dims_of(x):
    """First-class dims carried by ``x`` (empty for plain tensors and numbers)."""
    return tuple(x.dims) if is_dimtensor(x) else ()


def check_dims(dims, what="dims"):
    """Same complaints as utils.py:154-171: list/tuple, unique, all ``Dim``."""
    if not isinstance(dims, (list, tuple)):
        raise Exception(f"{what} must be a list or tuple")
    if len(set(dims)) != len(dims):
        raise Exception(f"Non-unique elements in {what}")
    for d in dims:
        if not isinstance(d, Dim):
            raise Exception(f"dim in {what} is not torchdim dimension")


def dim_in(d, seq):
    """Identity membership: ``Dim`` overloads ``==`` to build a tensor, so never use ``in`` on a
    list/tuple of dims."""
    return any(d is e for e in seq)


def union_dims(tensors):
    """Ordered union of the dims of several tensors."""
    seen = {}
    for x in tensors:
        for d in dims_of(x):
            seen.setdefault(d, None)
    return tuple(seen)


def unwrap(x, dims=None):
    """torchdim tensor -> positional tensor whose leading dims are ``dims`` (default: x's own dims,
    in the order torchdim reports them).  A view: strides are whatever the producer left behind."""
    if dims is None:
        dims = dims_of(x)
    dims = tuple(dims)
    if not dims:
        return x, ()
    return x.order(*dims), dims


def wrap(x, dims):
    """Positional tensor -> torchdim tensor binding the leading positional dims to ``dims``."""
    dims = tuple(dims)
    if not dims:
        return x
    return x[dims]


def named_to_dim(x, platedims):
    """Named tensor -> torchdim tensor (names must be plate names); unnamed dims stay positional."""
    if not isinstance(x, t.Tensor):
        return x
    for n in x.names:
        if n is not None and n not in platedims:
            raise Exception(f"No torchdim dimension for named dimension {n}")
    idx = [slice(None) if n is None else platedims[n] for n in x.names]
    x = x.rename(None)
    return x[tuple(idx)] if any(isinstance(i, Dim) for i in idx) else x


def dim_to_named(x, order=None):
    """torchdim tensor -> named tensor (dims first, in ``order`` if given)."""
    ds = dims_of(x)
    if order is not None:
        have = set(ds)
        ds = tuple(d for d in order if d in have)
    if not ds:
        return x
    pos = x.order(*ds)
    return pos.refine_names(*[str(d) for d in ds], *([None] * (pos.ndim - len(ds))))


def sum_positional(x):
    """Sum out every positional dim (utils.py:147-152 ``sum_non_dim``)."""
    if is_tensor(x) and x.ndim > 0:
        return x.sum(tuple(range(x.ndim)))
    return x


# ---------------------------------------------------------------------------------------------
# PT: the host stack's internal tensor representation.  functorch.dim in torch 2.10 is implemented
# in Python (every op, and even hash(Dim), costs tens of microseconds), so the plate recursion does
# NOT compute on torchdim tensors: it carries (plain positional tensor, leading Dim objects) pairs and
# only wraps into torchdim when calling a user's model lambda.
class PT:
    __slots__ = ("x", "dims", "ids")

    def __init__(self, x, dims=()):
        self.x = x
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @staticmethod
    def of(v):
        """torchdim tensor / plain tensor / number -> PT"""
        if isinstance(v, PT):
            return v
        if is_dimtensor(v):
            ds = tuple(v.dims)
            return PT(v.order(*ds), ds)
        if not isinstance(v, t.Tensor):
            v = t.as_tensor(v)
        return PT(v, ())

    def dim(self):
        """-> torchdim tensor (or the plain tensor when there are no first-class dims)"""
        return self.x[self.dims] if self.dims else self.x

    @property
    def n_pos(self):
        return self.x.ndim - len(self.dims)

    def size_of(self, dim_id):
        return self.x.shape[self.ids.index(dim_id)]

    def has(self, d):
        return id(d) in self.ids

    def detach(self):
        return PT(self.x.detach(), self.dims)


class ExpPT(PT):
    """A parameter declared as OptParam(..., transformation=torch.exp): behaves as PT(exp(raw)) -- ``.x``
    materialises the transform on first use -- but keeps ``raw`` so the fused Normal producer can take the
    log-scale directly (alan_reduce mode NORMAL_LOGSCALE) and skip the exp launch."""
    __slots__ = ("raw", "_val")

    def __init__(self, raw, dims=()):
        self.raw = raw
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def x(self):
        if self._val is None:
            self._val = self.raw.exp()
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return self.raw.ndim - len(self.dims)

    def size_of(self, dim_id):
        return self.raw.shape[self.ids.index(dim_id)]

    def detach(self):
        return ExpPT(self.raw.detach(), self.dims)


class ReparamPT(PT):
    """A reparameterised Normal sample x = loc + eps * scale, with a key of the parameter tensors that made it (address,
    layout, version: dist._tkey).  A log-prob of x under that very distribution then knows log q(x) = -eps^2 / 2 -
    log scale - const: its total gradient reaches the scale only (dist._OwnSampleLogProb)."""
    __slots__ = ("src", "x2")

    def __init__(self, x, dims, src, x2=None):
        PT.__init__(self, x, dims)
        self.src = src
        # a second autograd output of the node that drew the sample, aliasing the same memory: the consumer that takes it
        # (the variable's own log P term, logpq.logPQ_group) sends its gradient to the node separately, so autograd has no
        # two contributions to add with a kernel of its own (dist._ReparamNormalBatch)
        self.x2 = x2

    def detach(self):
        return PT(self.x.detach(), self.dims)

    def alias(self):
        """The sample for the ONE consumer that may take the second output (else the sample itself)."""
        return PT(self.x2, self.dims) if self.x2 is not None else self


class PendingPT(ReparamPT):
    """A Normal draw that waits in a batch of draws (dist._DrawBatch) for its launch: reading ``.x`` issues the batch.
    Lives only inside one ancestral sampling pass (BoundPlate._sample replaces it by a ReparamPT / PT at the end)."""
    __slots__ = ("_val", "_batch", "_shape", "_dtype", "_device")

    def __init__(self, batch, dims, src, shape, dtype, device):
        self._val, self._batch, self._shape, self._dtype, self._device = None, batch, tuple(shape), dtype, device
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)
        self.src = src
        self.x2 = None

    @property
    def x(self):
        if self._val is None:
            self._batch.flush()
        return self._val

    def settled(self):
        return ReparamPT(self.x, self.dims, self.src, self.x2) if self.src is not None else PT(self.x, self.dims)


class ShiftPT(PT):
    """The previous state of a timeseries, ``prev[t] = first if t == 0 else rest[t - 1]`` along the time dim (what
    Timeseries.py:205-245 builds with a concatenation), not concatenated yet: ``.x`` does it the first time anyone asks,
    but the chain's first round can read the transition's location from the two sources themselves
    (alan_chain_normal_t.loc0: logpq._chain_of_terms) -- no cat launch, 4.9 us of a 54 us evaluation at T = 1000, K = 30.
    ``first``: tensor over the leading dims; ``rest``: tensor over (leading dims, time), the full series; the time dim
    sits at position ``axis``."""
    __slots__ = ("first", "rest", "axis", "_val")

    def __init__(self, first, rest, axis, dims):
        self.first, self.rest, self.axis = first, rest, axis
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def x(self):
        if self._val is None:
            n = self.rest.shape[self.axis]
            self._val = t.cat([self.first.unsqueeze(self.axis), self.rest.narrow(self.axis, 0, n - 1)], self.axis)
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return self.rest.ndim - len(self.dims)

    def size_of(self, dim_id):
        return self.rest.shape[self.ids.index(dim_id)]

    def detach(self):
        return self


class ScaledPT(PT):
    """The value of a model lambda ``c * v`` (a timeseries transition's ``lambda prev: 0.9 * prev``), not evaluated:
    ``.x`` multiplies on first use, a fused Normal producer takes ``raw`` and the constant instead (the loc factor's
    scale field) and the multiply launch never happens.  Only built where no gradient is wanted."""
    __slots__ = ("_raw", "src", "mul", "_val")

    def __init__(self, raw, mul, dims=(), src=None):
        """``src``: the PT the value came from when that is itself lazy (a ShiftPT): ``raw`` then evaluates it."""
        self._raw, self.src, self.mul = raw, src, float(mul)
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def raw(self):
        if self._raw is None:
            self._raw = self.src.x
        return self._raw

    @property
    def x(self):
        if self._val is None:
            self._val = self.raw * self.mul
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return (self.src.n_pos if self._raw is None else self._raw.ndim - len(self.dims))

    def size_of(self, dim_id):
        return self.src.size_of(dim_id) if self._raw is None else self._raw.shape[self.ids.index(dim_id)]

    def detach(self):
        return self


class LazyNormalPT(PT):
    """The log-prob factor of a Normal whose value / loc / scale carry disjoint dims (the big [plate, K, K, K]
    tensor of a hierarchical model), not computed yet: ``.x`` produces it (alan_reduce mode NORMAL) the first time
    anyone asks, but the plate recursion can hand the ingredients to the fused plate-step kernel instead
    (engine.normal_lse) and never materialise it."""
    __slots__ = ("value", "loc", "scale", "log_scale", "grad", "loc_mul", "_val")

    def __init__(self, value, loc, scale, log_scale, dims, grad=False, loc_mul=1.0):
        """``scale`` holds log(scale) when ``log_scale``.  ``grad``: the arguments are attached to the autograd graph
        (elbo_vi / elbo_rws): materialising then goes through the producer's autograd function.  ``loc_mul``: the
        location is loc_mul * loc (a timeseries transition ``lambda prev: c * prev``; that factor's dims are NOT disjoint
        -- value and loc share the time dim -- and its consumer is the chain's first round, logpq._chain_of_terms)."""
        self.value, self.loc, self.scale, self.log_scale, self.grad = value, loc, scale, log_scale, grad
        self.loc_mul = float(loc_mul)
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def x(self):
        if self._val is None:
            if self.grad:
                from .dist import _FusedNormalLogProb
                spec = (self.value.dims, self.loc.dims, self.scale.dims, self.dims, self.log_scale, (1.0, 0.0))
                self._val = _FusedNormalLogProb.apply(spec, self.value.x, self.loc.x, self.scale.x)
            else:
                from . import engine as E
                self._val = E.normal_logprob((self.value.x, self.value.dims), (self.loc.x, self.loc.dims),
                                             (self.scale.x, self.scale.dims), self.dims, log_scale=self.log_scale,
                                             loc_scale=self.loc_mul)
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return 0

    def size_of(self, dim_id):
        return self.dims[self.ids.index(dim_id)].size

    def detach(self):
        return self


class PartialSumPT(PT):
    """A plate step's result left as the per-slice partial sums of its launch, ``parts`` = [slices, *dims] (the fused
    plate step with keep_partials): ``.x`` adds them up the first time anyone asks (one SUM launch, what the library's
    own second stage would have been), but the contraction that consumes the factor can take ``parts`` itself and add
    the slices on load (engine.contract, role PRESUM) -- one launch fewer per evaluation.  Gradient-free only."""
    __slots__ = ("parts", "_val")

    def __init__(self, parts, dims):
        self.parts = parts
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def x(self):
        if self._val is None:
            from . import engine as E
            self._val = E.sum_slices(self.parts)
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return 0

    def size_of(self, dim_id):
        return self.parts.shape[1 + self.ids.index(dim_id)]

    def detach(self):
        return self


class LinearPT(PT):
    """The value of a model lambda that is a sum of arguments and dot products of arguments (movielens' logits
    ``lambda z, x: z @ x``; bus_breakdown's ``alpha + phi @ bus_company_name + psi @ run_type``), not evaluated yet:
    ``.x`` evaluates it the usual way (``make``) the first time anyone asks, but a Bernoulli log-prob can hand the
    terms to the producer kernel that computes the logits itself (alan_reduce mode BERNOULLI_LINEAR) -- no batched
    GEMM, no adds, no logits tensor.  ``terms``: tuples of one PT (a plain summand, no positional dims) or two PTs
    (contracted over their single positional dim).  Only built where no gradient is wanted."""
    __slots__ = ("terms", "make", "_val", "grad")

    def __init__(self, terms, dims, make, grad=None):
        """``grad``: None, or the index of the ONE term whose first operand is attached to the autograd graph (a dot
        term: movielens' z under elbo_vi) -- the Bernoulli log-prob then goes through dist._BernoulliLinear, whose
        backward is the library's own launch; ``.x`` still evaluates the lambda through torch's autograd."""
        self.terms, self.make = tuple(terms), make
        self.grad = grad
        self._val = None
        self.dims = tuple(dims)
        self.ids = tuple(id(d) for d in self.dims)

    @property
    def x(self):
        if self._val is None:
            self._val = self.make()
            self.make = None
        return self._val

    @property
    def materialised(self):
        return self._val is not None

    @property
    def n_pos(self):
        return 0

    def size_of(self, dim_id):
        return self.dims[self.ids.index(dim_id)].size

    def detach(self):
        return self


def pt_order(pts, lead=(), last=()):
    """Ordered union of the dims of several PTs: ``lead`` dims first, ``last`` dims last, others between.
    Returns (dims, ids)."""
    seen = {}
    for p in pts:
        for d, i in zip(p.dims, p.ids):
            if i not in seen:
                seen[i] = d
    rank = {id(d): k for k, d in enumerate(lead)}
    mid = len(rank)
    for k, d in enumerate(last):
        rank[id(d)] = mid + 1 + k
    ids = sorted(seen, key=lambda i: rank.get(i, mid))
    return tuple(seen[i] for i in ids), tuple(ids)


def pt_align(p, ids, pad=0):
    """Plain tensor view of ``p`` laid out [ids (size 1 where absent)..., 1*pad, own positional...]."""
    x = p.x
    nd = len(p.dims)
    if nd:
        pos = {i: k for k, i in enumerate(p.ids)}
        perm = [pos[i] for i in ids if i in pos]
        if perm != list(range(nd)):
            x = x.permute(*perm, *range(nd, x.ndim))
        idx = tuple(slice(None) if i in pos else None for i in ids) + (None,) * pad
    else:
        idx = (None,) * (len(ids) + pad)
    return x[idx] if idx else x


def pt_add(a, b, alpha=1.0):
    """a + alpha*b with broadcasting over first-class dims (no positional dims allowed)."""
    dims, ids = pt_order((a, b))
    xa, xb = pt_align(a, ids), pt_align(b, ids)
    return PT(xa + xb if alpha == 1.0 else t.add(xa, xb, alpha=alpha), dims)
