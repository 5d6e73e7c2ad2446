"""
Computation strategies (Split.py of the reference): ``no_checkpoint``, ``checkpoint`` and
``Split(platename, split_size)``.

In the reference Split is a sequential loop over chunks of one plate on one device
(logpq.py:43-57).  Here the same chunking is also the multi-GPU shard axis: with
``Split(..., shard=True)`` under an initialised ``torch.distributed`` process group each rank
evaluates only its own chunks and the per-rank partial log-marginals -- a [K_parents...] tensor --
are combined with ONE ``all_reduce(SUM)`` (RCCL over xGMI on MI355X).  It is a plain sum, not a
log-sum-exp, because independent plate elements multiply (logpq.py:149-153).
"""
import warnings

import os

import torch as t
import torch.distributed as dist

from .dims import PT, ExpPT, Dim, dims_of, is_tensor


class NoSplit:
    platename = None
    shard = False

    def split_args(self, name, sample, inputs_params, extra_log_factors, data, all_platedims):
        return [dict(sample=sample, inputs_params=inputs_params, extra_log_factors=extra_log_factors,
                     data=data, all_platedims=all_platedims)]


class NoCheckpoint(NoSplit):
    pass


class Checkpoint(NoSplit):
    pass


no_checkpoint = NoCheckpoint()
checkpoint = Checkpoint()


def chunk_sizes(orig_size, split_size):
    """Chunk sizes along the plate (Split.py:84-95): full chunks, then the remainder; a remainder of 1
    borrows one element from the previous chunk when split_size > 2."""
    assert orig_size > split_size
    sizes = (orig_size // split_size) * [split_size]
    rem = orig_size % split_size
    if rem:
        sizes.append(rem)
    if split_size > 2 and sizes[-1] == 1:
        sizes[-2] -= 1
        sizes[-1] += 1
    if sizes[-1] == 1:
        warnings.warn("Split produced a chunk of size 1")
    return sizes


MERGE_CHUNKS = True
"""A rank evaluates its block of consecutive Split chunks as ONE contiguous slice of the plate (one set of launches)
instead of looping over them.  The reference chunks a plate to bound memory on a 24 GB card (docs/source/downstream/
computation_strategy.rst:12-22, examples/run_movielens.sh); the sum over chunks (logpq.py:151-153) is the plate sum
of the union, so the result is the same up to fp re-association, and on 288 GB of HBM the whole movielens K=100
plate (a 1.2 GB factor, when it is materialised at all) fits many times over.  ``Split(..., merge=False)`` -- or this
switch -- restores the reference's per-chunk loop."""

MERGE_MAX_BYTES = 16 << 30
"""Split's memory-bounding contract (Split.py, computation_strategy.rst) under merging: where the merged slice would make
the engine allocate any single tensor larger than this -- a materialised [plate, K, K, K] factor that the fused plate
step did not take, its saved log-sum-exp for a backward -- the plate is evaluated chunk by chunk as the reference does
(logpq.logPQ_plate catches engine.TooLargeForMergedSplit and starts the plate over).  The fused route never
materialises the factor, so the BASELINE configurations all merge."""


def rank_block(n_chunks, world, rank):
    """Contiguous block of chunk indices for ``rank`` (balanced to within one chunk)."""
    base, extra = divmod(n_chunks, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class Split:
    """``computation_strategy=Split(platename, split_size)``; ``shard=True`` distributes the chunks over
    the ranks of ``group`` (default: the world group) -- see module docstring.  ``merge`` (default: the module
    switch MERGE_CHUNKS): evaluate a rank's consecutive chunks as one slice."""

    def __init__(self, platename, split_size, shard=False, group=None, merge=None):
        assert isinstance(platename, str)
        assert isinstance(split_size, int)
        self.platename, self.split_size = platename, split_size
        self.shard, self.group = bool(shard), group
        self.merge = merge

    def merging(self):
        return MERGE_CHUNKS if self.merge is None else bool(self.merge)

    def split_args(self, name, sample, inputs_params, extra_log_factors, data, all_platedims, merge=None):
        """``merge``: override of ``merging()`` for this call (False: the plate did not fit as one merged slice)."""
        whole = dict(sample=sample, inputs_params=inputs_params, extra_log_factors=extra_log_factors,
                     data=data, all_platedims=all_platedims)
        if name != self.platename:
            return [whole]
        orig = all_platedims[self.platename]
        sizes = chunk_sizes(orig.size, self.split_size)
        self.last_sizes = list(sizes)
        if self.merging() if merge is None else merge:
            if not self.sharded():
                self.last_sizes = [orig.size]
                return [whole]                       # one rank: its block is the whole plate
            world = dist.get_world_size(self.group)
            self._check_ranks(len(sizes), world)
            sizes = [sum(sizes[i] for i in rank_block(len(sizes), world, r)) for r in range(world)]
            self.last_sizes = list(sizes)
        new_dims = [Dim(f"{self.platename}_split_{i}", s) for i, s in enumerate(sizes)]

        def split_tree(tree):
            outs = [{} for _ in sizes]
            for k, v in tree.items():
                if isinstance(v, dict):
                    for o, sub in zip(outs, split_tree(v)):
                        o[k] = sub
                elif isinstance(v, PT):
                    assert v.has(orig), f"{k} lacks the plate dim {orig} being split"
                    ax = v.ids.index(id(orig))
                    lazy = isinstance(v, ExpPT) and not v.materialised
                    for o, piece, nd in zip(outs, (v.raw if lazy else v.x).split(sizes, ax), new_dims):
                        o[k] = (ExpPT if lazy else PT)(piece, (*v.dims[:ax], nd, *v.dims[ax + 1:]))
                else:
                    assert is_tensor(v)
                    assert orig in set(dims_of(v)), f"{k} lacks the plate dim {orig} being split"
                    others = [d for d in dims_of(v) if d is not orig]
                    pos = v.order(orig, *others)
                    for o, piece, nd in zip(outs, pos.split(sizes, 0), new_dims):
                        o[k] = piece[(nd, *others)]
            return outs

        trees = {key: split_tree(whole[key]) for key in ("sample", "inputs_params", "extra_log_factors", "data")}
        return [dict(sample=trees["sample"][i], inputs_params=trees["inputs_params"][i],
                     extra_log_factors=trees["extra_log_factors"][i], data=trees["data"][i],
                     all_platedims={**all_platedims, self.platename: new_dims[i]})
                for i in range(len(sizes))]

    # ---- multi-GPU ------------------------------------------------------------------------
    def sharded(self):
        return self.shard and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def my_chunks(self, n_chunks):
        """Contiguous block of chunk indices for this rank (balanced to within one chunk)."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        self._check_ranks(n_chunks, world)
        return rank_block(n_chunks, world, rank)

    @staticmethod
    def _check_ranks(n_chunks, world):
        if n_chunks < world:
            raise Exception(f"Split produces {n_chunks} chunks but there are {world} ranks; choose a smaller "
                            "split_size so that every rank gets at least one chunk")


ALL_REDUCES = [0]        # [how many so far, then (group, numel, dtype) of the latest ones, newest last]


class _AllReduceSum(t.autograd.Function):
    """Sum of per-rank partial log-marginals.

    Backward: everything downstream of the sum is REPLICATED on every rank, everything upstream is
    this rank's shard.  Scaling the incoming gradient by the world size makes the usual data-parallel
    convention exact: AVERAGE parameter gradients over ranks (DistributedDataParallel, or
    all_reduce(AVG)) and you get d ELBO / d theta of the unsharded model -- sharded terms are summed,
    replicated terms (computed identically W times) are not over-counted."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.world = dist.get_world_size(group)
        if one_shot_takes(x, ctx.world):
            return exchange_for(group).sum(x.contiguous())   # (one launch of the library; nothing of RCCL's in a capture)
        x = x.contiguous().clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
        ALL_REDUCES.append((group, x.numel(), x.dtype))      # (what a captured sharded evaluation's graph may hold)
        ALL_REDUCES[0] += 1
        del ALL_REDUCES[1:-64]
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.world, None


ONE_SHOT_EXCHANGE = os.environ.get("ALAN_AMD_ONE_SHOT", "0") == "1"
"""The ranks' partials summed by the library's one-shot exchange (alan_exchange_sum: every rank writes its partial into
every peer's inbox over xGMI and adds what arrived, one launch) instead of RCCL's all_reduce.  Off by default: written
for the ranks of ONE node and tested with processes sharing one GPU -- the arithmetic and the protocol, not the fabric
(no multi-GPU node in reach of this repository's tests); RCCL stays the transport until it has been timed on one."""

ONE_SHOT_CAPACITY = int(os.environ.get("ALAN_AMD_ONE_SHOT_CAPACITY", str(1 << 16)))      # fp32 elements per partial

_EXCHANGES = {}
_EXCHANGE_LANE = [0]     # which of a group's exchanges is meant: an EvalPipeline's lanes each take their own (their exchanges
                         # are then issued on different streams, in an order that may differ between the ranks -- one inbox
                         # per lane keeps every inbox's sequence the same on all ranks)


def one_shot_takes(x, world):
    from . import native as N
    return ONE_SHOT_EXCHANGE and x.is_cuda and x.dtype == t.float32 and 1 <= x.numel() <= ONE_SHOT_CAPACITY \
        and 2 <= world <= N.EXCHANGE_MAX_RANKS


def exchange_for(group=None):
    """The group's exchange; the first call (every rank makes it together, outside any stream capture) allocates the
    inboxes and carries their IPC handles between the ranks with the group's own all_gather_object."""
    from . import native as N
    key = (id(group) if group is not None else None, _EXCHANGE_LANE[0])
    ex = _EXCHANGES.get(key)
    if ex is None:
        if t.cuda.is_current_stream_capturing():
            raise Exception("alan_amd: the one-shot exchange has to be set up before a stream capture: evaluate the "
                            "sharded ELBO once eagerly first")
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def carry(mine):
            got = [None] * world
            dist.all_gather_object(got, mine, group=group)
            return got
        ex = N.Exchange(world, rank, ONE_SHOT_CAPACITY, carry)
        # (the library's device code is loaded at its first launch -- hundreds of milliseconds a peer's bounded wait
        # should not be spent on: one trivial launch here, then the barrier)
        scratch = t.zeros(4, dtype=t.int64, device=t.device("cuda", t.cuda.current_device()))
        N.check(N.lib().alan_noise_handon(scratch.data_ptr(), scratch.data_ptr() + 16, N.current_stream(scratch.device)),
                "alan_noise_handon")
        t.cuda.synchronize()
        dist.barrier(group=group)                         # (every rank has opened every inbox: deliveries may start)
        _EXCHANGES[key] = ex
    return ex


def close_exchanges():
    """Frees the inboxes (call it on every rank, after a barrier: a peer may still be writing otherwise)."""
    for ex in _EXCHANGES.values():
        ex.close()
    _EXCHANGES.clear()


def all_reduce_sum(lp, group=None):
    """all_reduce(SUM) of a PT (or torchdim tensor); dims are matched across ranks by name."""
    from .dims import pt_align
    if isinstance(lp, PT):
        ds = sorted(lp.dims, key=str)
        pos = pt_align(lp, tuple(id(d) for d in ds))
        return PT(_AllReduceSum.apply(pos, group), ds)
    ds = sorted(dims_of(lp), key=str)
    pos = lp.order(*ds) if ds else lp
    out = _AllReduceSum.apply(pos, group)
    return out[tuple(ds)] if ds else out
