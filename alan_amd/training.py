"""
HIP-graph replay of a whole training iteration -- the loop of examples/basic_runner.py:81-112 of the
reference (``sample = prob.sample(K)`` -> ``elbo_vi | elbo_rws`` -> ``(-elbo).backward()`` -> ``opt.step()``).

Eagerly an iteration is a few milliseconds of Python driving ~150 small kernels; captured once, every
later iteration is one graph launch: fresh particles are drawn each replay (PyTorch registers the
generator's Philox state with the graph), the forward and backward alan_reduce launches, the fused
producers and the optimizer update all replay on the device.
"""
import ctypes as C
import warnings

import torch as t

from .split import no_checkpoint


class GraphContainsMemsetNodes(RuntimeError):
    pass


def memset_nodes(graph):
    """Number of memset nodes in a captured graph (``torch.cuda.CUDAGraph(keep_graph=True)``), or None when the handle
    cannot be inspected.  On this ROCm a replay that starts on an idle GPU mis-executes torch's multi-block
    reductions, whose semaphore buffer is cleared by a hipMemsetAsync captured as a memset node -- the ONLY structural
    difference between failing and passing graphs (tools/graph_race_probe.py: one linear chain either way; the
    reduction is wrong when the host synchronises between replays, right back to back).  alan_amd's own kernels never
    need one; a captured graph holding any is refused (GraphedStep / GraphedEval) or not used (Sample.elbo_nograd)."""
    try:
        raw = graph.raw_cuda_graph()
        hip = C.CDLL("libamdhip64.so")
        n = C.c_size_t(0)
        if hip.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) != 0:
            return None
        nodes = (C.c_void_p * max(1, n.value))()
        hip.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n))
        count = 0
        for nd in nodes[: n.value]:
            ty = C.c_int(-1)
            hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(ty))
            count += ty.value == 2                      # hipGraphNodeTypeMemset
        return count
    except Exception:
        return None


def node_kinds(graph):
    """(kernel nodes, nodes of any other kind but empty ones) of a captured graph, or None when it cannot be inspected."""
    try:
        raw = graph.raw_cuda_graph()
        hip = C.CDLL("libamdhip64.so")
        n = C.c_size_t(0)
        if hip.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) != 0:
            return None
        nodes = (C.c_void_p * max(1, n.value))()
        if hip.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n)) != 0:
            return None
        kernels = others = 0
        for nd in nodes[: n.value]:
            ty = C.c_int(-1)
            if hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(ty)) != 0:
                return None
            if ty.value == 0:                           # hipGraphNodeTypeKernel
                kernels += 1
            elif ty.value != 5:                         # (hipGraphNodeTypeEmpty joins branches: no work)
                others += 1
        return kernels, others
    except Exception:
        return None


class GraphCannotBeInspected(GraphContainsMemsetNodes):
    pass


_COLLECTIVE_MEMSETS = {}


def collective_memset_nodes(group, numel, dtype, device):
    """Memset nodes in the capture of ONE all_reduce of this size on this group (RCCL's own; cached), or None when that
    graph cannot be inspected: what a sharded evaluation's graph may hold per collective and no more."""
    import torch.distributed as dist
    key = (id(group), numel, dtype, str(device))
    if key not in _COLLECTIVE_MEMSETS:
        x = t.zeros(numel, dtype=dtype, device=device)
        side = t.cuda.Stream()
        side.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(side):
            dist.all_reduce(x, group=group)                 # (communicator set-up outside the capture)
        t.cuda.current_stream().wait_stream(side)
        t.cuda.synchronize()
        g = t.cuda.CUDAGraph(keep_graph=True)
        with t.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(x, group=group)
        _COLLECTIVE_MEMSETS[key] = memset_nodes(g)
    return _COLLECTIVE_MEMSETS[key]


def check_no_memset_nodes(graph, what, allow=False, expected=0):
    """Fails CLOSED: a graph that cannot be inspected is refused like one that holds memset nodes.  ``expected``: the
    memset nodes that belong to the collectives of a sharded evaluation (collective_memset_nodes)."""
    n = memset_nodes(graph)
    if n is None or expected is None:
        msg = (f"{what}: the captured HIP graph could not be inspected for memset nodes (hipGraphGetNodes); replays "
               "of a graph holding a torch multi-block reduction are wrong on this ROCm (DESIGN.md), so it is not used.")
        if not allow:
            raise GraphCannotBeInspected(msg)
        warnings.warn(msg)
        return n
    n -= expected
    if n:
        msg = (f"{what}: the captured HIP graph holds {n} memset node(s) -- a torch multi-block reduction (a sum over a "
               "long leading dim in a model lambda or a non-fused distribution's log_prob?).  Replays that start on an "
               "idle GPU compute such reductions wrongly on this ROCm (DESIGN.md, graph replay and memset nodes).")
        if not allow:
            raise GraphContainsMemsetNodes(msg)
        warnings.warn(msg)
    return n


_STALE_MSG = ("GraphedStep: a parameter still carries the gradient-accumulation node of an earlier forward / backward() "
              "made on another stream -- some tensor of that autograd graph (e.g. the ELBO you called .backward() on, or "
              "the reparameterised Sample it came from) is still alive.  Delete them or build the GraphedStep before any "
              "eager backward on this problem, then try again.")


def stale_grad_accumulators(params):
    """Parameters whose AccumulateGrad node is being kept alive by someone else -- an autograd graph of an earlier
    forward that is still referenced.  Such a node belongs to the stream of that forward; a captured backward would run
    it THERE, outside the capture, and on this ROCm the process then dies when the capture ends (DESIGN.md,
    profiles/r3_graphed_step_crash.txt).  The test is structural: a tensor holds its accumulator weakly, so a node
    fetched, tagged and released is gone at the next fetch unless another graph holds it."""
    import gc
    params = [p for p in params if p.requires_grad]
    with t.enable_grad():                                 # (under no_grad a view has no grad_fn to reach the node through)
        def fetch(p):
            return p.view_as(p).grad_fn.next_functions[0][0]
        for p in params:                                  # tag every node, ...
            fetch(p).metadata["alan_amd_probe"] = True
        gc.collect()                                      # ... ONE collection, ...
        held = []
        for p in params:                                  # ... and whoever still carries the tag is held by another graph
            node = fetch(p)
            if node.metadata.pop("alan_amd_probe", False):
                held.append(p)
            del node
    return held


_UNSAFE_SKIP_STALE_CHECK = False
"""Debugging only (tools/graphed_step_crash_probe.py, one recorded run: profiles/r3_graphed_step_crash.txt): with it set,
constructing a GraphedStep over a live autograd graph is known to end in SIGSEGV inside hipStreamEndCapture."""


class GraphedStep:
    """``step = GraphedStep(problem, K, optimizer, method="vi"); elbo = step()``.

    The optimizer must be graph-capturable (e.g. ``torch.optim.Adam(..., capturable=True)``); parameters
    are updated in place, so ``problem`` can be used normally between / after steps.  As in the
    reference's runner, "rws" expects an optimizer over ``problem.Q.parameters()`` built with
    ``maximize=True`` (the loss is ``-elbo`` for both methods)."""

    def __init__(self, problem, K, optimizer, method="vi", computation_strategy=no_checkpoint, warmup=3,
                 capture_stream=None, allow_memset_nodes=False, unroll=1):
        """``capture_stream``: diagnostics (tools/graph_race_probe.py) -- capture on another stream than the warm-up
        one.  ``allow_memset_nodes``: only warn about memset nodes in the captured graph (see ``memset_nodes``).
        ``unroll``: that many consecutive iterations per captured graph -- a replay then runs them all and returns their
        ELBOs as a vector (a graph launch leaves the GPU idle for several microseconds whatever it holds: tools/replay_trace.sh;
        with the draws' generator state on the device every inner iteration has its own particles)."""
        if unroll < 1:
            raise Exception("unroll must be at least 1")
        self.unroll = unroll
        if method not in ("vi", "rws"):
            raise Exception("method must be 'vi' or 'rws'")
        if problem.device.type != "cuda":
            raise Exception("GraphedStep needs the Problem on the GPU")
        if warmup < 1:
            raise Exception("GraphedStep needs at least one warm-up iteration (allocator and lazy initialisation must "
                            "happen outside the capture)")
        self.problem, self.K, self.opt, self.method = problem, K, optimizer, method
        self.strategy = computation_strategy
        held = stale_grad_accumulators(problem.parameters())
        if held and not _UNSAFE_SKIP_STALE_CHECK:
            raise RuntimeError(_STALE_MSG + f"  (parameters: {len(held)})")
        side = t.cuda.Stream()
        side.wait_stream(t.cuda.current_stream())
        import warnings
        warn_always = t.is_warn_always_enabled()
        t.set_warn_always(True)                        # (the stream-mismatch warning below is a warn-once one)
        try:
            with t.cuda.stream(side), warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                for _ in range(warmup):                # allocator / lazy-init warm-up outside capture
                    self._iteration()
        finally:
            t.set_warn_always(warn_always)
        t.cuda.current_stream().wait_stream(side)
        t.cuda.synchronize()
        stale = [w for w in seen if "AccumulateGrad node's stream does not match" in str(w.message)]
        for w in seen:
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale and not _UNSAFE_SKIP_STALE_CHECK:
            # (second net, by torch's own stream-mismatch warning during the warm-up: the structural check above is the
            # guard proper and does not depend on this text)
            raise RuntimeError(_STALE_MSG)
        self.graph = t.cuda.CUDAGraph(keep_graph=True)
        self.opt.zero_grad(set_to_none=True)
        # Capture on the SAME stream the warm-up ran on: a parameter's AccumulateGrad node remembers the stream it
        # was first used on and autograd runs it there; a different capture stream would put the gradient
        # accumulation on a parallel branch of the graph (PyTorch warns "AccumulateGrad node's stream does not match").
        from . import native as N
        from . import engine as E
        from . import sample as S
        # the ELBO's launch writes its value to the tensor the backward keeps AND through a result ring (a "mirror"):
        # no copy of the graph's output buffer after a replay; the library calls are recorded while they are captured
        self.ring = E.ResultRing.create(problem.device) if (S.RESULT_RING and unroll == 1) else None
        if self.ring is not None:
            self.ring.mirror = True
        rec = N.CallList() if S.DIRECT_REPLAY else None
        E._RING[0], N._REC[0] = self.ring, rec
        try:
            with N.own_graph_noise(problem.device) as self.noise, \
                    t.cuda.graph(self.graph, stream=side if capture_stream is None else capture_stream,
                                 capture_error_mode="thread_local"):
                if unroll == 1:
                    self.elbo = self._iteration()
                else:
                    self.elbo = t.stack([self._iteration() for _ in range(unroll)])
                self.noise.finish_capture()
        finally:
            E._RING[0], N._REC[0] = None, None
        self.n_memset_nodes = check_no_memset_nodes(self.graph, "GraphedStep", allow_memset_nodes)
        if self.ring is not None:
            if self.ring.taken == 1:
                self.ring.sync_position()
            else:
                self.ring = None                       # (the ELBO's launch did not take it: the buffer is copied out)
        # (sample.DIRECT_REPLAY: an iteration that is library launches alone -- the draws, the log-prob producers, the
        # contraction, its backward, alan_amd.Adam -- is re-issued from its recorded launch list instead of replayed as a
        # graph; torch.optim.Adam's kernels, or a model lambda's, keep it a graph replay)
        self.calls = S.calls_if_equivalent(self.graph, rec)

    def _iteration(self):
        self.opt.zero_grad(set_to_none=True)
        sample = self.problem.sample(self.K, reparam=(self.method == "vi"))
        elbo = sample.elbo_vi(self.strategy) if self.method == "vi" else sample.elbo_rws(self.strategy)
        # ascent on the ELBO: the upstream gradient is the constant -1 (no negation kernel, nor its backward's)
        if getattr(self, "_minus_one", None) is None or self._minus_one.dtype != elbo.dtype:
            self._minus_one = t.full((), -1.0, dtype=elbo.dtype, device=elbo.device)
        elbo.backward(self._minus_one)
        self.opt.step()
        return elbo.detach()

    def __call__(self):
        """Replay one iteration; returns a COPY of the ELBO (the graph's own output buffer is overwritten by the next
        replay, so ``[step() for _ in range(n)]`` holds n different values)."""
        self.noise.before_replay()
        slot = self.ring.claim() if self.ring is not None else None
        if self.calls is not None:
            self.calls.replay(t.cuda.current_stream().cuda_stream)
        else:
            self.graph.replay()
        return slot.detach() if slot is not None else self.elbo.clone()


class GraphedEval:
    """``ev = GraphedEval(problem, K); elbo = ev()``: ``problem.sample(K).elbo_nograd()`` with FRESH particles on every
    call -- the quantity the reference's runner times per iteration (examples/basic_runner.py:86-97) -- captured once
    as a HIP graph: the draws (their noise generated inside the launches from a counter that lives on the device:
    native.GraphNoise, kept in step with torch's generator), log-prob producers and the contraction all replay on the
    device."""

    def __init__(self, problem, K, computation_strategy=no_checkpoint, warmup=3, ring=True, unroll=1):
        """``unroll``: that many evaluations (each with its own fresh particles) per captured graph; a call then returns
        their ELBOs as a vector (GraphedStep.__init__ says why)."""
        if problem.device.type != "cuda":
            raise Exception("GraphedEval needs the Problem on the GPU")
        if unroll < 1:
            raise Exception("unroll must be at least 1")
        from . import engine as E
        from . import sample as S
        self.problem, self.K, self.strategy, self.unroll = problem, K, computation_strategy, unroll
        # the evaluation's last launch delivers through a result ring (engine.ResultRing): no copy after a replay
        self.ring = ring if isinstance(ring, E.ResultRing) else \
            (E.ResultRing.create(problem.device) if (ring and S.RESULT_RING and unroll == 1) else None)
        E._RING[0] = self.ring
        from . import native as N
        try:
            side = t.cuda.Stream()
            side.wait_stream(t.cuda.current_stream())
            with t.cuda.stream(side):
                for _ in range(warmup):
                    self._iteration()
            t.cuda.current_stream().wait_stream(side)
            t.cuda.synchronize()
            if self.ring is not None:
                self.ring.taken = 0
            self.graph = t.cuda.CUDAGraph(keep_graph=True)
            rec = N.CallList() if S.DIRECT_REPLAY else None
            N._REC[0] = rec
            try:
                with N.own_graph_noise(problem.device) as self.noise, \
                        t.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                    if unroll == 1:
                        self.elbo = self._iteration()
                    else:
                        self.elbo = t.stack([self._iteration().float() for _ in range(unroll)])
                    self.noise.finish_capture()
            finally:
                N._REC[0] = None
        finally:
            E._RING[0] = None
        self.n_memset_nodes = check_no_memset_nodes(self.graph, "GraphedEval")
        # (sample.DIRECT_REPLAY: the evaluation's library launches issued again one by one where that is all the graph holds)
        self.calls = S.calls_if_equivalent(self.graph, rec)
        if self.ring is not None:
            how = self.ring.settle(self.elbo)
            if how == "recapture":
                self.__init__(problem, K, computation_strategy, warmup, ring=False, unroll=unroll)
            elif how == "copy":
                self.ring = None

    def _iteration(self):
        with t.no_grad():
            return self.problem.sample(self.K, reparam=False).elbo_nograd(self.strategy)

    def __call__(self):
        """Replay; returns this replay's own ELBO tensor (a result-ring slot, or a copy of the graph's output buffer:
        see GraphedStep.__call__)."""
        self.noise.before_replay()
        if self.ring is None:
            self.replay()
            return self.elbo.clone()
        slot = self.ring.claim()
        self.replay()
        return slot.detach()

    def replay(self):
        if self.calls is not None:
            self.calls.replay(t.cuda.current_stream().cuda_stream)
        else:
            self.graph.replay()
