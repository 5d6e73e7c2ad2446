"""
HIP-graph replay of a whole training iteration -- the loop of examples/basic_runner.py:81-112 of the
reference (``sample = prob.sample(K)`` -> ``elbo_vi | elbo_rws`` -> ``(-elbo).backward()`` -> ``opt.step()``).

Eagerly an iteration is a few milliseconds of Python driving ~150 small kernels; captured once, every
later iteration is one graph launch: fresh particles are drawn each replay (PyTorch registers the
generator's Philox state with the graph), the forward and backward alan_reduce launches, the fused
producers and the optimizer update all replay on the device.
"""
import torch as t

from .split import no_checkpoint


class GraphedStep:
    """``step = GraphedStep(problem, K, optimizer, method="vi"); elbo = step()``.

    The optimizer must be graph-capturable (e.g. ``torch.optim.Adam(..., capturable=True)``); parameters
    are updated in place, so ``problem`` can be used normally between / after steps.  As in the
    reference's runner, "rws" expects an optimizer over ``problem.Q.parameters()`` built with
    ``maximize=True`` (the loss is ``-elbo`` for both methods)."""

    def __init__(self, problem, K, optimizer, method="vi", computation_strategy=no_checkpoint, warmup=3):
        if method not in ("vi", "rws"):
            raise Exception("method must be 'vi' or 'rws'")
        if problem.device.type != "cuda":
            raise Exception("GraphedStep needs the Problem on the GPU")
        self.problem, self.K, self.opt, self.method = problem, K, optimizer, method
        self.strategy = computation_strategy
        side = t.cuda.Stream()
        side.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(side):
            for _ in range(warmup):                    # allocator / lazy-init warm-up outside capture
                self._iteration()
        t.cuda.current_stream().wait_stream(side)
        t.cuda.synchronize()
        self.graph = t.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        # Capture on the SAME stream the warm-up ran on: a parameter's AccumulateGrad node remembers the stream it
        # was first used on and autograd runs it there; a different capture stream would put the gradient
        # accumulation on a parallel branch of the graph (PyTorch warns "AccumulateGrad node's stream does not match").
        with t.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.elbo = self._iteration()

    def _iteration(self):
        self.opt.zero_grad(set_to_none=True)
        sample = self.problem.sample(self.K, reparam=(self.method == "vi"))
        elbo = sample.elbo_vi(self.strategy) if self.method == "vi" else sample.elbo_rws(self.strategy)
        (-elbo).backward()
        self.opt.step()
        return elbo.detach()

    def __call__(self):
        """Replay one iteration; returns a COPY of the ELBO (the graph's own output buffer is overwritten by the next
        replay, so ``[step() for _ in range(n)]`` holds n different values)."""
        self.graph.replay()
        return self.elbo.clone()


class GraphedEval:
    """``ev = GraphedEval(problem, K); elbo = ev()``: ``problem.sample(K).elbo_nograd()`` with FRESH particles on every
    call -- the quantity the reference's runner times per iteration (examples/basic_runner.py:86-97) -- captured once
    as a HIP graph: sampling kernels (the generator's Philox state is registered with the graph), log-prob producers
    and the contraction all replay on the device."""

    def __init__(self, problem, K, computation_strategy=no_checkpoint, warmup=3):
        if problem.device.type != "cuda":
            raise Exception("GraphedEval needs the Problem on the GPU")
        self.problem, self.K, self.strategy = problem, K, computation_strategy
        side = t.cuda.Stream()
        side.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(side):
            for _ in range(warmup):
                self._iteration()
        t.cuda.current_stream().wait_stream(side)
        t.cuda.synchronize()
        self.graph = t.cuda.CUDAGraph()
        with t.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.elbo = self._iteration()

    def _iteration(self):
        with t.no_grad():
            return self.problem.sample(self.K, reparam=False).elbo_nograd(self.strategy)

    def __call__(self):
        """Replay; returns a COPY of the ELBO (see GraphedStep.__call__)."""
        self.graph.replay()
        return self.elbo.clone()
