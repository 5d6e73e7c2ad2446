"""The private / semi-private PyTorch surfaces the host side leans on, each pinned by a named test so that a PyTorch
upgrade which moves or re-words one of them fails HERE, by name, rather than as a silent slow path or a wrong guard
(robustness 14).  Tested against torch 2.10; every use site degrades as stated below if its test fails."""
import gc
import warnings

import pytest
import torch as t


def test_storage_use_count_counts_views():
    """engine._use_count (ResultRing: is a ring slot's storage still referenced?) -- torch._C._storage_Use_Count."""
    from alan_amd import engine as E
    x = t.zeros(4)
    base = E._use_count(x)
    y = x.detach()
    assert E._use_count(x) == base + 1
    del y
    assert E._use_count(x) == base


def test_functorch_batching_primitives_are_where_dist_expects_them():
    """dist._nested_vmap: torch._C._functorch's batch-dim primitives (else it falls back to torch.vmap, 10 x slower per
    level) -- and they still compose to what nested torch.vmap computes."""
    from alan_amd import dist as D
    assert D._add_batch_dim is not None, "torch._C._functorch lost _add_batch_dim & co: dist falls back to torch.vmap"
    a, b = t.arange(6.).reshape(2, 3), t.arange(4.)
    out = D._nested_vmap(lambda x, y: x.sum() + y, [a, b], [(0,), (1,)], [0, 1], {0: 2, 1: 4})
    want = t.vmap(t.vmap(lambda x, y: x.sum() + y, in_dims=(None, 0)), in_dims=(0, None))(a, b)
    assert t.equal(out, want)


def test_grad_accumulator_is_reachable_and_weakly_held():
    """training.stale_grad_accumulators (GraphedStep's guard): a leaf's AccumulateGrad node is reachable through a view's
    grad_fn, carries a metadata dict, and dies with its last holder."""
    from alan_amd.training import stale_grad_accumulators
    p = t.nn.Parameter(t.ones(3))
    assert stale_grad_accumulators([p]) == []
    loss = (p * 2).sum()                      # a live autograd graph holds p's accumulator
    assert stale_grad_accumulators([p]) == [p]
    del loss
    gc.collect()
    assert stale_grad_accumulators([p]) == []


@pytest.mark.gpu
def test_sync_debug_mode_warns_with_the_word_the_auto_promotion_looks_for():
    """sample._auto_eval keeps an evaluation eager if it synchronises: it recognises torch's sync-debug warning by the
    substring "synchroniz"."""
    x = t.ones(4, device="cuda")
    mode = t.cuda.get_sync_debug_mode()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        t.cuda.set_sync_debug_mode("warn")
        try:
            float(x.sum())
        finally:
            t.cuda.set_sync_debug_mode(mode)
    msgs = [str(w.message) for w in seen if "prototype feature" not in str(w.message)]
    assert any("synchroniz" in m.lower() for m in msgs), msgs


@pytest.mark.gpu
def test_cuda_generator_offset_can_be_read_and_moved_from_the_host():
    """dist._noise_source / native.GraphNoise: torch.cuda.default_generators[i].initial_seed() / get_offset() /
    set_offset() -- the draws key their in-launch noise by them and move the offset as torch's own kernels would; a
    captured graph's raw handle, pool and node list (sample.calls_if_equivalent, training.node_kinds)."""
    gen = t.cuda.default_generators[t.cuda.current_device()]
    t.manual_seed(123)
    assert gen.initial_seed() == 123 and gen.get_offset() == 0
    t.randn(10, device="cuda")
    moved = gen.get_offset()
    assert moved > 0 and moved % 4 == 0
    gen.set_offset(moved + 64)
    assert gen.get_offset() == moved + 64
    from alan_amd.training import node_kinds
    x = t.zeros(8, device="cuda")
    s = t.cuda.Stream()
    with t.cuda.stream(s):
        x.add_(1.0)
        t.cuda.synchronize()
        g = t.cuda.CUDAGraph(keep_graph=True)
        with t.cuda.graph(g, stream=s):
            x.add_(1.0)
            x.mul_(2.0)
    assert node_kinds(g) == (2, 0) and g.pool() is not None
