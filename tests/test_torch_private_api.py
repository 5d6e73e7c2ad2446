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
