"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
refuses malformed descriptors, and the Python front end refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch as t

from alan_amd import native as N
from alan_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "alan_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(alan_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = N.lib()
    syms = _declared_symbols()
    assert set(syms) == set(N.EXPORTS)
    for s in syms:
        assert getattr(L, s) is not None
    assert L.alan_abi_version() == 14
    assert L.alan_build_target() == b"gfx950"


def test_struct_layout_matches_header():
    # alan_tensor_t: ptr(8) + int32 + float + 8*int64 ; alan_reduce_desc_t per the header
    assert ctypes.sizeof(N.Tensor) == 8 + 4 + 4 + 8 * N.MAX_DIMS
    expect = 4 + 4 + 8 * N.MAX_DIMS + 4 * N.MAX_DIMS + 4 + 4 + ctypes.sizeof(N.Tensor) * (N.MAX_FACTORS + 3) + 8 + 16 + (8 + 8 + 4 + 4)
    expect += 4 + 4 + 8 + 8 + 3 * 8                      # alan_noise_t
    assert ctypes.sizeof(N.Noise) == 48
    assert ctypes.sizeof(N.ReduceDesc) == expect
    assert ctypes.sizeof(N.BackwardDesc) == expect + ctypes.sizeof(N.Tensor) * N.MAX_FACTORS


def test_generated_noise_is_only_for_the_draws_modes():
    """alan_noise_t on a log-sum-exp: a bad descriptor, said before anything touches a GPU."""
    L = N.lib()
    d = N.ReduceDesc()
    d.mode, d.ndim, d.n_factors = N.MODE_LSE, 1, 2
    d.size[0], d.role[0] = 8, N.REDUCE
    dummy = ctypes.c_void_p(0x1000)
    for x in (d.factor[0], d.factor[1], d.out):
        x.data, x.dtype, x.scale = dummy, N.F32, 1.0
    d.factor[0].stride[0] = d.factor[1].stride[0] = 1
    assert L.alan_reduce_check(ctypes.byref(d)) == 0
    d.noise.on = 1
    assert L.alan_reduce_check(ctypes.byref(d)) == -1
    d.mode, d.n_factors, d.role[0] = N.MODE_AFFINE, 3, N.KEEP
    d.factor[2].data, d.factor[2].dtype, d.factor[2].scale = dummy, N.F32, 1.0
    d.factor[2].stride[0] = d.out.stride[0] = 1
    assert L.alan_reduce_check(ctypes.byref(d)) == 0
    d.out.dtype = d.factor[0].dtype = d.factor[1].dtype = d.factor[2].dtype = N.F64      # (fp64: not the small kernel)
    assert L.alan_reduce_check(ctypes.byref(d)) == N.ERR_UNSUPPORTED


def test_backward_rejects_bad_descriptors_and_declines_unsuitable_shapes():
    L = N.lib()
    b = N.BackwardDesc()
    assert L.alan_reduce_backward(None, None, 0, None) == -1
    assert L.alan_reduce_backward(ctypes.byref(b), None, 0, None) == -1      # no weight / lse given
    assert L.alan_reduce_backward_workspace_bytes(ctypes.byref(b)) == 0


def test_fused_plate_step_struct_layout_and_rejections():
    """alan_normal_lse_desc_t / alan_normal_lse_backward_desc_t as the header lays them out (natural alignment), and
    malformed descriptors refused before any GPU work."""
    fwd = 8 + 3 * 8 + 8 + 2 * 8 + 8 + 2 * 8 + 4 + 4 + 4 * 8 + 4 * 8 + 4 * 8 + 5 * 8 + 8 + 2 * 8 + 8 + 8 + 2 * 8 + 8 + 8   # (keep_partials + padding, scale_table)
    assert ctypes.sizeof(N.NormalLseDesc) == fwd
    assert ctypes.sizeof(N.NormalLseBackwardDesc) == fwd + 8 + 8 + 2 * 8 + 4 * 8
    L = N.lib()
    b = N.NormalLseBackwardDesc()
    assert L.alan_normal_lse_backward(None, None, 0, None) == -1
    assert L.alan_normal_lse_backward(ctypes.byref(b), None, 0, None) == -1          # null pointers
    assert L.alan_normal_lse_backward_workspace_bytes(ctypes.byref(b)) == 0
    d = N.NormalLseDesc()
    assert L.alan_normal_lse(ctypes.byref(d), None, 0, None) == -1
    # a well-formed problem the kernels do not take (event length 40): declined, not attempted
    dummy = ctypes.c_void_p(0x1000)
    for x in (b.fwd, d):
        x.value = x.loc = x.scale = x.out = dummy
        x.M, x.NK, x.NL, x.NS, x.E = 4, 5, 3, 3, 40
    b.lse = b.grad_out = b.grad_small = dummy
    assert L.alan_normal_lse_workspace_bytes(ctypes.byref(d)) == 0
    assert L.alan_normal_lse(ctypes.byref(d), dummy, 1 << 20, None) == N.ERR_UNSUPPORTED
    assert L.alan_normal_lse_backward(ctypes.byref(b), dummy, 1 << 20, None) == N.ERR_UNSUPPORTED
    b.fwd.E = d.E = 18
    assert L.alan_normal_lse_workspace_bytes(ctypes.byref(d)) > 0
    assert L.alan_normal_lse_backward_workspace_bytes(ctypes.byref(b)) > 0
    assert L.alan_normal_lse_backward(ctypes.byref(b), dummy, 16, None) == -3         # workspace too small
    b.fwd.n_small = 5
    assert L.alan_normal_lse_backward(ctypes.byref(b), dummy, 1 << 20, None) == -1    # too many small factors
    # the scale table (ABI 14): sized for one tile of scale rows, 0 beyond; a misaligned one is refused; the problem that
    # builds it is checked like any other
    assert L.alan_normal_lse_table_bytes(None) == 0
    d.v_sm, d.v_sk, d.v_se, d.l_sl, d.l_se, d.s_ss, d.s_se = 5 * 18, 18, 1, 18, 1, 18, 1
    assert L.alan_normal_lse_table_bytes(ctypes.byref(d)) == 8 * 64 * 16 + 32 * 4    # E = 18: ten event pairs, eight MFMA steps
    d.NS = 33
    assert L.alan_normal_lse_table_bytes(ctypes.byref(d)) == 0
    d.NS = 3
    d.scale_table = ctypes.c_void_p(0x1004)
    assert L.alan_normal_lse(ctypes.byref(d), dummy, 1 << 20, None) == -1
    r = N.ReduceDesc()
    r.mode, r.ndim, r.n_factors = N.MODE_NORMAL_TABLE, 2, 1
    r.size[0], r.size[1], r.role[0], r.role[1] = 3, 18, N.KEEP, N.REDUCE
    r.factor[0].data, r.factor[0].dtype, r.factor[0].scale = dummy, N.F32, 1.0
    r.factor[0].stride[0], r.factor[0].stride[1] = 18, 1
    assert L.alan_reduce_check(ctypes.byref(r)) == -1                                 # no table to write
    r.out.data, r.out.dtype = dummy, N.F32
    assert L.alan_reduce_check(ctypes.byref(r)) == 0
    assert L.alan_reduce_workspace_bytes(ctypes.byref(r)) == 0
    r.size[0] = 33
    assert L.alan_reduce_check(ctypes.byref(r)) == N.ERR_UNSUPPORTED
    r.size[0], r.role[1] = 3, N.PLATE
    assert L.alan_reduce_check(ctypes.byref(r)) == -1


def test_bad_descriptors_are_rejected_without_touching_the_gpu():
    L = N.lib()
    d = N.ReduceDesc()
    d.ndim = 99
    assert L.alan_reduce(ctypes.byref(d), None, 0, None) == -1
    d = N.ReduceDesc()
    d.ndim = 1
    d.size[0] = 4
    d.role[0] = 7
    assert L.alan_reduce(ctypes.byref(d), None, 0, None) == -1
    d.role[0] = N.PLATE
    d.mode = N.MODE_SUM          # PLATE only valid with LSE
    assert L.alan_reduce(ctypes.byref(d), None, 0, None) == -1
    d.mode = N.MODE_LSE
    d.role[0] = N.REDUCE
    d.n_factors = 1              # null factor pointer
    assert L.alan_reduce(ctypes.byref(d), None, 0, None) == -1
    assert L.alan_chain_logmmexp_batched(None, 0, 1, 4, 3, 0, 9, 3, 1, None, None, None, 0, None) == -1


def test_workspace_query():
    L = N.lib()
    d = N.ReduceDesc()
    d.mode = N.MODE_LSE
    d.ndim = 3
    for i, (s, r) in enumerate([(10, N.PLATE), (3, N.KEEP), (5, N.REDUCE)]):
        d.size[i], d.role[i] = s, r
    d.out.dtype = N.F32
    assert L.alan_reduce_workspace_bytes(ctypes.byref(d)) == 256      # 30 floats, 256-aligned
    d.role[0] = N.KEEP
    assert L.alan_reduce_workspace_bytes(ctypes.byref(d)) == 0
    # every round of the pairwise tree stays in the workspace: 500 + 250 + 125 + 63 + 32 + 16 + 8 + 4 + 2 + 1 nodes
    assert L.alan_chain_batched_workspace_bytes(1, 1000, 30, N.F32) >= 1001 * 30 * 30 * 4
    assert L.alan_chain_batched_workspace_bytes(1, 1000, 30, N.F32) < 1001 * 30 * 30 * 4 + 10 * 256
    assert L.alan_chain_batched_workspace_bytes(7, 1, 5, N.F64) == (7 * 25 * 8 + 255) // 256 * 256
    assert L.alan_chain_backward_batched_workspace_bytes(3, 9, 4, N.F32) == L.alan_chain_batched_workspace_bytes(3, 9, 4, N.F32)


def test_cpu_tensors_are_refused_not_silently_computed():
    x = t.zeros(3, 4)
    with pytest.raises(N.NativeError, match="no CPU fallback"):
        E.reduce_factors([(x, ("a", "k"))], reduce=("k",))
    with pytest.raises(N.NativeError):
        N.chain_logmmexp(t.zeros(4, 3, 3))


def test_planner_single_launch_for_movielens_shapes():
    sizes = {"M": 300, "Ka": 30, "Kb": 30, "Kz": 30}
    steps = E.plan_elimination([("M", "Ka", "Kb", "Kz"), ("M", "Kz")], sizes, ["Kz"])
    assert steps == [((0, 1), ("Kz",))]
    steps = E.plan_elimination([("Ka",), ("Kb",), ("Ka", "Kb")], sizes, ["Ka", "Kb"])
    assert len(steps) == 1 and set(steps[0][1]) == {"Ka", "Kb"} and set(steps[0][0]) == {0, 1, 2}


def test_planner_peels_one_K_when_the_joint_reduction_would_be_one_long_chain():
    """movielens top level at K=100: a[Ka] + b[Kb] + T[Ka,Kb] over both Ks is 10^4 elements into ONE output --
    a single workgroup.  The planner eliminates the innermost K of T first (all three factors in that launch:
    a is constant over Kb), then the other."""
    sizes = {"Ka": 100, "Kb": 100}
    steps = E.plan_elimination([("Ka",), ("Kb",), ("Ka", "Kb")], sizes, ["Ka", "Kb"])
    assert steps == [((0, 1, 2), ("Kb",)), ((3,), ("Ka",))]


def test_planner_chain_is_pairwise_not_joint():
    sizes = {"K1": 30, "K2": 30, "K3": 30, "K4": 30}
    steps = E.plan_elimination([("K1", "K2"), ("K2", "K3"), ("K3", "K4")], sizes, ["K1", "K2", "K3", "K4"])
    # never materialise more than K^2 elements' worth of index space x K
    for ids, now in steps:
        assert len(now) >= 1
    eliminated = [k for _, now in steps for k in now]
    assert sorted(eliminated) == ["K1", "K2", "K3", "K4"]


@pytest.mark.parametrize("n", list(range(2, 26)))
def test_planner_never_exceeds_the_launch_factor_limit(n):
    """n factors on one K (a Group of n/2 variables under elbo_vi: log P and -log Q each): every step, pre-adds
    included, takes at most MAX_FACTORS factors, LSE steps one fewer (their backward appends the saved lse)."""
    steps = E.plan_elimination([("K",)] * (n - 1) + [("K", "p")], {"K": 10, "p": 4}, ["K"])
    used = [i for ids, _ in steps for i in ids]
    assert sorted(used) == list(range(len(used)))                 # every factor and intermediate consumed once
    for ids, now in steps:
        assert len(ids) <= (N.MAX_FACTORS - 1 if now else N.MAX_FACTORS)
    assert [now for _, now in steps if now] == [("K",)]


def test_planner_empty_Ks():
    assert E.plan_elimination([("T", "K")], {"T": 10, "K": 3}, []) == []
    steps = E.plan_elimination([("T", "K"), ("K",)], {"T": 10, "K": 3}, [])
    assert steps == [((0, 1), ())]


@pytest.mark.gpu
def test_graft_entry_smoke_runs():
    """The driver's smoke() entry point (one small plate step and one chain, each checked against the oracle)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    importlib.import_module("__graft_entry__").smoke()


def test_new_entry_points_reject_malformed_arguments_without_touching_the_gpu():
    L = N.lib()
    assert L.alan_reduce_batch(None, 0, None) == -1
    one = (ctypes.POINTER(N.ReduceDesc) * 1)(ctypes.POINTER(N.ReduceDesc)())
    assert L.alan_reduce_batch(one, 1, None) == -1                        # a null descriptor in the list
    assert L.alan_chain_logmmexp_terms_final(None, None, 1, None, None, N.F32, 1, 4, 3, None, None, None, 0, None) == -1
    ptrs = (ctypes.c_void_p * 1)(None)
    st = (ctypes.c_int64 * 4)(0, 9, 3, 1)
    assert L.alan_chain_logmmexp_terms_final(ptrs, st, 1, None, None, N.F32, 1, 4, 3, None, None, None, 0, None) == -1     # null term
    assert L.alan_chain_logmmexp_terms_final(ptrs, st, 4, None, None, N.F32, 1, 4, 3, None, None, None, 0, None) == -1     # > 3 terms
    assert L.alan_chain_logmmexp_batched(None, N.F32, 2, 4, 3, 36, 9, 3, 1, None, None, None, 0, None) == -1
    assert L.alan_chain_logmmexp_backward_batched(None, N.F32, 2, 4, 3, 36, 9, 3, 1, None, None, None, None, None, None,
                                                  0, None) == -1
    assert L.alan_normal_lse(None, None, 0, None) == -1
    assert L.alan_normal_lse_workspace_bytes(None) == 0
    assert L.alan_chain_batched_workspace_bytes(0, 4, 3, N.F32) == 0
    d = N.ReduceDesc()
    d.mode = N.MODE_PRODUCER_GRAD                 # which gradient? factor[0].scale must say (1..4)
    d.ndim, d.n_factors = 1, 4
    d.size[0], d.role[0] = 4, N.KEEP
    assert L.alan_reduce(ctypes.byref(d), None, 0, None) == -1


def test_launch_lists_and_pipelines_reject_misuse_without_touching_the_gpu():
    """alan_calls_* / alan_pipeline_*: an empty list records and counts nothing; a pipeline needs lanes that hold launches."""
    L = N.lib()
    h = ctypes.c_void_p()
    assert L.alan_calls_create(ctypes.byref(h)) == 0
    assert L.alan_calls_count(h) == 0
    assert L.alan_calls_end(h) == -1                     # (not being recorded)
    assert L.alan_calls_begin(h) == 0
    h2 = ctypes.c_void_p()
    assert L.alan_calls_create(ctypes.byref(h2)) == 0
    assert L.alan_calls_begin(h2) == -1                  # (one list at a time per thread)
    assert L.alan_calls_end(h2) == -1
    assert L.alan_calls_end(h) == 0
    assert L.alan_calls_count(h) == 0
    pipe = ctypes.c_void_p()
    lanes = (ctypes.c_void_p * 2)(h, h2)
    assert L.alan_pipeline_create(lanes, 2, 2, ctypes.byref(pipe)) == -1     # (lanes without launches)
    assert L.alan_pipeline_create(lanes, 0, 0, ctypes.byref(pipe)) == -1
    assert L.alan_pipeline_create(lanes, 99, 0, ctypes.byref(pipe)) == -1
    assert L.alan_pipeline_submit(None, 1) == -1 and L.alan_pipeline_join(None, None) == -1
    assert L.alan_calls_destroy(h) == 0 and L.alan_calls_destroy(h2) == 0


def test_documented_deviations_are_the_switches_actual_defaults():
    """alan_amd/__init__.py lists every place where a drop-in user sees something other than the reference as
    ``module.SWITCH`` = default | reference value: each switch exists, holds the documented default, and accepts the value
    that restores the reference's behaviour (own the deviations)."""
    import ast
    import importlib
    import alan_amd
    rows = re.findall(r"``(\w+)\.(\w+)`` = (\S+) \| (\S+)", alan_amd.__doc__)
    assert len(rows) >= 4, rows
    for mod, name, default, ref in rows:
        m = importlib.import_module(f"alan_amd.{mod}")
        assert hasattr(m, name), f"alan_amd.{mod}.{name} is documented but does not exist"
        assert getattr(m, name) == ast.literal_eval(default), (mod, name, getattr(m, name), default)
        ast.literal_eval(ref)
    assert ("posterior", "TIMESERIES_POSTERIOR", '"reference"', '"reference"') in rows      # (the reference's draws by default)
