"""
ctypes binding of libalan_mi355.so (C ABI in include/alan_mi355.h).

There is deliberately no fallback: if the shared library is missing, or a tensor is not on a
HIP device, the call raises.  PyTorch is used only for device memory and the current stream.
"""
import contextlib
import threading
import ctypes as C
import os

import torch as t

MAX_DIMS = 8
MAX_FACTORS = 6

F32, F64 = 0, 1
KEEP, REDUCE, PLATE, DOT, PRESUM = 0, 1, 2, 3, 4
MODE_LSE, MODE_SUM, MODE_WEXPSUM, MODE_NORMAL, MODE_BERNOULLI, MODE_NORMAL_LOGSCALE = 0, 1, 2, 3, 4, 5
MODE_PRODUCER_GRAD = 6
MODE_BERNOULLI_LINEAR = 7
MODE_DOT = 8
MODE_BERNOULLI_LINEAR_GRAD = 9
MODE_NORMAL_TABLE = 11
MODE_AFFINE = 10
MODE_FUSED_FWD, MODE_FUSED_BWD = 100, 101      # (KernelTimer record tags of alan_normal_lse / _backward; not library modes)
GRAD_VALUE, GRAD_LOC, GRAD_SCALE, GRAD_LOGITS = 1.0, 2.0, 3.0, 4.0      # factor[0].scale of a MODE_PRODUCER_GRAD call

_STATUS = {
    -1: "bad descriptor",
    -2: "unsupported dtype/size",
    -3: "workspace too small",
    -4: "kernel launch failed",
}

# (ALAN_AMD_LIB: another build of the same library, e.g. the diagnostic one of `make TIMELINE=1`)
LIB_PATH = os.environ.get("ALAN_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libalan_mi355.so")


class NativeError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("scale", C.c_float),
        ("stride", C.c_int64 * MAX_DIMS),
    ]


class Noise(C.Structure):
    _fields_ = [("on", C.c_int32), ("advance_by", C.c_uint32), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("cell", C.c_void_p), ("receipt", C.c_void_p), ("advance", C.c_void_p)]


class ReduceDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("ndim", C.c_int32),
        ("size", C.c_int64 * MAX_DIMS),
        ("role", C.c_int32 * MAX_DIMS),
        ("n_factors", C.c_int32),
        ("factor", Tensor * MAX_FACTORS),
        ("weight", Tensor),
        ("out", Tensor),
        ("lse_out", Tensor),
        ("add_const", C.c_double),
        ("ev_start", C.c_void_p),
        ("ev_stop", C.c_void_p),
        ("ring_slots", C.c_void_p),
        ("ring_counter", C.c_void_p),
        ("ring_n", C.c_int32), ("ring_and_out", C.c_int32),
        ("noise", Noise),
    ]


class ChainNormal(C.Structure):
    _fields_ = [("value", C.c_void_p), ("loc", C.c_void_p), ("scale", C.c_void_p),
                ("v_stride", C.c_int64 * 4), ("l_stride", C.c_int64 * 4), ("s_stride", C.c_int64 * 4),
                ("loc_mul", C.c_double), ("log_scale", C.c_int32), ("loc0", C.c_void_p), ("l0_stride", C.c_int64 * 4)]


class ChainFinal(C.Structure):
    _fields_ = [("n_extra", C.c_int32), ("extra", C.c_void_p * 3), ("stride", C.c_int64 * 3), ("add_const", C.c_double),
                ("out", C.c_void_p), ("ring_slots", C.c_void_p), ("ring_counter", C.c_void_p), ("ring_n", C.c_int32)]


class BackwardDesc(C.Structure):
    _fields_ = [("fwd", ReduceDesc), ("grad", Tensor * MAX_FACTORS)]


class NormalLseDesc(C.Structure):
    _fields_ = [("value", C.c_void_p), ("v_sm", C.c_int64), ("v_sk", C.c_int64), ("v_se", C.c_int64),
                ("loc", C.c_void_p), ("l_sl", C.c_int64), ("l_se", C.c_int64),
                ("scale", C.c_void_p), ("s_ss", C.c_int64), ("s_se", C.c_int64),
                ("log_scale", C.c_int32), ("n_small", C.c_int32),
                ("small", C.c_void_p * 4), ("small_sm", C.c_int64 * 4), ("small_sk", C.c_int64 * 4),
                ("M", C.c_int64), ("NK", C.c_int64), ("NL", C.c_int64), ("NS", C.c_int64), ("E", C.c_int64),
                ("out", C.c_void_p), ("o_sl", C.c_int64), ("o_ss", C.c_int64),
                ("lse_out", C.c_void_p), ("add_const", C.c_double),
                ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p), ("keep_partials", C.c_int32),
                ("scale_table", C.c_void_p)]


class NormalLseBackwardDesc(C.Structure):
    _fields_ = [("fwd", NormalLseDesc), ("lse", C.c_void_p),
                ("grad_out", C.c_void_p), ("g_sl", C.c_int64), ("g_ss", C.c_int64),
                ("grad_value", C.c_void_p), ("grad_loc", C.c_void_p), ("grad_scale", C.c_void_p),
                ("grad_small", C.c_void_p)]


ADAM_MAX_TENSORS = 24


class AdamDesc(C.Structure):
    _fields_ = [("n_tensors", C.c_int32), ("maximize", C.c_int32),
                ("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS),
                ("numel", C.c_int64 * ADAM_MAX_TENSORS),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("step", C.c_void_p), ("ticket", C.c_void_p)]


_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `make -C alan_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "alan_amd has no CPU or PyTorch fallback for the reduce_Ks hot path.")
        L = C.CDLL(LIB_PATH)
        L.alan_reduce.restype = C.c_int
        L.alan_reduce.argtypes = [C.POINTER(ReduceDesc), C.c_void_p, C.c_size_t, C.c_void_p]
        L.alan_reduce_workspace_bytes.restype = C.c_size_t
        L.alan_reduce_workspace_bytes.argtypes = [C.POINTER(ReduceDesc)]
        L.alan_reduce_check.restype = C.c_int
        L.alan_reduce_check.argtypes = [C.POINTER(ReduceDesc)]
        L.alan_reduce_batch.restype = C.c_int
        L.alan_reduce_batch.argtypes = [C.POINTER(C.POINTER(ReduceDesc)), C.c_int32, C.c_void_p]
        L.alan_reduce_backward.restype = C.c_int
        L.alan_reduce_backward.argtypes = [C.POINTER(BackwardDesc), C.c_void_p, C.c_size_t, C.c_void_p]
        L.alan_reduce_backward_workspace_bytes.restype = C.c_size_t
        L.alan_reduce_backward_workspace_bytes.argtypes = [C.POINTER(BackwardDesc)]
        L.alan_normal_lse.restype = C.c_int
        L.alan_normal_lse.argtypes = [C.POINTER(NormalLseDesc), C.c_void_p, C.c_size_t, C.c_void_p]
        L.alan_normal_lse_workspace_bytes.restype = C.c_size_t
        L.alan_normal_lse_workspace_bytes.argtypes = [C.POINTER(NormalLseDesc)]
        L.alan_normal_lse_table_bytes.restype = C.c_size_t
        L.alan_normal_lse_table_bytes.argtypes = [C.POINTER(NormalLseDesc)]
        L.alan_normal_lse_n_partials.restype = C.c_int64
        L.alan_normal_lse_n_partials.argtypes = [C.POINTER(NormalLseDesc)]
        L.alan_normal_lse_backward.restype = C.c_int
        L.alan_normal_lse_backward.argtypes = [C.POINTER(NormalLseBackwardDesc), C.c_void_p, C.c_size_t, C.c_void_p]
        L.alan_normal_lse_backward_workspace_bytes.restype = C.c_size_t
        L.alan_normal_lse_backward_workspace_bytes.argtypes = [C.POINTER(NormalLseBackwardDesc)]
        L.alan_chain_batched_workspace_bytes.restype = C.c_size_t
        L.alan_chain_batched_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32]
        L.alan_chain_logmmexp_batched.restype = C.c_int
        L.alan_chain_logmmexp_batched.argtypes = [C.c_void_p, C.c_int32, *([C.c_int64] * 7), C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_size_t, C.c_void_p]
        L.alan_chain_logmmexp_terms_final.restype = C.c_int
        L.alan_chain_logmmexp_terms_final.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32,
                                                      C.POINTER(ChainNormal), C.POINTER(ChainFinal), C.c_int32, C.c_int64,
                                                      C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                      C.c_void_p]
        L.alan_chain_backward_batched_workspace_bytes.restype = C.c_size_t
        L.alan_chain_backward_batched_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32]
        L.alan_chain_logmmexp_backward_batched.restype = C.c_int
        L.alan_chain_logmmexp_backward_batched.argtypes = [C.c_void_p, C.c_int32, *([C.c_int64] * 7),
                                                           *([C.c_void_p] * 6), C.c_size_t, C.c_void_p]
        L.alan_chain_messages.restype = C.c_int
        L.alan_chain_messages.argtypes = [C.c_void_p, *([C.c_int64] * 7), C.c_void_p, C.c_void_p]
        L.alan_chain_sample.restype = C.c_int
        L.alan_chain_sample.argtypes = [C.c_void_p, *([C.c_int64] * 6), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_void_p, *([C.c_int64] * 4), C.c_void_p, C.c_void_p]
        L.alan_chain_filter.restype = C.c_int
        L.alan_chain_filter.argtypes = [C.c_void_p, *([C.c_int64] * 7), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.alan_exchange_create.restype = C.c_int
        L.alan_exchange_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]
        L.alan_exchange_connect.restype = C.c_int
        L.alan_exchange_connect.argtypes = [C.c_void_p, C.c_char_p]
        L.alan_exchange_sum.restype = C.c_int
        L.alan_exchange_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.alan_exchange_status.restype = C.c_int
        L.alan_exchange_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.alan_exchange_destroy.restype = C.c_int
        L.alan_exchange_destroy.argtypes = [C.c_void_p]
        L.alan_calls_create.restype = C.c_int
        L.alan_calls_create.argtypes = [C.POINTER(C.c_void_p)]
        L.alan_calls_begin.restype = C.c_int
        L.alan_calls_begin.argtypes = [C.c_void_p]
        L.alan_calls_end.restype = C.c_int
        L.alan_calls_end.argtypes = [C.c_void_p]
        L.alan_calls_count.restype = C.c_int64
        L.alan_calls_count.argtypes = [C.c_void_p]
        L.alan_noise_handon.restype = C.c_int
        L.alan_noise_handon.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.alan_pipeline_create.restype = C.c_int
        L.alan_pipeline_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.alan_pipeline_submit.restype = C.c_int
        L.alan_pipeline_submit.argtypes = [C.c_void_p, C.c_int64]
        L.alan_pipeline_join.restype = C.c_int
        L.alan_pipeline_join.argtypes = [C.c_void_p, C.c_void_p]
        L.alan_pipeline_fence.restype = C.c_int
        L.alan_pipeline_fence.argtypes = [C.c_void_p, C.c_void_p]
        L.alan_pipeline_destroy.restype = C.c_int
        L.alan_pipeline_destroy.argtypes = [C.c_void_p]
        L.alan_calls_replay.restype = C.c_int
        L.alan_calls_replay.argtypes = [C.c_void_p, C.c_void_p]
        L.alan_calls_destroy.restype = C.c_int
        L.alan_calls_destroy.argtypes = [C.c_void_p]
        L.alan_adam_step.restype = C.c_int
        L.alan_adam_step.argtypes = [C.POINTER(AdamDesc), C.c_void_p]
        L.alan_abi_version.restype = C.c_int
        L.alan_build_target.restype = C.c_char_p
        _lib = L
    return _lib


EXPORTS = ("alan_reduce", "alan_reduce_check", "alan_reduce_workspace_bytes", "alan_reduce_batch", "alan_reduce_backward",
           "alan_reduce_backward_workspace_bytes", "alan_normal_lse", "alan_normal_lse_workspace_bytes",
           "alan_normal_lse_n_partials", "alan_normal_lse_table_bytes", "alan_normal_lse_backward", "alan_normal_lse_backward_workspace_bytes",
           "alan_chain_batched_workspace_bytes", "alan_chain_logmmexp_batched", "alan_chain_logmmexp_terms_final",
           "alan_chain_backward_batched_workspace_bytes", "alan_chain_logmmexp_backward_batched",
           "alan_chain_messages", "alan_chain_sample", "alan_chain_filter",
           "alan_exchange_create", "alan_exchange_connect", "alan_exchange_sum", "alan_exchange_status",
           "alan_exchange_destroy",
           "alan_calls_create", "alan_calls_begin", "alan_calls_end", "alan_calls_count", "alan_calls_replay",
           "alan_calls_destroy", "alan_noise_handon",
           "alan_pipeline_create", "alan_pipeline_submit", "alan_pipeline_join", "alan_pipeline_fence",
           "alan_pipeline_destroy", "alan_adam_step",
           "alan_abi_version", "alan_build_target")


class CallList:
    """The library calls of one evaluation, recorded while it is captured into a HIP graph (sample._GraphedELBO) and
    issued again by one call (alan_calls_replay) where that is the same work as replaying the graph.  ``spoiled``: a
    library launch the list cannot hold was made (the graph is replayed then)."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().alan_calls_create(C.byref(h)), "alan_calls_create")
        self._h, self.n, self.spoiled, self.keep = h, 0, False, []

    def record(self, fn, *args):
        """The library call fn(*args) -- one that was just issued for real -- once more, with its launches kept in this
        list instead of issued (alan_calls_begin .. alan_calls_end)."""
        L = lib()
        if L.alan_calls_begin(self._h) != 0:
            self.spoiled = True
            return
        rc = fn(*args)
        if L.alan_calls_end(self._h) != 0 or rc != 0:
            self.spoiled = True
        self.n += 1

    def launches(self):
        return int(lib().alan_calls_count(self._h))

    def replay(self, stream):
        check(lib().alan_calls_replay(self._h, stream), "alan_calls_replay")

    def __del__(self):
        try:
            if self._h is not None:
                lib().alan_calls_destroy(self._h)
                self._h = None
        except Exception:
            pass


_REC = [None]            # the CallList being recorded (sample._GraphedELBO sets it around its capture pass)
_TRACE = [None]          # a list collecting what an evaluation did (Sample.explain): launches and model-lambda routes

_MODE_NAMES = {0: "LSE", 1: "SUM", 2: "WEXPSUM", 3: "NORMAL", 4: "BERNOULLI", 5: "NORMAL_LOGSCALE", 6: "PRODUCER_GRAD",
               7: "BERNOULLI_LINEAR", 8: "DOT", 9: "BERNOULLI_LINEAR_GRAD", 10: "AFFINE",
               11: "NORMAL_TABLE (the scale table of the fused plate step that follows)"}


def trace(kind, what, **info):
    """One line of Sample.explain()'s report (nothing unless a report is being collected)."""
    if _TRACE[0] is not None:
        _TRACE[0].append({"kind": kind, "what": what, **info})


def _desc_info(desc):
    sizes = [int(desc.size[i]) for i in range(desc.ndim)]
    roles = [int(desc.role[i]) for i in range(desc.ndim)]
    red = 1
    out = 1
    for s, r in zip(sizes, roles):
        if r in (REDUCE, 3):
            red *= s
        elif r in (KEEP, PLATE) and r == KEEP:
            out *= s
    return {"mode": _MODE_NAMES.get(int(desc.mode), str(int(desc.mode))), "factors": int(desc.n_factors), "outputs": out,
            "reduced_per_output": red, "plate_dims": sum(r == PLATE for r in roles), "presum": any(r == PRESUM for r in roles),
            "result_ring": bool(desc.ring_n)}


def _spoil():
    if _REC[0] is not None:
        _REC[0].spoiled = True

EXCHANGE_MAX_RANKS = 8
EXCHANGE_HANDLE_BYTES = 64


class Exchange:
    """This rank's end of the one-shot sum of per-rank partials (include/alan_mi355.h: alan_exchange_*).  `carry` takes
    this rank's handle bytes and returns every rank's, rank-major (the caller's transport between the processes)."""

    def __init__(self, world, rank, capacity, carry):
        L = lib()
        if not 1 <= world <= EXCHANGE_MAX_RANKS:
            raise NativeError(f"alan_amd: a one-shot exchange takes up to {EXCHANGE_MAX_RANKS} ranks, not {world}")
        buf = C.create_string_buffer(EXCHANGE_HANDLE_BYTES)
        h = C.c_void_p()
        rc = L.alan_exchange_create(world, rank, capacity, buf, C.byref(h))
        if rc != 0:
            raise NativeError(f"alan_exchange_create failed ({rc})")
        self._h, self.world, self.rank, self.capacity = h, world, rank, capacity
        handles = carry(buf.raw)
        if len(handles) != world or any(len(x) != EXCHANGE_HANDLE_BYTES for x in handles):
            raise NativeError("alan_amd: the exchange's handles did not come back one per rank")
        rc = L.alan_exchange_connect(h, b"".join(handles))
        if rc != 0:
            raise NativeError(f"alan_exchange_connect failed ({rc}): are all ranks on one node, HSA_ENABLE_IPC_MODE_LEGACY=0?")

    def sum(self, x):
        """Sum over ranks of the fp32 device tensor x (contiguous), as a new tensor; enqueued on the current stream."""
        if not (x.is_cuda and x.dtype == t.float32 and x.is_contiguous() and 1 <= x.numel() <= self.capacity):
            raise NativeError("alan_amd: the one-shot exchange takes a contiguous fp32 device tensor within its capacity")
        out = t.empty_like(x)
        rc = lib().alan_exchange_sum(self._h, x.data_ptr(), out.data_ptr(), x.numel(),
                                     t.cuda.current_stream(x.device).cuda_stream)
        if rc != 0:
            raise NativeError(f"alan_exchange_sum failed ({rc})")
        if _REC[0] is not None:
            _REC[0].keep.append((x, out))
            _REC[0].record(lib().alan_exchange_sum, self._h, x.data_ptr(), out.data_ptr(), x.numel(), None)
        return out

    def status(self):
        """(exchanges completed on this rank, number of the first exchange that timed out or 0); synchronises."""
        t.cuda.synchronize()
        done, bad = C.c_uint32(), C.c_uint32()
        rc = lib().alan_exchange_status(self._h, C.byref(done), C.byref(bad))
        if rc != 0:
            raise NativeError(f"alan_exchange_status failed ({rc})")
        return done.value, bad.value

    def close(self):
        if self._h is not None:
            lib().alan_exchange_destroy(self._h)
            self._h = None


def dtype_code(dtype):
    if dtype == t.float32:
        return F32
    if dtype == t.float64:
        return F64
    raise NativeError(f"alan_amd: unsupported dtype {dtype} (float32/float64 only)")


def require_device(x, what="tensor"):
    if not x.is_cuda:
        raise NativeError(
            f"alan_amd: {what} is on {x.device}; the reduce_Ks hot path runs only as HIP kernels on an "
            "MI355X (no CPU fallback). Move the Problem to the GPU with problem.to('cuda').")


def check(status, what):
    if status != 0:
        raise NativeError(f"{what} failed: {_STATUS.get(status, status)}")


def fill_tensor(dst, x, strides, scale=1.0):
    dst.data = x.data_ptr()
    dst.dtype = dtype_code(x.dtype)
    dst.scale = scale
    for i, s in enumerate(strides):
        dst.stride[i] = s


def current_stream(device):
    return t.cuda.current_stream(device).cuda_stream


_TIMER = [None]     # set by profiling.KernelTimer


# ---- deferred small launches -------------------------------------------------------------------------------------
# The per-variable log-prob producers of a plate are independent, launch-latency-bound kernels (4-5 us each even
# inside a replayed HIP graph).  Inside ``deferring()`` (one gradient-free evaluation) a producer issued under
# ``may_defer()`` is only queued; the queue goes out as ONE multi-problem launch (alan_reduce_batch) as soon as
# anything else is launched, or at the end.  Stream order is unchanged, so this is invisible to every later alan
# launch; it is NOT invisible to a torch op that reads a queued output -- hence the explicit ``may_defer()`` at the
# call sites whose result only ever feeds another alan launch (logpq.py), and ``flush()`` for anyone in doubt.
class _Queue(threading.local):
    """Per-thread state: evaluations running in different Python threads keep separate queues."""

    def __init__(self):
        self.pending = []        # [(desc, device, keepalive tensors)]
        self.depth = [0, 0]      # nesting of deferring() / may_defer()
        self.chain = None        # a single timeseries chain waiting for its launch (_PendingChain): the final contraction may join it
        self.tail_ok = 0         # nesting of tail_attach(): launches of an evaluation's final contraction


_Q = _Queue()
DEFER_SMALL_LAUNCHES = True
_POISON_QUEUED = os.environ.get("ALAN_AMD_POISON_QUEUED") == "1"   # fill queued launches' outputs with NaN until they run


def queue_active():
    """Inside native.deferring() (one gradient-free evaluation)?"""
    return _Q.depth[0] > 0


def n_pending():
    return len(_Q.pending)


@contextlib.contextmanager
def deferring():
    _Q.depth[0] += 1
    try:
        yield
        if _Q.depth[0] == 1:
            flush()
    finally:
        _Q.depth[0] -= 1
        if _Q.depth[0] == 0:
            _Q.pending.clear()          # (only non-empty after an exception)
            _Q.chain = None


class GraphNoise:
    """The generator state of ONE captured graph whose launches generate their noise (alan_noise_t.cell): a ring of
    {counter, seed} slots on the device, one per captured launch that draws; each such launch reads its own slot and
    writes the next one's (the last writes the first's: the ring is closed when the capture has ended), so every replay
    draws fresh noise with no kernel in front of it and nothing inside it waiting.  Kept in step with torch's generator
    from the host: before a replay the first slot is rewritten if the generator was re-seeded or used by anyone else since
    the last one (a 16-byte copy, otherwise nothing), and the generator's offset is moved past what the replay will
    consume -- the replays draw what the same iterations launched one by one would."""
    MAX_LAUNCHES = 64

    def __init__(self, device):
        self.device = device
        self.state = t.zeros(3 * self.MAX_LAUNCHES, dtype=t.int64, device=device)     # slots [MAX][2], then the ring's addresses
        self.n = 0                   # captured launches that draw
        self.handon = False          # the single-launch hand-on has been arranged (carried by a batch, or a launch of its own)
        self.per_replay = 0          # counter increments of one replay (summed while capturing)
        self._expect = None

    def next_launch(self):
        """(cell, advance) of the next captured launch that draws, or None when the ring is full (its last address word
        belongs to a carried hand-on)."""
        if self.n >= self.MAX_LAUNCHES - 1:
            return None
        j, self.n = self.n, self.n + 1
        return self.state[2 * j:2 * j + 2], self.state[2 * self.MAX_LAUNCHES + j:2 * self.MAX_LAUNCHES + j + 1]

    def release_last(self, inc):
        """The most recent next_launch() taken back: no launch reads or writes that slot (dist._release_unused_slot)."""
        if self.n > 0:
            self.n -= 1
            self.per_replay -= inc

    def carry(self, items):
        """A batch of queued launches issued inside the capture, behind the only launch that drew so far: its first
        small-problem launch also copies slot 1 back to slot 0 (alan_noise_t.on = 2) -- the hand-on without a launch of
        its own.  (Should more launches draw later on, the last of them writes slot 0 after this copy: the ring is whole
        either way.)"""
        L = lib()
        for d, dev, _ in items:
            if dev.type != self.device.type:
                return
            z = d.noise
            z.on, z.cell, z.advance = 2, self.state.data_ptr() + 16, self.state.data_ptr() + 8 * (3 * self.MAX_LAUNCHES - 1)
            if L.alan_reduce_check(C.byref(d)) == 0 and d.mode != MODE_BERNOULLI_LINEAR:
                self.handon = True
                return
            z.on, z.cell, z.advance = 0, None, None

    def finish_capture(self):
        """Still inside the capture, behind everything that draws.  A single launch that draws cannot hand its state on
        to its own slot (its other workgroups may not have read it yet): it hands on to a second slot, and a one-thread
        launch here copies that back to the first."""
        if self.n == 1 and not self.handon:
            L = lib()
            a, b = self.state.data_ptr(), self.state.data_ptr() + 16
            check(L.alan_noise_handon(b, a, current_stream(self.device)), "alan_noise_handon")
            if _REC[0] is not None:
                _REC[0].record(L.alan_noise_handon, b, a, None)
            self.handon = True

    def close(self):
        """After the capture: launch j hands on to slot j + 1, the last one to slot 0 (a single one: to slot 1, which
        finish_capture's launch copies back)."""
        if self.n == 1 and not self.handon:
            raise NativeError("alan_amd: GraphNoise.finish_capture() was not called inside the capture")
        if self.n:
            base = self.state.data_ptr()
            self.state[3 * self.MAX_LAUNCHES - 1:3 * self.MAX_LAUNCHES].copy_(t.tensor([base], dtype=t.int64))   # (carry's target: slot 0)
            ring = [base + 16 * ((j + 1) % max(self.n, 2)) for j in range(self.n)]
            self.state[2 * self.MAX_LAUNCHES:2 * self.MAX_LAUNCHES + self.n].copy_(t.tensor(ring, dtype=t.int64))
            t.cuda.current_stream().synchronize()       # (whatever stream the replays are issued on: the ring is there)

    def before_replay(self):
        if not self.per_replay:
            return
        gen = t.cuda.default_generators[self.device.index if self.device.index is not None else t.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        if (seed, off) != self._expect:
            self.state[0:2].copy_(t.tensor([off, seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=t.int64))
        gen.set_offset(off + self.per_replay)
        self._expect = (seed, off + self.per_replay)


_GRAPH_NOISE = [None]


def graph_noise(device):
    """The state of the graph being captured (own_graph_noise), or None."""
    st = _GRAPH_NOISE[0]
    if st is None or st.device.type != device.type:
        return None
    cur = t.cuda.current_device()
    same = (st.device.index if st.device.index is not None else cur) == (device.index if device.index is not None else cur)
    return st if same else None


@contextlib.contextmanager
def own_graph_noise(device):
    """Around the capture of one graph that draws samples: its launches take their noise from a state of its own."""
    saved, _GRAPH_NOISE[0] = _GRAPH_NOISE[0], GraphNoise(device)
    mine = _GRAPH_NOISE[0]
    try:
        yield mine
    finally:
        _GRAPH_NOISE[0] = saved
    mine.close()                                         # (the capture has ended -- the context managers nest that way)


def fused_pending():
    return _Q.chain is not None


CHAIN_FINAL = os.environ.get("ALAN_AMD_CHAIN_FINAL", "1") != "0"
"""A single timeseries chain (one-wave-per-product kernel: fp32, 12 < K <= 32) is queued like a fused plate step, and the
evaluation's final contraction -- a log-sum-exp over [K_init] vectors, one of them the chain's result -- runs behind the
last round of the chain's last launch, which is ONE workgroup: no hand-off between workgroups, one launch fewer
(alan_chain_logmmexp_terms_final)."""


class _PendingChain:
    def __init__(self, launch, vec, K, device, keepalive):
        self.launch, self.vec, self.K, self.device, self.keepalive = launch, vec, K, device, keepalive

    def try_final(self, desc, device, keepalive):
        """The alan_reduce call `desc` as this chain's final contraction, if it is one.  True: launched together."""
        if device != self.device or desc.mode != MODE_LSE or desc.weight.data or desc.lse_out.data \
                or desc.out.dtype != F32 or not (1 <= desc.n_factors <= 4):
            return False
        big = [i for i in range(desc.ndim) if desc.size[i] > 1]
        if len(big) != 1 or desc.role[big[0]] != REDUCE or desc.size[big[0]] != self.K:
            return False
        k = big[0]
        fin, mine = ChainFinal(), 0
        for f in range(desc.n_factors):
            fac = desc.factor[f]
            if fac.dtype != F32 or fac.scale != 1.0:
                return False
            if fac.data == self.vec.data_ptr() and fac.stride[k] == 1:
                mine += 1
                continue
            if fin.n_extra == 3 or fac.stride[k] < 0:
                return False
            fin.extra[fin.n_extra], fin.stride[fin.n_extra] = fac.data, fac.stride[k]
            fin.n_extra += 1
        if mine != 1:
            return False
        fin.add_const, fin.out = desc.add_const, desc.out.data
        fin.ring_slots, fin.ring_counter, fin.ring_n = desc.ring_slots, desc.ring_counter, desc.ring_n
        _Q.chain = None
        self.keepalive = (self.keepalive, keepalive)
        if not self.launch(fin):                          # (the library declined: the chain alone, the caller goes on)
            self.launch(None)
            return False
        return True


@contextlib.contextmanager
def tail_attach():
    """Around the launches of an evaluation's FINAL contraction: nothing but the caller reads their results, and it does
    so only after the enclosing deferring() has flushed."""
    _Q.tail_ok += 1
    try:
        yield
    finally:
        _Q.tail_ok -= 1
        if _Q.tail_ok == 0:
            flush()


@contextlib.contextmanager
def may_defer():
    _Q.depth[1] += 1
    try:
        yield
    finally:
        _Q.depth[1] -= 1


def ride_along(desc, device, keepalive=()):
    """A small problem nobody asked to wait for (the scale table of a fused plate step): it joins the queued launches if
    there are any -- one more workgroup of a launch that goes out anyway -- and is NOT launched otherwise (False)."""
    if not (_Q.depth[0] and _Q.pending and _Q.pending[0][1] == device and _TIMER[0] is None and not t.is_grad_enabled()):
        return False
    if desc.noise.on or lib().alan_reduce_check(C.byref(desc)) != 0:
        return False
    _Q.pending.append((desc, device, keepalive))
    return True


def flush():
    """Issue every queued launch now (in order)."""
    if _Q.chain is not None:
        c, _Q.chain = _Q.chain, None
        c.launch(None)
    if not _Q.pending:
        return
    items = list(_Q.pending)
    _Q.pending.clear()
    _flush_items(items)


def _flush_items(items):
    if not items:
        return
    L = lib()
    device = items[0][1]
    rec = _REC[0]
    st = _GRAPH_NOISE[0]
    if st is not None and st.n == 1 and not st.handon and not any(d.noise.on for d, _, _ in items):
        st.carry(items)                                   # (the single launch that drew hands on through this batch's launch)
    if _TRACE[0] is not None:
        trace("launch", "alan_reduce" if len(items) == 1 else "alan_reduce_batch (one multi-problem launch per eight small problems)",
              problems=[_desc_info(d) for d, _, _ in items])
    if len(items) == 1:
        rc = L.alan_reduce(C.byref(items[0][0]), None, 0, current_stream(device))
        check(rc, "alan_reduce")
        if rec is not None:
            rec.record(L.alan_reduce, C.byref(items[0][0]), None, 0, None)
        return
    arr = (C.POINTER(ReduceDesc) * len(items))(*[C.pointer(d) for d, _, _ in items])
    rc = L.alan_reduce_batch(arr, len(items), current_stream(device))
    check(rc, "alan_reduce_batch")
    if rec is not None:
        rec.record(L.alan_reduce_batch, arr, len(items), None)


def run_reduce(desc, device, algo_bytes=0, keepalive=()):
    """Enqueue one alan_reduce call.  False only for a call the library may decline -- one with a result ring
    (desc.ring_n), of mode BERNOULLI_LINEAR, or with a PRESUM dim -- when it does: nothing was enqueued, the caller takes
    its other route."""
    L = lib()
    presum = any(desc.role[i] == PRESUM for i in range(desc.ndim))
    lin_grad = desc.mode == MODE_BERNOULLI_LINEAR_GRAD
    if (desc.mode == MODE_BERNOULLI_LINEAR or presum or lin_grad or desc.noise.on) and \
            L.alan_reduce_check(C.byref(desc)) == ERR_UNSUPPORTED:
        return False
    if desc.noise.on and _Q.pending and any(d.noise.on and _noise_key(d) != _noise_key(desc) for d, _, _ in _Q.pending):
        flush()                                     # (one generator per alan_reduce_batch call)
    if (DEFER_SMALL_LAUNCHES and not presum and not lin_grad and _Q.depth[0] and _Q.depth[1] and _TIMER[0] is None and not t.is_grad_enabled()
            and not desc.ring_n
            and L.alan_reduce_workspace_bytes(C.byref(desc)) == 0
            and (not _Q.pending or _Q.pending[0][1] == device)):
        if _POISON_QUEUED and len(keepalive) > 1 and keepalive[1] is not None:
            keepalive[1].fill_(float("nan"))       # debugging aid: a premature read of a queued output shows up as NaN
        _Q.pending.append((desc, device, keepalive))
        if len(_Q.pending) >= 16:
            flush()
        return True
    if _Q.chain is not None and _Q.tail_ok and _TIMER[0] is None and not t.is_grad_enabled() and \
            _Q.chain.try_final(desc, device, keepalive):
        return True
    flush()
    if _TIMER[0] is not None and not desc.noise.on:     # (a launch that generates noise carries no timing events)
        _TIMER[0].attach(desc, algo_bytes)
    nbytes = L.alan_reduce_workspace_bytes(C.byref(desc))
    ws = t.empty(nbytes, dtype=t.uint8, device=device) if nbytes else None
    rc = L.alan_reduce(C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes,
                       current_stream(device))
    if rc == ERR_UNSUPPORTED and desc.ring_n:
        return False
    check(rc, "alan_reduce")
    if _TRACE[0] is not None:
        trace("launch", "alan_reduce", problems=[_desc_info(desc)], workspace_bytes=int(nbytes))
    if _REC[0] is not None:
        _REC[0].keep.append(ws)
        _REC[0].record(L.alan_reduce, C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes, None)
    return True


ERR_UNSUPPORTED = -2


def _noise_key(desc):
    z = desc.noise
    return (z.seed, z.cell, z.receipt, z.advance, z.advance_by)


def run_reduce_backward(desc, device):
    """All gradients of an LSE call in one pass (alan_reduce_backward).  False when the problem does not fit
    the streaming kernel -- the caller then falls back to one WEXPSUM launch per factor."""
    L = lib()
    flush()
    nbytes = L.alan_reduce_backward_workspace_bytes(C.byref(desc))
    ws = t.empty(nbytes, dtype=t.uint8, device=device) if nbytes else None
    rc = L.alan_reduce_backward(C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes,
                                current_stream(device))
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc, "alan_reduce_backward")
    trace("launch", "alan_reduce_backward (all gradients of a log-sum-exp in one pass)", problems=[_desc_info(desc.fwd)])
    if _REC[0] is not None:
        _REC[0].keep.append(ws)
        _REC[0].record(L.alan_reduce_backward, C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes, None)
    return True


def run_normal_lse(desc, device, keepalive=()):
    """The fused plate step (alan_normal_lse).  False when the library declines the shape."""
    L = lib()
    flush()
    nbytes = L.alan_normal_lse_workspace_bytes(C.byref(desc))
    if nbytes == 0:
        return False
    if _TIMER[0] is not None:
        _TIMER[0].attach(desc, 0, mode=MODE_FUSED_FWD, flops=2.0 * desc.M * desc.NK * desc.NL * desc.NS * (desc.E + 1))
    ws = None if desc.keep_partials else t.empty(nbytes, dtype=t.uint8, device=device)     # (the partials ARE the output)
    rc = L.alan_normal_lse(C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes, current_stream(device))
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc, "alan_normal_lse")
    trace("launch", "alan_normal_lse (fused plate step: Normal producer + log-sum-exp + plate sum, the factor never written)",
          M=int(desc.M), K_child=int(desc.NK), loc_rows=int(desc.NL), scale_rows=int(desc.NS), event=int(desc.E),
          small_factors=int(desc.n_small), partial_slices_kept=bool(desc.keep_partials),
          scale_table_prebuilt=bool(desc.scale_table))
    if _REC[0] is not None:
        _REC[0].keep.append(ws)
        _REC[0].record(L.alan_normal_lse, C.byref(desc), ws.data_ptr() if ws is not None else None, nbytes, None)
    return True


def run_normal_lse_backward(desc, device):
    """Every gradient of the fused plate step in one pass (alan_normal_lse_backward).  False when the library
    declines the shape."""
    L = lib()
    flush()
    nbytes = L.alan_normal_lse_backward_workspace_bytes(C.byref(desc))
    if nbytes == 0:
        return False
    if _TIMER[0] is not None:
        f = desc.fwd
        # per 32 x 32 tile: D recomputed over the E + 1 event slots, then the V and U products (32 rows summed each)
        _TIMER[0].attach(f, 0, mode=MODE_FUSED_BWD, flops=2.0 * f.M * f.NK * f.NL * f.NS * ((f.E + 1) + 32 + 32))
    ws = t.empty(nbytes, dtype=t.uint8, device=device)
    rc = L.alan_normal_lse_backward(C.byref(desc), ws.data_ptr(), nbytes, current_stream(device))
    if rc == ERR_UNSUPPORTED:
        return False
    check(rc, "alan_normal_lse_backward")
    trace("launch", "alan_normal_lse_backward (every gradient of the fused plate step in one pass)")
    if _REC[0] is not None:
        _REC[0].keep.append(ws)
        _REC[0].record(L.alan_normal_lse_backward, C.byref(desc), ws.data_ptr(), nbytes, None)
    return True


CHAIN_MAX_BATCH = 65535      # the batch rides on gridDim.y


def chain_logmmexp(ms, want_chain=False):
    """ms: [T,K,K] or a batch [B,T,K,K] (device tensor) -> (vec[(B,)K], chain[(B,)K,K] or None, tree).
    ``tree`` holds every round of the reference's pairwise tree (what the backward walks)."""
    require_device(ms, "timeseries factor")
    L = lib()
    flush()
    batched = ms.ndim == 4
    m4 = ms if batched else ms.unsqueeze(0)
    B, T, K, K2 = m4.shape
    assert K == K2
    if B > CHAIN_MAX_BATCH:
        raise NativeError(f"alan_chain_logmmexp_batched: at most {CHAIN_MAX_BATCH} chains per call, got {B}")
    code = dtype_code(ms.dtype)
    vec = t.empty(B, K, dtype=ms.dtype, device=ms.device)
    chain = t.empty(B, K, K, dtype=ms.dtype, device=ms.device) if want_chain else None
    nbytes = L.alan_chain_batched_workspace_bytes(B, T, K, code)
    tree = t.empty(max(nbytes, 1), dtype=t.uint8, device=ms.device)
    args = (m4.data_ptr(), code, B, T, K, *m4.stride(), chain.data_ptr() if want_chain else None, vec.data_ptr(),
            tree.data_ptr(), nbytes)
    rc = L.alan_chain_logmmexp_batched(*args, current_stream(ms.device))
    check(rc, "alan_chain_logmmexp_batched")
    trace("launch", "alan_chain_logmmexp_batched (timeseries chain)", chains=int(B), T=int(T), K=int(K))
    if _REC[0] is not None:
        _REC[0].keep.append((m4, chain, vec, tree))
        _REC[0].record(L.alan_chain_logmmexp_batched, *args, None)
    if not batched:
        return vec[0], (chain[0] if want_chain else None), tree
    return vec, chain, tree


def chain_logmmexp_terms(terms, normal=None):
    """logsumexp(chain_logmmexp(sum of terms), -1) with the sum taken on load: ``terms`` = up to 3 device tensors
    [B,T,K,K] (expanded / stride-0 views welcome).  ``normal`` = (value, loc, scale, loc_mul, log_scale), three more
    such views: one further term log N(value; loc_mul * loc, scale) computed on load (the transition factor of a
    timeseries, never written).  -> vec [B,K]."""
    L = lib()
    flush()
    assert 1 <= len(terms) <= 3
    B, T, K, K2 = terms[0].shape
    assert K == K2 and all(x.shape == terms[0].shape and x.dtype == terms[0].dtype for x in terms)
    for x in terms:
        require_device(x, "timeseries factor")
    if B > CHAIN_MAX_BATCH:
        raise NativeError(f"alan_chain_logmmexp_terms: at most {CHAIN_MAX_BATCH} chains per call, got {B}")
    code = dtype_code(terms[0].dtype)
    device = terms[0].device
    vec = t.empty(B, K, dtype=terms[0].dtype, device=device)
    nbytes = L.alan_chain_batched_workspace_bytes(B, T, K, code)
    tree = t.empty(max(nbytes, 1), dtype=t.uint8, device=device)
    ptrs = (C.c_void_p * len(terms))(*[x.data_ptr() for x in terms])
    strides = (C.c_int64 * (4 * len(terms)))(*[s for x in terms for s in x.stride()])
    nd = None
    if normal is not None:
        v, l, sc, mul, log_scale, *rest = normal
        l0 = rest[0] if rest else None             # (the location of step 0: then `l` is read one step behind)
        assert all(x.shape == terms[0].shape and x.dtype == terms[0].dtype for x in (v, sc))
        assert l.dtype == terms[0].dtype and l.shape[0] == B and l.shape[2:] == terms[0].shape[2:] and \
            l.shape[1] >= (T - 1 if l0 is not None else T)
        nd = ChainNormal()
        nd.value, nd.loc, nd.scale = v.data_ptr(), l.data_ptr(), sc.data_ptr()
        for q in range(4):
            nd.v_stride[q], nd.l_stride[q], nd.s_stride[q] = v.stride(q), l.stride(q), sc.stride(q)
        nd.loc_mul, nd.log_scale = float(mul), int(bool(log_scale))
        if l0 is not None:
            require_device(l0, "timeseries initial state")
            assert l0.dtype == terms[0].dtype and l0.shape == (B, 1, *terms[0].shape[2:])
            nd.loc0 = l0.data_ptr()
            for q in range(4):
                nd.l0_stride[q] = l0.stride(q)
    def launch(fin):
        rc = L.alan_chain_logmmexp_terms_final(ptrs, strides, len(terms), C.byref(nd) if nd is not None else None,
                                               C.byref(fin) if fin is not None else None, code, B, T, K, None,
                                               vec.data_ptr(), tree.data_ptr(), nbytes, current_stream(device))
        if rc == ERR_UNSUPPORTED and fin is not None:
            return False
        check(rc, "alan_chain_logmmexp_terms_final")
        trace("launch", "alan_chain_logmmexp_terms_final (timeseries chain" + (", Normal transition computed on load" if nd is not None else "")
              + (", the evaluation's final log-sum-exp behind its last round" if fin is not None else "") + ")", chains=int(B), T=int(T), K=int(K),
              terms=len(terms))
        if _REC[0] is not None:
            _REC[0].keep.append((terms, normal, tree, vec))
            _REC[0].record(L.alan_chain_logmmexp_terms_final, ptrs, strides, len(terms),
                           C.byref(nd) if nd is not None else None, C.byref(fin) if fin is not None else None, code, B, T, K,
                           None, vec.data_ptr(), tree.data_ptr(), nbytes, None)
        return True

    if CHAIN_FINAL and _Q.depth[0] and _Q.chain is None and B == 1 and 12 < K <= 32 and terms[0].dtype == t.float32 \
            and _TIMER[0] is None and not t.is_grad_enabled():
        # queued: its result only ever feeds the parent's contraction (logpq._chain_of_terms), which may join its launch
        _Q.chain = _PendingChain(launch, vec, K, device, (terms, normal, tree, ptrs, strides, nd))
        return vec
    launch(None)
    return vec


POSTERIOR_MAX_K = 128


def chain_messages(ms):
    """ms [C,T,K,K] fp32 -> backward messages beta [C,T+1,K] (alan_chain_messages: one launch)."""
    _spoil()
    require_device(ms, "timeseries factor")
    flush()
    C_, T, K, _ = ms.shape
    beta = t.empty(C_, T + 1, K, dtype=t.float32, device=ms.device)
    check(lib().alan_chain_messages(ms.data_ptr(), C_, T, K, *ms.stride(), beta.data_ptr(), current_stream(ms.device)),
          "alan_chain_messages")
    return beta


def chain_sample(ms, beta, init, N, B, chain_of_n, chain_of_b, generator=None):
    """Forward sampling given the messages (alan_chain_sample: one launch).  init: int64 [N, B] (stride 0 welcome);
    sample (n, b) walks chain n * chain_of_n + b * chain_of_b.  -> int64 [N, B, T]."""
    _, T, K, _ = ms.shape
    u = t.rand(N, B, T, dtype=t.float32, device=ms.device, generator=generator)
    out = t.empty(N, B, T, dtype=t.int64, device=ms.device)
    init = init.expand(N, B)
    check(lib().alan_chain_sample(ms.data_ptr(), T, K, *ms.stride(), beta.data_ptr(), init.data_ptr(), init.stride(0),
                                  init.stride(1), u.data_ptr(), N, B, chain_of_n, chain_of_b, out.data_ptr(),
                                  current_stream(ms.device)), "alan_chain_sample")
    return out


def chain_filter(ms, init):
    """ms [C,T,K,K], init int64 [N] -> alpha [C,T,N,K]: the forward recursion from each initial state."""
    _spoil()
    require_device(ms, "timeseries factor")
    flush()
    C_, T, K, _ = ms.shape
    init = init.contiguous()
    alpha = t.empty(C_, T, init.numel(), K, dtype=t.float32, device=ms.device)
    check(lib().alan_chain_filter(ms.data_ptr(), C_, T, K, *ms.stride(), init.data_ptr(), init.numel(), alpha.data_ptr(),
                                  current_stream(ms.device)), "alan_chain_filter")
    return alpha


def chain_logmmexp_backward(ms, tree, out_vec=None, grad_vec=None, grad_chain=None):
    """Gradient wrt ms [T,K,K] (or [B,T,K,K]) of logsumexp(chain_logmmexp(ms), -1) given grad_vec, and / or of
    chain_logmmexp(ms) given grad_chain -- autograd through utils.py:478-510, walked down the forward's tree."""
    require_device(ms, "timeseries factor")
    L = lib()
    batched = ms.ndim == 4
    m4 = ms if batched else ms.unsqueeze(0)
    B, T, K, _ = m4.shape
    code = dtype_code(ms.dtype)
    grad = t.empty(B, T, K, K, dtype=ms.dtype, device=ms.device)
    if grad_vec is not None:
        out_vec = out_vec.reshape(B, K).contiguous()
        grad_vec = grad_vec.to(ms.dtype).reshape(B, K).contiguous()
    if grad_chain is not None:
        grad_chain = grad_chain.to(ms.dtype).reshape(B, K, K).contiguous()
    nbytes = L.alan_chain_backward_batched_workspace_bytes(B, T, K, code)
    ws = t.empty(max(nbytes, 1), dtype=t.uint8, device=ms.device)
    ptr = lambda x: None if x is None else x.data_ptr()
    args = (m4.data_ptr(), code, B, T, K, *m4.stride(), tree.data_ptr(), ptr(out_vec) if grad_vec is not None else None,
            ptr(grad_vec), ptr(grad_chain), grad.data_ptr(), ws.data_ptr(), nbytes)
    rc = L.alan_chain_logmmexp_backward_batched(*args, current_stream(ms.device))
    check(rc, "alan_chain_logmmexp_backward_batched")
    if _REC[0] is not None:
        _REC[0].keep.append((m4, tree, out_vec, grad_vec, grad_chain, grad, ws))
        _REC[0].record(L.alan_chain_logmmexp_backward_batched, *args, None)
    return grad if batched else grad[0]
