"""
Per-kernel timing with raw HIP events on the launch stream (bench.py's ``roofline`` leg).

``with KernelTimer() as kt: ...`` makes every ``alan_reduce`` call inside carry a hipEvent pair that the
library records immediately before/after the call's DOMINANT kernel (alan_reduce_desc_t.ev_start/stop),
so the measured interval is that kernel alone -- not the Python, ctypes or second-stage launches.
"""
import ctypes as C

from . import native as N

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
    return _hip


class KernelTimer:
    def __init__(self, min_bytes=0):
        self.min_bytes = min_bytes
        self.records = []          # (mode, algorithmic_bytes, ev_start, ev_stop)

    def __enter__(self):
        self._prev = N._TIMER[0]
        N._TIMER[0] = self
        return self

    def __exit__(self, *a):
        N._TIMER[0] = self._prev

    def attach(self, desc, algo_bytes, mode=None, flops=0.0):
        """``mode`` / ``flops``: the fused plate step's descriptors carry no mode field; their records are tagged
        native.MODE_FUSED_FWD / _BWD and carry the algorithmic FLOPs of the tile GEMM in place of bytes."""
        if mode is None and algo_bytes < self.min_bytes:
            return
        h = hip()
        a, b = C.c_void_p(), C.c_void_p()
        assert h.hipEventCreate(C.byref(a)) == 0 and h.hipEventCreate(C.byref(b)) == 0
        desc.ev_start, desc.ev_stop = a.value, b.value
        self.records.append((int(desc.mode) if mode is None else mode, algo_bytes if mode is None else flops, a, b))

    def results(self):
        """[(mode, algorithmic_bytes, milliseconds)] -- call after the stream has been synchronised."""
        h = hip()
        out = []
        for mode, nbytes, a, b in self.records:
            ms = C.c_float()
            h.hipEventSynchronize(b)
            rc = h.hipEventElapsedTime(C.byref(ms), a, b)
            if rc == 0:
                out.append((mode, nbytes, ms.value))
            h.hipEventDestroy(a)
            h.hipEventDestroy(b)
        self.records = []
        return out
