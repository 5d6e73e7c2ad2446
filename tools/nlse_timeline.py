#!/usr/bin/env python3
"""Per-phase timeline of the fused plate step's waves, from the diagnostic build:
    make -C alan_amd/csrc TIMELINE=1 && gpurun -- python3 tools/nlse_timeline.py [M K E]
Every wave of ONE launch stamps s_memtime at: 0 entry, 1 B table + log-normalisers written (before the barrier),
2 barrier passed, 3 operands requested (loc rows, first value tile) and loop entered, 4 first value tile landed,
5 first tile's units done, 6 last tile's units done, 7 partial sums stored (issued), 8 stores drained; plus
s_memrealtime (100 MHz) at entry and exit.  Printed: the distribution of each phase over the waves, when waves start
and end relative to the first, and the launch's duration by HIP events -- what the kernel's microseconds are made of."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALAN_AMD_LIB"] = os.path.join(ROOT, "tools", "_build", "timeline", "libalan_mi355.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch as t
from alan_amd import engine as E, native as N
from alan_amd.dims import Dim

M, K, Ev = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (300, 30, 18)
TABLE = "table" in sys.argv[4:]           # the scale table built ahead of the launch (a launch of its own here; in an evaluation it
                                          # rides in the producers' launch): the kernel's TBL variant
if TABLE:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _scale_table import force_scale_table
    force_scale_table()
g = t.Generator(device="cuda").manual_seed(0)
pl, Kz, dl, ds = Dim("plate", M), Dim("K", K), Dim("Kl", K), Dim("Ks", K)
z = t.randn(M, K, Ev, device="cuda", generator=g)
mu = t.randn(K, Ev, device="cuda", generator=g)
raw = 0.3 * t.randn(K, Ev, device="cuda", generator=g)
sm = [(t.randn(M, K, device="cuda", generator=g), (pl, Kz)) for _ in range(2)]
call = lambda: E.normal_lse((z, (pl, Kz)), (mu, (dl,)), (raw, (ds,)), sm, pl, Kz, log_scale=True)
for _ in range(20):
    call()
t.cuda.synchronize()
a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
a.record(); call(); b.record(); t.cuda.synchronize()
L = N.lib()
SLOTS, NW = 16, 8192
buf = (C.c_ulonglong * (SLOTS * NW))()
L.alan_nlse_timeline_read.restype = C.c_int
assert L.alan_nlse_timeline_read(buf, NW) == SLOTS
tl = np.frombuffer(buf, dtype=np.uint64).reshape(NW, SLOTS).astype(np.int64)
tl = tl[tl[:, 8] > 0]                                  # waves that ran to the end (idle waves of a workgroup leave early)
cyc = tl[:, :9] - tl[:, :1]
real0 = (tl[:, 9] - tl[:, 9].min()) * 10.0             # ns after the first wave's entry
real1 = (tl[:, 12] - tl[:, 9].min()) * 10.0
clock = np.median((tl[:, 8] - tl[:, 0]) / np.maximum(1, (tl[:, 12] - tl[:, 9]) * 10.0))     # cycles per ns
print(f"M={M} K={K} E={Ev}{' (scale table built ahead of the launch)' if TABLE else ''}: {len(tl)} waves, {int(np.median(tl[:, 11]))} units per wave (median), in-kernel clock {clock:.2f} GHz, "
      f"events around the two launches of the call (kernel + second stage) {a.elapsed_time(b) * 1e3:.1f} us")
names = ["entry -> B table written", "-> barrier passed", "-> operands requested, loop entered", "-> first value tile landed",
         "-> first tile's units done", "-> last tile's units done", "-> partial sums stored", "-> stores drained"]
print(f"{'phase':44s} {'median':>9s} {'p10':>9s} {'p90':>9s} {'max':>9s}   (ns; cycles / clock)")
for i, nm in enumerate(names):
    dphase = (cyc[:, i + 1] - cyc[:, i]) / clock
    print(f"{nm:44s} {np.median(dphase):9.0f} {np.percentile(dphase, 10):9.0f} {np.percentile(dphase, 90):9.0f} {dphase.max():9.0f}")
pro = (tl[:, 13:16] - tl[:, :1]) / clock
print("inside the prologue (ns after entry, median / p90): kernel arguments read %.0f / %.0f, every load issued %.0f / %.0f, "
      "all of them landed %.0f / %.0f" % tuple(x for i in range(3) for x in (np.median(pro[:, i]), np.percentile(pro[:, i], 90))))
life = cyc[:, 8] / clock
print(f"{'wave lifetime':44s} {np.median(life):9.0f} {np.percentile(life, 10):9.0f} {np.percentile(life, 90):9.0f} {life.max():9.0f}")
per_unit = (cyc[:, 6] - cyc[:, 4]) / clock / np.maximum(1, tl[:, 11])
print(f"{'per unit, first tile landed -> last done':44s} {np.median(per_unit):9.0f} {np.percentile(per_unit, 10):9.0f} {np.percentile(per_unit, 90):9.0f} {per_unit.max():9.0f}")
print(f"wave entry after the first wave's: median {np.median(real0):.0f} ns, p90 {np.percentile(real0, 90):.0f}, last {real0.max():.0f}")
print(f"wave exit  after the first wave's entry: median {np.median(real1):.0f} ns, p90 {np.percentile(real1, 90):.0f}, last {real1.max():.0f}  "
      f"(= the kernel's span seen from inside)")
# where the waves ran: (XCC, shader engine, CU) from HW_REG_XCC_ID / HW_REG_HW_ID, and how a wave's pace depends on company
hw = tl[:, 10]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
cu = (xcc << 12) | (((hwid >> 13) & 0x7) << 8) | (((hwid >> 12) & 0x1) << 4) | ((hwid >> 8) & 0xf)      # se_id, sh_id, cu_id
simd = (hwid >> 4) & 0x3
per_cu = {}
for c in cu:
    per_cu[int(c)] = per_cu.get(int(c), 0) + 1
counts = np.array(sorted(per_cu.values()))
print(f"CUs used: {len(per_cu)}; waves per CU: min {counts.min()}, median {int(np.median(counts))}, max {counts.max()}; "
      f"histogram " + ", ".join(f"{v} waves: {int((counts == v).sum())} CUs" for v in sorted(set(counts.tolist()))))
company = np.array([per_cu[int(c)] for c in cu])
for v in sorted(set(company.tolist())):
    sel = company == v
    print(f"  waves on a CU holding {v} of this launch's waves: {int(sel.sum())}, per unit median {np.median(per_unit[sel]):.0f} ns, lifetime median {np.median(life[sel]):.0f} ns")
late = real0 > 1500
print(f"waves entering more than 1.5 us after the first: {int(late.sum())} of {len(tl)} (a second round of workgroups)")
hist, edges = np.histogram(real0, bins=12)
print("entry histogram (ns):", ", ".join(f"{int(edges[i])}-{int(edges[i + 1])}: {hist[i]}" for i in range(len(hist))))
