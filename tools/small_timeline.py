#!/usr/bin/env python3
"""Where the producers' multi-problem launch of a movielens evaluation spends its life, from the
diagnostic build:   make -C alan_amd/csrc TIMELINE=1 && gpurun -- python3 tools/small_timeline.py [K]
Every workgroup of reduce_small_multi_kernel stamps s_memtime at: 0 entry, 1 its problem found and the descriptor's head
read, 2 the first round of loads landed, 3 the walk over the reduced dim done, 4 lanes combined + result stored (issued),
5 stores drained; plus s_memrealtime (100 MHz) at entry and exit, its problem and where it ran.  Printed per problem: when its
workgroups start and end relative to the launch's first, and the phases' durations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALAN_AMD_LIB"] = os.path.join(ROOT, "tools", "_build", "timeline", "libalan_mi355.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch as t
import alan_amd as alan
from alan_amd import native as N
import models

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = 300
g = t.Generator().manual_seed(5)
xx = t.randn(M, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
obs = (t.rand(M, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
prob = models.movielens(sizes={"plate_1": M, "plate_2": 5}, x=xx, obs=obs)
prob.to("cuda")
t.manual_seed(3)
sample = prob.sample(K, reparam=False)
for _ in range(20):
    sample.elbo_nograd(alan.no_checkpoint, graph=False)          # eager: the launches one by one, warm
t.cuda.synchronize()
L = N.lib()
SLOTS, NW = 12, 4096
buf = (C.c_ulonglong * (SLOTS * NW))()
L.alan_small_timeline_read.restype = C.c_int
assert L.alan_small_timeline_read(buf, NW) == SLOTS
tl = np.frombuffer(buf, dtype=np.uint64).reshape(NW, SLOTS).astype(np.int64)
wide = tl[NW - 1].copy()                    # the evaluation's last launch (reduce_wide_kernel: one workgroup of 1024 threads)
grid = int(tl[0, 10])
tl = tl[:min(grid, NW - 1)]
clock = np.median((tl[:, 5] - tl[:, 0]) / np.maximum(1, (tl[:, 7] - tl[:, 6]) * 10.0))      # cycles per ns
first = tl[:, 6].min()
entry, exit_ = (tl[:, 6] - first) * 10.0, (tl[:, 7] - first) * 10.0
names = {0: "LSE", 1: "SUM", 3: "NORMAL", 4: "BERNOULLI", 5: "NORMAL_LOGSCALE", 7: "BERNOULLI_LINEAR", 11: "NORMAL_TABLE"}
print(f"movielens K={K} M={M}: the producers' launch = {grid} workgroups of 256 threads; in-kernel clock {clock:.2f} GHz; "
      f"first entry -> last exit {exit_.max():.0f} ns (the kernel's span seen from inside)")
print(f"{'problem':34s} {'wgs':>5s} | entry after the first (ns): med   p90  last | exit: med   p90  last | "
      f"decode  loads landed  walk  combine+store  drain (ns, median)")
prob_id = tl[:, 9] & 0xffffffff
for p in sorted(set(prob_id.tolist())):
    sel = prob_id == p
    mode = int(tl[sel][0, 9] >> 32)
    st = tl[sel][:, 0:6].astype(np.float64)
    for i in range(1, 6):                       # (a phase a problem's body does not stamp: its slot stays 0 -> the previous stamp)
        st[:, i] = np.where(st[:, i] > 0, st[:, i], st[:, i - 1])
    ph = (st[:, 1:6] - st[:, 0:5]) / clock
    ph = np.where(tl[sel][:, 1:6] > 0, ph, np.nan)
    med = [np.nanmedian(ph[:, i]) if np.isfinite(ph[:, i]).any() else float("nan") for i in range(5)]
    e, x = entry[sel], exit_[sel]
    print(f"{p}: {names.get(mode, str(mode)):31s} {int(sel.sum()):5d} | {np.median(e):27.0f} {np.percentile(e, 90):5.0f} {e.max():5.0f} | "
          f"{np.median(x):9.0f} {np.percentile(x, 90):5.0f} {x.max():5.0f} | " + "  ".join(f"{v:8.0f}" for v in med))
life = (tl[:, 5] - tl[:, 0]) / clock
print(f"workgroup lifetime: median {np.median(life):.0f} ns, p90 {np.percentile(life, 90):.0f}, max {life.max():.0f}")
hw = tl[:, 8]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
cu = (xcc << 12) | (((hwid >> 13) & 0x7) << 8) | (((hwid >> 12) & 0x1) << 4) | ((hwid >> 8) & 0xf)
per_cu = np.unique(cu, return_counts=True)[1]
print(f"CUs used: {len(per_cu)}; workgroups per CU: min {per_cu.min()}, median {int(np.median(per_cu))}, max {per_cu.max()}")
hist, edges = np.histogram(entry, bins=10)
print("entry histogram (ns):", ", ".join(f"{int(edges[i])}-{int(edges[i + 1])}: {hist[i]}" for i in range(len(hist))))
if wide[5] > 0:
    w = (wide[1:6] - wide[0]) / clock
    print(f"the evaluation's last launch (reduce_wide_kernel, 1 workgroup x 1024 threads; ns after its entry): descriptor + addresses + "
          f"first loads landed {w[1]:.0f}, the slices' loads + walk done {w[2]:.0f}, 16 waves combined + result stored {w[3]:.0f}, "
          f"drained {w[4]:.0f}; entry -> exit by the 100 MHz clock {(wide[7] - wide[6]) * 10} ns")
