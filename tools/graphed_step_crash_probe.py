#!/usr/bin/env python3
"""ONE deliberate run of what training.GraphedStep refuses: a training step captured while the autograd
graph of an earlier eager backward is still alive -- in a fresh child process, with the guard bypassed, so that what the
runtime says when the process dies is on record (profiles/r3_graphed_step_crash.txt).  Not a test: run once.
    gpurun -- python3 tools/graphed_step_crash_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, %r)
import torch as t, bench, alan_amd as alan
from alan_amd.training import GraphedStep, stale_grad_accumulators
prob = bench.build_problem("cuda")
loss = -prob.sample(10, reparam=True).elbo_vi(alan.no_checkpoint)
loss.backward()                                   # eager, on the default stream; `loss` stays alive
held = stale_grad_accumulators(prob.parameters())
print("parameters whose AccumulateGrad node is held by the live graph:", len(held), flush=True)
opt = t.optim.Adam(prob.parameters(), lr=1e-2, capturable=True)
try:
    GraphedStep(prob, 10, opt)
    print("guard: NOT refused", flush=True)
except RuntimeError as e:
    print("guard: refused --", str(e)[:120], flush=True)
print("now with the guard bypassed:", flush=True)
import alan_amd.training as _T
_T._UNSAFE_SKIP_STALE_CHECK = True            # (the guard bypassed: this probe exists to record the crash, once)
step = GraphedStep(prob, 10, opt)
print("capture ended without a crash; replaying", flush=True)
for _ in range(3):
    v = step()
t.cuda.synchronize()
print("replayed:", float(v), flush=True)
''' % ROOT
env = dict(os.environ, AMD_LOG_LEVEL="1")
r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=240, env=env)
out = (f"child return code: {r.returncode} (negative = killed by that signal)\n---- stdout\n{r.stdout[-3000:]}\n---- stderr (tail)\n"
       f"{r.stderr[-6000:]}\n")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "graphed_step_crash.txt"), "w").write(out)
print(out)
