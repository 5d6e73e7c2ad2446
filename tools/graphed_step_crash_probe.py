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
# AMD_LOG_LEVEL=3 (round 4, VERDICT r3 item 9 ii): every HIP API call is logged; kept are the lines that name a stream,
# capture, graph or event call -- the last of them is the call the process died in
level = os.environ.get("ALAN_CRASH_PROBE_LOG_LEVEL", "3")
env = dict(os.environ, AMD_LOG_LEVEL=level)
r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=400, env=env)
lines = r.stderr.splitlines()
api = [l[:260] for l in lines if any(k in l for k in ("hipStream", "Capture", "hipGraph", "hipEvent", "hipLaunchHostFunc"))
       and not any(k in l for k in ("Current capture node LaunchKernel", "Add KernelNode", "hipStreamGetCaptureInfo", "hipStreamIsCapturing"))]
rest = [l[:300] for l in lines if "Fatal Python error" in l or "File \"" in l or "UserWarning" in l]
out = (f"child return code: {r.returncode} (negative = killed by that signal)\n---- stdout\n{r.stdout[-3000:]}\n"
       f"---- the last 60 stream / capture / graph / event API lines of AMD_LOG_LEVEL={level} (of {len(api)})\n" + "\n".join(api[-60:]) +
       "\n---- warnings and the fault handler\n" + "\n".join(rest[-30:]) + "\n")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "graphed_step_crash.txt"), "w").write(out)
print(out)
