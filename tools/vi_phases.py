#!/usr/bin/env python3
"""Kernel launches of one eager training iteration, by phase (sample / elbo forward / backward / optimiser) and by op:
    python3 tools/vi_phases.py [vi|rws]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
from torch.profiler import profile, ProfilerActivity
mode = sys.argv[1] if len(sys.argv) > 1 else "vi"
prob = bench.build_problem("cuda")
params = list(prob.parameters()) if mode == "vi" else list(prob.Q.parameters())
opt = t.optim.Adam(params, lr=1e-2, capturable=True, fused=True, maximize=(mode == "rws"))
state = {}
def p_sample():
    opt.zero_grad(set_to_none=True)
    state["s"] = prob.sample(30, reparam=(mode == "vi"))
def p_fwd():
    s = state["s"]
    state["e"] = s.elbo_vi(alan.no_checkpoint) if mode == "vi" else s.elbo_rws(alan.no_checkpoint)
def p_bwd():
    (-state["e"]).backward()
def p_opt():
    opt.step()
phases = [("sample", p_sample), ("forward", p_fwd), ("backward", p_bwd), ("optimiser", p_opt)]
for _ in range(3):
    for _, f in phases: f()
t.cuda.synchronize()
for name, f in phases:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        f(); t.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type.name == "CUDA"]
    ops = collections.Counter()
    for e in prof.key_averages():
        if e.key.startswith("aten::") and e.self_device_time_total > 0:
            ops[e.key] += e.count
    print(f"== {name}: {len(ks)} kernels, {sum(e.device_time_total for e in ks):.0f} us of kernel time")
    print("   aten ops with kernels:", dict(ops.most_common(14)))
    names = collections.Counter(e.name.split("(")[0][-60:] for e in ks if "alan::" in e.name or "Cijk" in e.name)
    print("   alan / BLAS kernels:", dict(names))
