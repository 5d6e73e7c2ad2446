#!/usr/bin/env python3
"""Which torch ops make up a training iteration?  One eager elbo_vi iteration under the torch profiler, grouped by op and
by the Python frame that issued it:  python3 tools/vi_torch_profile.py [vi|rws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
from torch.profiler import profile, ProfilerActivity
mode = sys.argv[1] if len(sys.argv) > 1 else "vi"
prob = bench.build_problem("cuda")
params = list(prob.parameters()) if mode == "vi" else list(prob.Q.parameters())
opt = t.optim.Adam(params, lr=1e-2, capturable=True, fused=True, maximize=(mode == "rws"))
def it():
    opt.zero_grad(set_to_none=True)
    s = prob.sample(30, reparam=(mode == "vi"))
    e = s.elbo_vi(alan.no_checkpoint) if mode == "vi" else s.elbo_rws(alan.no_checkpoint)
    (-e).backward()
    opt.step()
for _ in range(3): it()
t.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    it(); t.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type.name == "CUDA" or (e.cuda_time_total if hasattr(e, "cuda_time_total") else 0)]
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=50, max_src_column_width=90))
