import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench, alan_amd as alan
from torch.profiler import profile, ProfilerActivity
K=int(sys.argv[1]) if len(sys.argv)>1 else 100
prob=bench.build_problem("cuda"); s=bench.draw(prob,K); st=bench.strategy_for(1,K)
for _ in range(2): s.elbo_nograd(st)
t.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    s.elbo_nograd(st); t.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if any(k in e.key for k in ("mm","matmul","bmm","addmm")):
        print(e.key, e.input_shapes, "cuda_us", round(e.device_time_total,1), "count", e.count)
