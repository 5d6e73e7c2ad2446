import os, torch as t, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=t.device("cuda",0))
x = t.ones(900, device="cuda")
dist.all_reduce(x); t.cuda.synchronize()
g = t.cuda.CUDAGraph()
s = t.cuda.Stream(); s.wait_stream(t.cuda.current_stream())
with t.cuda.stream(s):
    dist.all_reduce(x)
t.cuda.current_stream().wait_stream(s); t.cuda.synchronize()
with t.cuda.graph(g):
    y = x * 2
    dist.all_reduce(y)
g.replay(); t.cuda.synchronize()
print("nccl-in-graph ok", float(y.sum()))
dist.destroy_process_group()
