"""Does a sharded Split evaluation -- all_reduce(SUM) over RCCL included -- capture into a HIP graph and replay?
One rank on one GPU (a 1-rank "nccl" group; Split.sharded is forced on), which exercises torch's capture path for the
collective; the multi-rank run is the driver's.  Usage: python tools/rccl_capture_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch as t
import torch.distributed as dist
import alan_amd as alan
from alan_amd import split as S
import bench

t.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=t.device("cuda", 0))
S.Split.sharded = lambda self: self.shard                      # a 1-rank group still goes through all_reduce
prob = bench.build_problem("cuda")
sample = bench.draw(prob, 30)
strat = alan.Split("plate_1", 38, shard=True)
eager = float(sample.elbo_nograd(strat))
ref = float(sample.elbo_nograd(alan.no_checkpoint))
print("eager sharded", eager, "unsplit", ref)
try:
    g1 = float(sample.elbo_nograd(strat, graph=True))
    g2 = float(sample.elbo_nograd(strat, graph=True))
    print("graph capture + replay with the collective inside:", g1, g2)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        v = sample.elbo_nograd(strat, graph=True)
    t.cuda.synchronize()
    print(f"replay: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us/eval")
except Exception as e:
    print("capture failed:", type(e).__name__, e)
t0 = time.perf_counter()
for _ in range(20):
    v = sample.elbo_nograd(strat)
t.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us/eval")
dist.destroy_process_group()
