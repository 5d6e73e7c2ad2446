"""What one alan_exchange_sum costs with `world` processes sharing this GPU (40 KB partials, 500 back-to-back exchanges):
launch + flag round trip through HBM -- the fabric is not in it.   python3 tools/exchange_probe.py [world]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    import torch as t
    import torch.multiprocessing as mp
    import exchange_worker
    from test_exchange import _free_port
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        mp.spawn(exchange_worker.run, args=(world, _free_port(), out, "time"), nprocs=world, join=True)
        for r in range(world):
            print(t.load(f"{out}.{r}"))
