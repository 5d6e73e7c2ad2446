"""Worker for the world_size-2 gloo tests of the sharded Split (CPU, oracle backend = test-only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rank, world, port, fixture, model, platename, split_size, out_path, device="cpu", merge=None, backend_name="gloo"):
    import torch as t
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend_name == "nccl":                 # one GPU per rank, RCCL carries the partials (the production layout)
        device = f"cuda:{rank}"
        t.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=t.device(device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import alan_amd as alan
        import models
        from conftest import load_golden
        from oracle import backend
        t.set_num_threads(2)
        fx = load_golden(fixture)
        prob = models.BUILDERS[model](fx).to(device)
        sample = models.sample_from_fixture(prob, fx, device)
        strat = alan.Split(platename, split_size, shard=True, merge=merge)
        import contextlib
        # CPU ranks: test-only oracle backend; GPU ranks: the real HIP library (gloo moves the partials)
        with (backend.installed() if device == "cpu" else contextlib.nullcontext()):
            assert strat.sharded()
            n_chunks = len(alan.split.chunk_sizes(prob.all_platedims[platename].size, split_size))
            mine = list(strat.my_chunks(n_chunks))
            val = sample.elbo_nograd(strat)
            graphed = None
            if backend_name == "nccl":         # the sharded evaluation, collective included, as a replayed HIP graph
                graphed = [float(sample.elbo_nograd(strat, graph=True)) for _ in range(3)]
            # gradients flow through the all-reduce; AVERAGING them over ranks (the DDP convention)
            # gives exactly the gradient of the unsharded ELBO
            lp = sample.elbo_rws(strat)
            params = [p for p in prob.parameters() if p.requires_grad]
            grads = t.autograd.grad(lp, params, allow_unused=True) if params else []
            gsum = []
            for g, p in zip(grads, params):
                g = t.zeros_like(p) if g is None else g.clone()
                dist.all_reduce(g)
                gsum.append(g / world)
            ref_val = sample.elbo_rws(alan.Split(platename, split_size))      # sequential Split, this rank alone
            ref_grads = t.autograd.grad(ref_val, params, allow_unused=True) if params else []
        t.save({"rank": rank, "elbo": float(val), "chunks": mine, "ref": float(fx["elbo"]["split"]), "graphed": graphed,
                "elbo_seq": float(ref_val),
                "grad_err": max([float((a - (t.zeros_like(a) if b is None else b)).abs().max())
                                 for a, b in zip(gsum, ref_grads)] or [0.0])},
               f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()
