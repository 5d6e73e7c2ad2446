"""Worker of the one-shot exchange tests: `world` processes SHARING cuda:0 (gloo carries the IPC handles and the
checks), each one rank of alan_exchange_sum.  On one GPU this tests the arithmetic and the protocol (slots, flags, the
device-side running number under graph replay) -- not the xGMI fabric."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _payload(rank, step, n):
    import torch as t
    g = t.Generator().manual_seed(1000 * step + rank)
    return t.randn(n, generator=g) * 10.0 ** (rank - 1)          # (different magnitudes: the order of the sum shows)


def run(rank, world, port, out_path, what="direct"):
    import torch as t
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["ALAN_AMD_ONE_SHOT"] = "1"
    os.environ.setdefault("ALAN_EXCHANGE_SPIN_MS", "20000")       # (ranks time-slicing ONE GPU: generous waits)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        import alan_amd as alan
        from alan_amd import split as S
        t.cuda.set_device(0)
        if what == "direct":
            ex = S.exchange_for(None)
            worst, bitwise = 0.0, True
            sizes = [1, 37, 10000, 4096, 65536, 3, 10000, 10000]
            for step, n in enumerate(sizes):
                out = ex.sum(_payload(rank, step, n).cuda())
                want = _payload(0, step, n)
                for q in range(1, world):
                    want = want + _payload(q, step, n)                     # (rank order, fp32: what the kernel does)
                bitwise &= bool((out.cpu() == want).all())
                worst = max(worst, float((out.cpu() - want).abs().max()))
            # the same launch captured once and replayed: the running number advances on the device
            src = t.zeros(10000, device="cuda")
            ex.sum(src)                                                    # (warm-up on the capture's side stream)
            t.cuda.synchronize()
            dist.barrier()
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g):
                out = ex.sum(src)
            replays = []
            for step in range(100, 105):
                src.copy_(_payload(rank, step, 10000))
                g.replay()
                t.cuda.synchronize()
                want = _payload(0, step, 10000)
                for q in range(1, world):
                    want = want + _payload(q, step, 10000)
                replays.append(bool((out.cpu() == want).all()))
            dist.barrier()
            done, bad = ex.status()
            res.update(bitwise=bitwise, worst=worst, replays=replays, done=done, bad=bad)
        elif what == "time":
            ex = S.exchange_for(None)
            src = t.randn(10000, device="cuda")
            for _ in range(20):
                ex.sum(src)
            t.cuda.synchronize()
            dist.barrier()
            import time
            t0 = time.perf_counter()
            for _ in range(500):
                ex.sum(src)
            t.cuda.synchronize()
            res.update(us_per_exchange=(time.perf_counter() - t0) / 500 * 1e6, bad=ex.status()[1])
        elif what == "split":
            import models
            g = t.Generator().manual_seed(5)
            x = t.randn(300, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
            obs = (t.rand(300, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
            prob = models.movielens(sizes={"plate_1": 300, "plate_2": 5}, x=x, obs=obs)
            prob.to("cuda")
            t.manual_seed(3)                                               # (every rank draws the same particles)
            sample = prob.sample(10, reparam=False)
            strat = alan.Split("plate_1", 150, shard=True)
            one_shot = float(sample.elbo_nograd(strat, graph=False))
            graphed = [float(sample.elbo_nograd(strat, graph=True)) for _ in range(3)]
            res["direct"] = next(iter(sample.__dict__["_graphs"].values())).calls is not None
            S.ONE_SHOT_EXCHANGE = False
            through_gloo = float(sample.elbo_nograd(strat, graph=False))
            alone = float(sample.elbo_nograd(alan.Split("plate_1", 150), graph=False))
            S.ONE_SHOT_EXCHANGE = True
            lp = sample.elbo_rws(strat)                                    # (gradients flow through the exchange)
            params = [p for p in prob.parameters() if p.requires_grad]
            grads = t.autograd.grad(lp, params, allow_unused=True)
            gsum = []
            for gr, p in zip(grads, params):
                gr = (t.zeros_like(p) if gr is None else gr.clone()).cpu()
                dist.all_reduce(gr)
                gsum.append(gr / world)
            ref = t.autograd.grad(sample.elbo_rws(alan.Split("plate_1", 150)), params, allow_unused=True)
            gerr = max(float((a - (t.zeros_like(a) if b is None else b.cpu())).abs().max() /
                             (1e-6 + (t.zeros_like(a) if b is None else b.cpu()).abs().max())) for a, b in zip(gsum, ref))
            dist.barrier()
            done, bad = S.exchange_for(None).status()
            res.update(one_shot=one_shot, graphed=graphed, through_gloo=through_gloo, alone=alone, grad_err=gerr,
                       done=done, bad=bad)
        elif what == "pipeline":
            import models
            g = t.Generator().manual_seed(5)
            x = t.randn(120, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
            obs = (t.rand(120, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
            prob = models.movielens(sizes={"plate_1": 120, "plate_2": 5}, x=x, obs=obs)
            prob.to("cuda")
            t.manual_seed(3)
            sample = prob.sample(10, reparam=False)
            strat = alan.Split("plate_1", 60, shard=True)
            eager = float(sample.elbo_nograd(strat, graph=False))
            alone = float(sample.elbo_nograd(alan.Split("plate_1", 60), graph=False))
            pipe = sample.pipeline(strat, lanes=2, results=64)
            vals = pipe.run(12).cpu()
            vals2 = pipe.run(5).cpu()
            t.cuda.synchronize()
            dist.barrier()
            bad = [ex.status()[1] for ex in S._EXCHANGES.values()]
            res.update(eager=eager, alone=alone, vals=vals.tolist() + vals2.tolist(), bad=max(bad), n_exchanges=len(S._EXCHANGES))
            pipe.close()
        dist.barrier()
        S.close_exchanges()
        t.save(res, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()
