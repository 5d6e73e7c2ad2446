"""world_size-2 tests of Split(..., shard=True): each rank evaluates only its chunks of the plate and
the partial log-marginals are combined by one all_reduce(SUM) (gloo here; RCCL on MI355X).  CPU only;
the contraction seam is the test-only oracle backend."""
import os
import socket
import tempfile

import pytest
import torch as t
import torch.multiprocessing as mp

import dist_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("merge", [True, False], ids=["merged", "chunked"])
@pytest.mark.parametrize("fixture,model,plate,size", [
    ("e2e_linear_gaussian_latents.pt", "linear_gaussian_latents", "T", 3),     # chunks [3,3,2,2] -> 2+2
    ("e2e_movielens_K3.pt", "movielens", "plate_1", 150),                      # one chunk per rank
    ("e2e_model1.pt", "model1", "p1", 2),                                      # nested plates, Opt params
])
def test_sharded_split_world2(fixture, model, plate, size, merge):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        mp.spawn(dist_worker.run, args=(world, _free_port(), fixture, model, plate, size, out, "cpu", merge),
                 nprocs=world, join=True)
        res = [t.load(f"{out}.{r}") for r in range(world)]
    # every rank ends with the same, full-plate ELBO == the reference's Split value
    for r in res:
        assert abs(r["elbo"] - r["ref"]) <= 1e-4 * abs(r["ref"]) + 1e-5, r
        assert abs(r["elbo"] - r["elbo_seq"]) <= 1e-5 * abs(r["elbo_seq"]) + 1e-5
        assert r["grad_err"] < 1e-3, r
    assert res[0]["elbo"] == res[1]["elbo"]
    # the chunks were really partitioned
    all_chunks = sorted(c for r in res for c in r["chunks"])
    assert all_chunks == list(range(len(all_chunks))) and all(len(r["chunks"]) >= 1 for r in res)


@pytest.mark.gpu
def test_sharded_split_world2_on_gpu():
    """Two ranks sharing cuda:0 (gloo carries the [K,K] partials): the sharded plate runs the real HIP
    kernels on each rank's chunk; every rank ends with the reference's full-plate ELBO and gradients."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        mp.spawn(dist_worker.run, args=(world, _free_port(), "e2e_movielens_K10.pt", "movielens", "plate_1", 150,
                                        out, "cuda"), nprocs=world, join=True)
        res = [t.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert abs(r["elbo"] - r["ref"]) <= 1e-4 * abs(r["ref"]) + 1e-5, r
        assert r["grad_err"] < 5e-2, r
    assert res[0]["elbo"] == res[1]["elbo"]


@pytest.mark.gpu
@pytest.mark.parametrize("merge", [True, False], ids=["merged", "chunked"])
def test_sharded_split_two_rccl_ranks(merge):
    """Two ranks, one GPU each, backend "nccl" (= RCCL over xGMI): the production layout of the sharded Split.  Every
    rank ends with the reference's full-plate ELBO -- eagerly and as a replayed HIP graph that contains the
    all-reduce -- and rank-averaged gradients equal the unsharded ones.  Needs two GPUs; skipped on a one-GPU box."""
    if t.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU node); the arithmetic is covered with gloo above")
    world = 2
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "res")
        mp.spawn(dist_worker.run, args=(world, _free_port(), "e2e_movielens_K10.pt", "movielens", "plate_1", 38,
                                        out, "cuda", merge, "nccl"), nprocs=world, join=True)
        res = [t.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert abs(r["elbo"] - r["ref"]) <= 1e-4 * abs(r["ref"]) + 1e-5, r
        assert abs(r["elbo"] - r["elbo_seq"]) <= 1e-5 * abs(r["elbo_seq"]) + 1e-5
        assert all(abs(g - r["elbo"]) <= 1e-6 * abs(r["elbo"]) for g in r["graphed"]), r
        assert r["grad_err"] < 5e-2, r
    assert res[0]["elbo"] == res[1]["elbo"]
    all_chunks = sorted(c for r in res for c in r["chunks"])
    assert all_chunks == list(range(len(all_chunks)))


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun starts two child ranks itself (before touching the GPU) and
    prints rank 0's JSON line with the C4 figure.  Needs two GPUs."""
    if t.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--c4-only"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "c4_movielens_K100" in line


@pytest.mark.gpu
def test_sharded_split_with_its_rccl_all_reduce_captures_into_a_graph(monkeypatch):
    """One rank on one GPU (a 1-rank "nccl" group, Split.sharded forced on): the sharded evaluation -- queued producer
    launches flushed before the collective, the RCCL all-reduce itself -- gives the unsharded value eagerly and as a
    captured, replayed HIP graph.  (The multi-rank arithmetic is covered on the CPU with gloo above.)"""
    import os
    import torch as t
    import torch.distributed as dist
    import alan_amd as alan
    from alan_amd import split as S
    import models
    from conftest import load_golden
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=t.device("cuda", 0))
    except Exception as e:                                   # no RCCL on this box: nothing to pin here
        pytest.skip(f"cannot create a 1-rank nccl group: {e}")
    try:
        monkeypatch.setattr(S.Split, "sharded", lambda self: self.shard)
        fx = load_golden("e2e_movielens_K10.pt")
        prob = models.BUILDERS["movielens"](fx).to("cuda")
        sample = models.sample_from_fixture(prob, fx, "cuda")
        ref = float(sample.elbo_nograd(alan.no_checkpoint))
        strat = alan.Split("plate_1", 38, shard=True)
        eager = float(sample.elbo_nograd(strat))
        g1 = float(sample.elbo_nograd(strat, graph=True))
        g2 = float(sample.elbo_nograd(strat, graph=True))
        assert abs(eager - ref) <= 1e-6 * abs(ref)
        assert g1 == g2 and abs(g1 - eager) <= 1e-6 * abs(eager)
    finally:
        dist.destroy_process_group()
