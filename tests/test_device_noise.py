"""Noise generated inside the launches that use it (alan_noise_t, dist.DEVICE_NOISE): the draws x = loc + eps * scale of
Problem.sample (TorchDimDist.py:88-125) and the reparameterised gradient's sum G * eps * scale."""
import math

import numpy as np
import pytest
import torch as t

import alan_amd as alan
import models

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) on arrays of counters [n, 4] and one key (k0, k1)."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return np.stack(c, 1).astype(np.uint32)


def reference_noise(seed, first, n):
    """Elements first .. first + n - 1 of the stream alan_noise_t defines (include/alan_mi355.h)."""
    i = np.arange(first, first + n, dtype=np.uint64)
    c = i >> np.uint64(2)
    ctr = np.stack([c & np.uint64(0xFFFFFFFF), c >> np.uint64(32), np.full_like(c, 0x414C414E), np.zeros_like(c)], 1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.float64)
    lane = (i & np.uint64(3)).astype(np.int64)
    a = np.where(lane & 2, r[:, 2], r[:, 0])
    b = np.where(lane & 2, r[:, 3], r[:, 1])
    u1 = (a.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float64)
    u2 = (b.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.where(lane & 1, np.sin(2 * np.pi * u2), np.cos(2 * np.pi * u2))


def test_philox_restatement_against_the_published_known_answers():
    """Random123's kat_vectors for philox4x32 with 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want


def _draw(shape, seed, offset, perm=None):
    """eps through the library alone: x = 0 + eps * 1."""
    from alan_amd import engine as E
    from alan_amd import native as N
    flat = t.empty(math.prod(shape), device="cuda")
    e = flat.view(shape)
    if perm is not None:
        e = e.permute(perm)
    axes = tuple(range(e.ndim))
    zero, one = t.zeros((), device="cuda"), t.ones((), device="cuda")
    out = E._produce(N.MODE_AFFINE, [(zero.expand_as(e), axes), (e, axes), (one.expand_as(e), axes)], axes,
                     scales=[1.0, 1.0, 1.0], noise=(seed, offset, None, None, None, 0))
    N.flush()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seed,offset", [(0, 0), (1234, 4), (2 ** 63 + 12345, 2 ** 33 + 8), (7, 3)])
def test_generated_noise_is_the_documented_stream(seed, offset):
    n = 10007
    got = _draw((n,), seed, offset).cpu().double().numpy()
    want = reference_noise(seed, offset, n)
    assert np.abs(got - want).max() < 3e-5          # (v_log / v_sqrt / v_sin / v_cos: ~1e-6 each, radius up to 6.8)


@pytest.mark.gpu
def test_generated_noise_follows_the_placeholders_strides():
    """A permuted placeholder: element [i, j] of the view is element j * 5 + i of the stream."""
    got = _draw((7, 5), 99, 16, perm=(1, 0)).cpu().double().numpy()             # view shape [5, 7]
    want = reference_noise(99, 16, 35).reshape(7, 5).T
    assert np.abs(got - want).max() < 3e-5


@pytest.mark.gpu
def test_generated_noise_is_standard_normal():
    from scipy import stats
    x = _draw((1 << 20,), 2024, 0).cpu().double().numpy()
    assert abs(x.mean()) < 4e-3 and abs(x.var() - 1) < 6e-3
    assert abs(stats.skew(x)) < 1e-2 and abs(stats.kurtosis(x)) < 2e-2
    assert stats.kstest(x, "norm").pvalue > 1e-3
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 4e-3 and abs(np.corrcoef(x[:-2], x[2:])[0, 1]) < 4e-3


def _movielens():
    g = t.Generator().manual_seed(5)
    x = t.randn(60, 5, 18, generator=g).refine_names("plate_1", "plate_2", None)
    obs = (t.rand(60, 5, generator=g) < 0.5).float().refine_names("plate_1", "plate_2")
    prob = models.movielens(sizes={"plate_1": 60, "plate_2": 5}, x=x, obs=obs)
    prob.to("cuda")
    return prob


@pytest.mark.gpu
def test_draws_follow_torchs_generator_and_do_not_touch_memory(monkeypatch):
    """Same seed, same draws; the generator's offset moves as the draws consume it; no torch noise kernel runs."""
    from alan_amd import dist as D
    assert D.DEVICE_NOISE
    prob = _movielens()
    called = []
    real = t.Tensor.normal_
    monkeypatch.setattr(t.Tensor, "normal_", lambda self, *a, **k: (called.append(1), real(self, *a, **k))[1])
    gen = t.cuda.default_generators[0]
    t.manual_seed(11)
    o0 = gen.get_offset()
    a = prob.sample(7, reparam=False)
    o1 = gen.get_offset()
    b = prob.sample(7, reparam=False)
    t.manual_seed(11)
    c = prob.sample(7, reparam=False)
    assert not called and o1 > o0 and (o1 - o0) % 4 == 0
    ea, eb, ec = (float(s.elbo_nograd(graph=False)) for s in (a, b, c))
    assert ea == ec and ea != eb


@pytest.mark.gpu
def test_reparameterised_gradient_regenerates_the_forwards_noise():
    """d/d loc and d/d raw of sum(w * x), x = loc + eps * exp(raw) drawn by a batch: eps is never in memory -- the backward
    makes it again; checked against eps recovered from the sample itself."""
    from alan_amd import dist as D
    t.manual_seed(3)
    loc = t.randn(6, device="cuda", requires_grad=True)
    raw = (0.3 * t.randn(6, device="cuda")).requires_grad_()
    w = t.randn(50, 9, 6, device="cuda")
    batch = D._DrawBatch()
    shape = t.Size((50, 9, 6))
    la, sa = loc.expand(shape), raw.expand(shape)
    pt = batch.add(la, sa, True, shape, None, (), None, True, holder=None)
    batch.flush()
    x = pt._val
    (x * w).sum().backward()
    eps = ((x.detach() - loc.detach()) / raw.detach().exp())
    want_loc = w.sum((0, 1))
    want_raw = (w * eps * raw.detach().exp()).sum((0, 1))
    assert t.allclose(loc.grad, want_loc, rtol=1e-5, atol=1e-5)
    assert t.allclose(raw.grad, want_raw, rtol=2e-4, atol=2e-4)
    assert abs(float(eps.mean())) < 0.1 and abs(float(eps.std()) - 1) < 0.1


@pytest.mark.gpu
def test_replayed_graph_draws_what_the_iterations_launched_one_by_one_would():
    """GraphedEval: the counter lives on the device (no generator fills in front of a replay); replays after a re-seed
    repeat, and equal the same evaluations launched one by one under that seed."""
    prob = _movielens()
    ev = alan.GraphedEval(prob, 8)
    assert ev.noise.per_replay > 0
    # (one launch draws: it hands its state on through a second slot, copied back by the producers' launch -- no launch of
    # its own for that, alan_noise_t.on = 2)
    assert ev.noise.n == 1 and ev.noise.handon
    from alan_amd.training import node_kinds
    assert node_kinds(ev.graph) == (4, 0)                # draws, producers (+ hand-on), plate step, final log-sum-exp
    t.manual_seed(21)
    first = [float(ev()) for _ in range(4)]
    t.manual_seed(21)
    again = [float(ev()) for _ in range(4)]
    t.manual_seed(21)
    with t.no_grad():
        eager = [float(prob.sample(8, reparam=False).elbo_nograd(graph=False)) for _ in range(4)]
    assert first == again and len(set(first)) == 4
    for a, b in zip(first, eager):
        assert abs(a - b) <= 2e-6 * abs(b)
    # somebody else used the generator between two replays: the graph moves on from where the generator is now
    t.manual_seed(21)
    one = float(ev())
    t.randn(10, device="cuda")
    two = float(ev())
    assert one == first[0] and two != first[1]


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["vi", "rws"])
def test_graphed_step_with_generated_noise_trains(method):
    prob = _movielens()
    opt = t.optim.Adam(prob.parameters(), lr=0.01, capturable=True)
    step = alan.GraphedStep(prob, 10, opt, method=method)
    assert step.noise.per_replay > 0
    t.manual_seed(1)
    a = [float(step()) for _ in range(40)]
    assert all(math.isfinite(v) for v in a) and len(set(a)) == 40
    if method == "vi":
        assert sum(a[-10:]) / 10 > sum(a[:10]) / 10


@pytest.mark.gpu
def test_a_kernel_timer_around_draws_with_generated_noise():
    """bench.py times the plate step's backward with profiling.KernelTimer around whole VI iterations: the draws go out one
    by one then (nothing is queued under a timer), without timing events of their own."""
    from alan_amd.profiling import KernelTimer
    prob = _movielens()
    with KernelTimer() as kt:
        for _ in range(2):
            for q in prob.parameters():
                q.grad = None
            prob.sample(6, reparam=True).elbo_vi(alan.no_checkpoint).backward()
        t.cuda.synchronize()
    assert all(q.grad is None or bool(t.isfinite(q.grad).all()) for q in prob.parameters())
    assert kt.results() is not None


@pytest.mark.gpu
def test_unrolled_graphs_run_their_iterations_in_order_with_fresh_particles():
    """GraphedEval(unroll=4): one replay = four evaluations, each with its own draws -- the same twelve values as twelve
    replays of the plain graph under the same seed.  GraphedStep(unroll=3): two replays = six iterations of the one-by-one
    graph (same parameters afterwards)."""
    prob = _movielens()
    ev1, ev4 = alan.GraphedEval(prob, 8), alan.GraphedEval(prob, 8, unroll=4)
    t.manual_seed(5)
    one = [float(ev1()) for _ in range(12)]
    t.manual_seed(5)
    four = [v for _ in range(3) for v in ev4().tolist()]
    assert len(set(one)) == 12
    for a, b in zip(one, four):
        assert abs(a - b) <= 2e-6 * abs(a)
    results = []
    for unroll in (1, 3):
        t.manual_seed(0)
        p = _movielens()
        opt = t.optim.Adam(p.parameters(), lr=0.01, capturable=True)
        step = alan.GraphedStep(p, 6, opt, method="vi", unroll=unroll)
        t.manual_seed(9)
        vals = [float(step()) for _ in range(6)] if unroll == 1 else [v for _ in range(2) for v in step().tolist()]
        results.append((vals, [q.detach().clone() for q in p.parameters()]))
    (v1, p1), (v3, p3) = results
    for a, b in zip(v1, v3):
        assert abs(a - b) <= 1e-4 * abs(a)
    for a, b in zip(p1, p3):
        assert t.allclose(a, b, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_a_draw_the_library_declines_inside_a_capture_leaves_the_noise_ring_whole(monkeypatch):
    """a batch of draws ALL of whose jobs the library declines (here: every draw of more than 10,000 elements,
    i.e. the plate's z) falls back to torch's normal_ -- the ring slot reserved for it must be given back, or the ring the
    capture closes wires the neighbouring launches to a slot nobody writes and every replay repeats the same particles.
    GraphedEval(unroll=2): four draw batches per replay, two of them declined; twelve evaluations, twelve different values,
    and the values' spread is that of fresh draws."""
    from alan_amd import engine as E
    real = E._produce

    def picky(mode, factors, *a, noise=None, **k):
        if noise is not None and factors[1][0].numel() > 10000:
            return None
        return real(mode, factors, *a, noise=noise, **k)

    monkeypatch.setattr(E, "_produce", picky)
    prob = _movielens()
    ev = alan.GraphedEval(prob, 8, unroll=2)
    assert ev.noise.n == 2                      # (the two top-level batches; the two declined ones gave their slots back)
    t.manual_seed(5)
    vals = [v for _ in range(6) for v in ev().tolist()]
    assert len(set(vals)) == 12, vals
    monkeypatch.setattr(E, "_produce", real)
    ref = alan.GraphedEval(prob, 8)
    t.manual_seed(5)
    plain = t.tensor([float(ref()) for _ in range(24)])
    got = t.tensor(vals)
    assert abs(float(got.mean() - plain.mean())) <= 4 * float(plain.std()) / (12 ** 0.5) + 1e-3 * abs(float(plain.mean()))


@pytest.mark.gpu
def test_graphed_eval_replays_through_its_recorded_launches():
    """sample() + elbo on movielens is library launches from the noise to the final log-sum-exp: GraphedEval issues them
    again one by one (sample.DIRECT_REPLAY) -- the same values as its graph would produce under the same seed."""
    from alan_amd import sample as S
    prob = _movielens()
    ev = alan.GraphedEval(prob, 8)
    assert ev.calls is not None
    t.manual_seed(2)
    direct = [float(ev()) for _ in range(5)]
    S.DIRECT_REPLAY = False
    try:
        ev2 = alan.GraphedEval(prob, 8)
    finally:
        S.DIRECT_REPLAY = True
    assert ev2.calls is None
    t.manual_seed(2)
    through_graph = [float(ev2()) for _ in range(5)]
    assert direct == through_graph and len(set(direct)) == 5


@pytest.mark.gpu
def test_sampling_pipeline_draws_fresh_particles_per_evaluation_reproducibly():
    """SamplingPipeline: n overlapped `sample() + elbo` evaluations, each with its own particles (a generator state per lane
    on the device, keyed by torch's seed and the lane's number): all values different, the same values again under the same
    seed, other values under another, and distributed as the one-by-one evaluations' (GraphedEval) are."""
    prob = _movielens()
    pipe = alan.SamplingPipeline(prob, 8, alan.no_checkpoint, lanes=3, results=256)
    t.manual_seed(5)
    a = pipe.run(300).cpu()
    t.manual_seed(5)
    b = pipe.run(300).cpu()
    t.manual_seed(6)
    c = pipe.run(300).cpu()
    assert len(set(a.tolist())) == 300
    assert t.equal(a, b) and not t.equal(a, c)
    d = pipe.run(300).cpu()                       # (the generator moved on: new draws without a re-seed)
    assert not t.equal(c, d) and len(set(c.tolist()) & set(d.tolist())) == 0
    ev = alan.GraphedEval(prob, 8)
    t.manual_seed(7)
    one = t.tensor([float(ev()) for _ in range(300)])
    se = float(one.std()) / 300 ** 0.5
    assert abs(float(a.mean() - one.mean())) <= 5 * 2 ** 0.5 * se, (float(a.mean()), float(one.mean()), se)
    assert 0.7 <= float(a.std() / one.std()) <= 1.4
    pipe.close()
