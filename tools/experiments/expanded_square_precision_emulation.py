import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16(x); e1 = (x - h).astype(np.float32); m = bf16(e1); e2 = (e1 - m).astype(np.float32); l = bf16(e2)
    return h, m, l
def dot6(a, b):
    """sum over last axis of a*b using the six bf16 terms, fp32 accumulation per 16-slot MFMA step (inner sum exact)"""
    ah, am, al = split3(a); bh, bm, bl = split3(b)
    terms = [ah * bh.astype(np.float64), ah * bm.astype(np.float64), am * bh.astype(np.float64), ah * bl.astype(np.float64), al * bh.astype(np.float64), am * bm.astype(np.float64)]
    # slots: per event 6 terms; steps of 16 slots -> ~2.67 events per step; emulate: accumulate per event-triple in fp32
    per_event = sum(terms)   # float64 [..., E]
    E = per_event.shape[-1]
    acc = np.zeros(per_event.shape[:-1], np.float32)
    for e0 in range(0, E, 3):
        acc = (acc.astype(np.float64) + per_event[..., e0:e0 + 3].sum(-1)).astype(np.float32)
    return acc
def run(v, mu, sig, small, name):
    M, K, E = v.shape; L = mu.shape[0]; S = sig.shape[0]
    LOG2E = 1.4426950408889634
    w64 = LOG2E / (2 * sig.astype(np.float64) ** 2)
    # exact fp64: D[m,k,l,s] in log2 units
    d = v[:, :, None, None, :].astype(np.float64) - mu[None, None, :, None, :]
    D64 = (d * d * w64[None, None, None]).sum(-1) - small[:, :, None, None].astype(np.float64) * LOG2E
    lgn = np.log(sig.astype(np.float64)).sum(-1) + E * 0.9189385332046727
    def finish(D):
        mn = D.min(1, keepdims=True)
        lse = -(mn[:, 0] - np.log2(np.exp2(mn - D).sum(1) + 1.19e-7)) / LOG2E - lgn[None, None, :]
        return lse.sum(0)
    ref = finish(D64)
    # current kernel: A = (v-mu)^2 in fp32, B = w fp32, 6-term product
    w32 = w64.astype(np.float32)
    d32 = (v[:, :, None, :] - mu[None, None, :, :]).astype(np.float32); a = (d32 * d32).astype(np.float32)   # [M,K,L,E]
    Dcur = np.stack([dot6(a, w32[s][None, None, None, :]) for s in range(S)], -1) - (small[:, :, None, None] * np.float32(LOG2E)).astype(np.float32)
    cur = finish(Dcur.astype(np.float64))
    # expanded: centre on mu[0]
    c = mu[0]
    vp = (v - c).astype(np.float32); mp = (mu - c).astype(np.float32)
    T1 = np.stack([dot6((vp * vp).astype(np.float32), w32[s][None, None, :]) for s in range(S)], -1)     # [M,K,S]
    T1 = (T1 - (small[:, :, None] * np.float32(LOG2E)).astype(np.float32)).astype(np.float32)
    Dexp = np.empty((M, K, L, S), np.float32)
    for l in range(L):
        for s in range(S):
            b = (-2 * mp[l] * w32[s]).astype(np.float32)
            cross = dot6(vp, b[None, None, :])
            Dexp[:, :, l, s] = (T1[:, :, s].astype(np.float64) + cross).astype(np.float32)   # accumulate on top of T1
    cconst = (mp[:, None, :].astype(np.float64) ** 2 * w64[None]).sum(-1)   # [L,S] exact-ish (fp32 in kernel)
    mn = Dexp.astype(np.float64).min(1, keepdims=True)
    lse = -(mn[:, 0] + cconst[None] - np.log2(np.exp2(mn - Dexp).sum(1) + 1.19e-7)) / LOG2E - lgn[None, None, :]
    exp_ = lse.sum(0)
    sc = np.abs(ref)
    print(f"{name}: |out| median {np.median(sc):.1f}; max rel err current {np.max(np.abs(cur - ref) / sc):.2e}, expanded {np.max(np.abs(exp_ - ref) / sc):.2e}; "
          f"max abs err current {np.max(np.abs(cur - ref)):.2e} expanded {np.max(np.abs(exp_ - ref)):.2e}; T1 max {T1.max():.3g}")
M, K, E = 40, 30, 18
f = lambda *s: rng.standard_normal(s).astype(np.float32)
# 1. movielens at initialisation
v, mu, psi, small = f(M, K, E), f(K, E), f(K, E), f(M, K)
run(v, mu, np.exp(psi), small, "movielens init (mu~N(0,1), sigma=exp(N(0,1)))")
# 2. trained-like: concentrated Q(mu) (sd .1) around 0.5, users' z around their own means (sd 1), sigma ~ exp(N(-1, .1))
mu = (0.5 + 0.1 * f(K, E)); psi = -1 + 0.1 * f(K, E); zc = f(M, 1, E); v = (zc + 0.3 * f(M, K, E)).astype(np.float32)
run(v, mu, np.exp(psi), small, "trained-like")
# 3. judge's adversarial: loc ~ 10 + .05 noise, sigma .05, v ~ N(loc, sigma)
mu = (10 + 0.05 * f(K, E)); sig = np.full((K, E), 0.05, np.float32); v = (10 + 0.05 * f(M, K, E)).astype(np.float32)
run(v, mu, sig, small, "loc ~ 10 +- .05, sigma .05")
# 4. nasty: loc spread 10, sigma .05, v hits one loc row per k
mu = 10 * f(K, E); v = (mu[rng.integers(0, K, (M, K))] + 0.05 * f(M, K, E)).astype(np.float32)
run(v, mu, sig, small, "loc spread 10, sigma .05, v on a loc row")
# 5. E = 1 scalar latents (bus_breakdown alpha ~ N(beta, exp(sigma_alpha)))
E = 1
mu, psi, v, small = f(K, E), f(K, E), f(M, K, E), f(M, K)
run(v, mu, np.exp(psi), small, "scalar latent init")
mu = 3 * f(K, E); sig = np.full((K, E), 0.02, np.float32); v = (mu[rng.integers(0, K, (M, K))] + 0.02 * f(M, K, E)).astype(np.float32)
run(v, mu, sig, small, "scalar latent, loc spread 3, sigma .02")
