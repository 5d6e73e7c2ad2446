// alan_normal_lse: the S-ML plate step with the factor producer fused in --
//
//   out[l, s] = sum_m  LSE_k( log N(value[m,k,:]; loc[l,:], scale[s,:]) + sum_f small_f[m,k] )  + add_const
//
// i.e. movielens' plate_1:  z[plate_1, K_z, d] ~ N(mu_z[K_mu, d], exp(psi_z)[K_psi, d]) with the -(log Q + log K)
// and data-likelihood terms as the small factors, log-sum-exp over K_z, sum over plate_1
// (TorchDimDist.py:127-162 + utils.py:147-152 + reduce_Ks.py:249-251 + utils.py:218-220 + logpq.py:149).
// The factor F[plate_1, K_mu, K_psi, K_z] (32 MB at K=30, 1.2 GB at K=100) is never written or read: the kernel is
// bound by its exps and MFMAs, not by HBM.
//
// Per-chunk partial sums go to the workspace; a small second stage adds the chunks (no float atomics).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "plan.h"

namespace alan {

struct NLDesc {
    const float *val, *loc, *scl;
    float *part;                      // [n_chunks][NL][NS]
    float *lse;                       // optional [M][NL][NS]: the per-plate-element log-sum-exp (the backward's input)
    int32_t M, NK, NL, NS, E, m_chunk, n_small, log_scale;
    int32_t small_f64;                // bit f: small factor f is fp64 (converted on load)
    int64_t v_sm, v_sk, v_se, l_sl, l_se, s_ss, s_se;
    const void *small[4];
    int64_t small_sm[4], small_sk[4];
};

// small factor f at element offset off (fp32, or fp64 converted on load: the likelihood of fp64 observations)
__device__ __forceinline__ float nl_small(const void *p, int64_t off, bool f64) {
    return f64 ? (float)((const double *)p)[off] : ((const float *)p)[off];
}

// On the matrix cores: for one (plate element m, loc row l) the block F[m, l, :, :] is a GEMM over the event dim;
// with v_mfma_f32_32x32x2_f32 computing D[i = k][j = s] = sum_e d2[(m,k), e] * w[s, e] the log-sum-exp over
// k runs DOWN the accumulator registers of a lane (16 rows per lane + one exchange between the two half-waves), the
// sum over m stays in a register, and nothing but out[l, s] partials is ever stored:
//   A (one VGPR per step): d2 of this lane's k row, rebuilt per plate element with two VALU ops per event pair;
//   B (one VGPR per step, held for the whole kernel): w of this lane's scale row;  C of the first MFMA: lg[s];
//   one extra event slot carries the small factors: A = -sum_f small_f[m,k], B = 1.
// A wave owns one (l, tile of 32 scale rows) pair and a chunk of the plate; k runs in tiles of 32 with an online
// log-sum-exp across tiles; the value rows of the next tile are loaded while the current one is multiplied.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EH>       // MFMA steps: ceil((E + 1) / 2) -- the event dim plus the small-factor slot
__global__ __launch_bounds__(256) void normal_lse_mfma_kernel(const NLDesc d) {
    extern __shared__ __align__(16) float lds[];
    constexpr int EP = 2 * EH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS;
    float *wt = lds;                                  // [NS][EP]   1/(2 scale^2), pads 0
    float *lgs = wt + (size_t)NS * EP;                // [NS][EP]   log scale, pads 0
    float *lgt = lgs + (size_t)NS * EP;               // [NS]       sum_e log scale + E log sqrt(2 pi)
    const int nst = (NS + 31) >> 5, nkt = (NK + 31) >> 5;
    const int q = blockIdx.x * 4 + wave;              // (l, scale tile) pair of this wave
    const bool wave_ok = q < d.NL * nst;
    const int l = wave_ok ? q / nst : 0, st = wave_ok ? q - l * nst : 0;
    const int s = 32 * st + j;
    const bool s_ok = s < NS;
    const bool small_slot = (E & 1) == h;             // this lane's element of the last step is the small-factor slot
    // everything this wave needs from global memory is requested before the tables are built: the loc row ...
    float mreg[EH];
#pragma unroll
    for (int step = 0; step < EH; ++step)
        mreg[step] = d.loc[(int64_t)l * d.l_sl + (int64_t)min(2 * step + h, E - 1) * d.l_se];
    // ... and (below) the first tile; meanwhile the workgroup's tables, one scale element per thread and pass
    for (int i = tid; i < NS * EP; i += 256) {
        const int is = i / EP, e = i - is * EP;
        float w = 0.f, lg = 0.f;
        if (e < E) {
            const float x = d.scl[(int64_t)is * d.s_ss + (int64_t)e * d.s_se];
            w = d.log_scale ? 0.5f * expf(-2.f * x) : 0.5f / (x * x);
            lg = d.log_scale ? x : logf(x);
        }
        wt[i] = w;
        lgs[i] = lg;
    }
    __syncthreads();
    for (int is = tid; is < NS; is += 256) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < EP; ++e) a += lgs[is * EP + e];
        lgt[is] = a + (float)E * 0.91893853320467274178f;
    }
    __syncthreads();
    if (!wave_ok) return;                             // (no barriers below)
    float breg[EH];
#pragma unroll
    for (int step = 0; step < EH; ++step) {
        const int e = 2 * step + h;
        breg[step] = (s_ok && e < E) ? wt[(size_t)s * EP + e] : (s_ok && e == E) ? 1.f : 0.f;
        mreg[step] = e < E ? mreg[step] : 0.f;
    }
    f32x16 cinit;
    {
        const float lg = s_ok ? lgt[s] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = lg;
    }
    const int m0 = blockIdx.y * d.m_chunk, m1 = min(d.M, m0 + d.m_chunk);
    const int n_tiles = (m1 - m0) * nkt;
    // this lane's slice of tile t, RAW: value[m, k, 2 step + h] for k = 32 kt + j and the small factors at (m, k), from
    // clamped (always valid) addresses, every load issued before anything waits; masks are applied where the values
    // are used, one iteration later (selects next to the loads made the compiler branch around each pair of loads
    // and wait for it on the spot)
    auto load_tile = [&](int t, float (&x)[EH], float (&hs)[4]) {
        const int m = m0 + t / nkt, kt_ = t - (t / nkt) * nkt;
        const int k = min(32 * kt_ + j, NK - 1);
        const float *vp = d.val + (int64_t)m * d.v_sm + (int64_t)k * d.v_sk;
#pragma unroll
        for (int step = 0; step < EH; ++step) x[step] = vp[(int64_t)min(2 * step + h, E - 1) * d.v_se];
#pragma unroll
        for (int f = 0; f < 4; ++f)                   // (the launcher points unused slots at valid memory, stride 0)
            hs[f] = nl_small(d.small[f], (int64_t)m * d.small_sm[f] + (int64_t)k * d.small_sk[f], (d.small_f64 >> f) & 1);
        asm volatile("" ::: "memory");
    };
    float zc[EH], zn[EH], hc[4], hn[4];
    if (n_tiles > 0) load_tile(0, zc, hc);
    float accm = 0.f, mn = __builtin_huge_valf(), sm = 0.f;
    int kt = 0;
    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) load_tile(t + 1, zn, hn);
        // A operand.  No masks: pad events meet a zero in B; rows beyond NK (and plate elements' -inf small factors)
        // put +inf into the small-factor slot, which makes their whole row of D +inf = a log-prob of -inf
        const bool k_ok = 32 * kt + j < NK;
        float a[EH];
#pragma unroll
        for (int step = 0; step < EH; ++step) {
            const float df = zc[step] - mreg[step];
            a[step] = df * df;
        }
        {   // the small-factor slot is element E = 2 (EH - 1) + (E & 1): always in the LAST step's register
            float hsum = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f) hsum += f < d.n_small ? hc[f] : 0.f;
            if (small_slot) a[EH - 1] = k_ok ? -hsum : __builtin_huge_valf();
        }
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], breg[0], cinit, 0, 0, 0);
#pragma unroll
        for (int step = 1; step < EH; ++step) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step], breg[step], acc, 0, 0, 0);
        // acc[r] = -log-prob of row k = 32 kt + (r & 3) + 8 (r >> 2) + 4 h: log-sum-exp down the registers
        // (utils.py:218-220), kept on u = -value: running minimum mn = -max, sm = sum exp(mn - u)
        float tmin = __builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < 16; ++r) tmin = fminf(tmin, acc[r]);
        if (tmin < mn) {
            sm *= __expf(tmin - mn);                  // (mn = +inf at the start: exp(-inf) = 0, and sm is 0 anyway)
            mn = tmin;
        }
        if (mn != __builtin_huge_valf()) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += __expf(mn - acc[r]);
        }
        if (++kt == nkt) {                            // plate element done: join the two half-waves, add to the plate sum
            float mx = -mn, mx2 = -__shfl_xor(mn, 32), sm2 = __shfl_xor(sm, 32);
            lse_merge(mx, sm, mx2, sm2);
            const float lse_m = lse_finish(mx, sm);
            accm += lse_m;
            if (d.lse && h == 0 && s_ok) d.lse[((int64_t)(m0 + t / nkt) * d.NL + l) * NS + s] = lse_m;
            mn = __builtin_huge_valf(), sm = 0.f, kt = 0;
        }
#pragma unroll
        for (int step = 0; step < EH; ++step) zc[step] = zn[step];
#pragma unroll
        for (int f = 0; f < 4; ++f) hc[f] = hn[f];
    }
    if (h == 0 && s_ok) d.part[((int64_t)blockIdx.y * d.NL + l) * NS + s] = accm;
}

}  // namespace alan

using namespace alan;

namespace {

struct NLPlan {
    int eh = 0, m_chunk = 1, n_chunks = 1;
    size_t lds = 0, part_bytes = 0;
    dim3 grid;
};

int plan_nl(const alan_normal_lse_desc_t &a, NLPlan &p) {
    if (!a.value || !a.loc || !a.scale || !a.out) return ALAN_ERR_BAD_DESC;
    if (a.M < 1 || a.NK < 1 || a.NL < 1 || a.NS < 1 || a.E < 1) return ALAN_ERR_BAD_DESC;
    if (a.n_small < 0 || a.n_small > 4) return ALAN_ERR_BAD_DESC;
    for (int f = 0; f < a.n_small; ++f) {
        if (!a.small[f]) return ALAN_ERR_BAD_DESC;
        if (a.small_dtype[f] != ALAN_F32 && a.small_dtype[f] != ALAN_F64) return ALAN_ERR_BAD_DESC;
    }
    if (a.out_dtype != ALAN_F32 && a.out_dtype != ALAN_F64) return ALAN_ERR_BAD_DESC;
    if (a.E > 32 || a.NK > 4096 || a.NS > 4096 || a.NL > (1 << 20) || a.M > (1 << 24)) return ALAN_ERR_UNSUPPORTED;
    // a wave per (loc row, tile of 32 scale rows), 4 per workgroup; the plate in chunks so that ~4096 waves exist
    p.eh = (int)(a.E + 2) / 2;
    const int64_t nst = (a.NS + 31) / 32;
    const int64_t gx = (a.NL * nst + 3) / 4;
    int64_t target = 1024;                                                   // workgroups (x 4 waves)
    static const int blocks_knob = env_knob("ALAN_NLSE_BLOCKS");                      // tuning knob
    if (blocks_knob != ENV_UNSET) target = std::max(1, blocks_knob);
    int64_t nch = std::max<int64_t>(1, std::min<int64_t>(a.M, target / std::max<int64_t>(1, gx)));
    p.m_chunk = (int)((a.M + nch - 1) / nch);
    p.n_chunks = (int)((a.M + p.m_chunk - 1) / p.m_chunk);
    p.lds = ((size_t)a.NS * 4 * p.eh + a.NS + 8) * sizeof(float);
    if (p.lds > 150 * 1024) return ALAN_ERR_UNSUPPORTED;
    p.part_bytes = (size_t)p.n_chunks * a.NL * a.NS * sizeof(float);
    p.grid = dim3((uint32_t)gx, (uint32_t)p.n_chunks);
    return ALAN_OK;
}

}  // namespace

extern "C" size_t alan_normal_lse_workspace_bytes(const alan_normal_lse_desc_t *a) {
    if (!a) return 0;
    NLPlan p;
    if (plan_nl(*a, p) != ALAN_OK) return 0;
    return (p.part_bytes + 255) & ~(size_t)255;
}

extern "C" int alan_normal_lse(const alan_normal_lse_desc_t *a, void *workspace, size_t workspace_bytes,
                               void *stream_) {
    if (!a) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    NLPlan p;
    int rc = plan_nl(*a, p);
    if (rc != ALAN_OK) return rc;
    if (!workspace || workspace_bytes < p.part_bytes) return ALAN_ERR_WORKSPACE;
    NLDesc d;
    std::memset(&d, 0, sizeof(d));
    d.val = (const float *)a->value;
    d.loc = (const float *)a->loc;
    d.scl = (const float *)a->scale;
    d.part = (float *)workspace;
    d.lse = (float *)a->lse_out;
    d.M = (int)a->M, d.NK = (int)a->NK, d.NL = (int)a->NL, d.NS = (int)a->NS, d.E = (int)a->E;
    d.m_chunk = p.m_chunk, d.n_small = a->n_small, d.log_scale = a->log_scale;
    d.v_sm = a->v_sm, d.v_sk = a->v_sk, d.v_se = a->v_se;
    d.l_sl = a->l_sl, d.l_se = a->l_se, d.s_ss = a->s_ss, d.s_se = a->s_se;
    for (int f = 0; f < 4; ++f) {
        const bool used = f < a->n_small;
        d.small[f] = used ? a->small[f] : a->value;                                  // (unused: any valid address)
        if (used && a->small_dtype[f] == ALAN_F64) d.small_f64 |= 1 << f;
        d.small_sm[f] = used ? a->small_sm[f] : 0;
        d.small_sk[f] = used ? a->small_sk[f] : 0;
    }
    auto launch = [&](auto kern) {
        if (p.lds > 64 * 1024)
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) !=
                hipSuccess)
                return ALAN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, p.grid, dim3(256), p.lds, stream, d);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
    switch (p.eh) {
        case 1: rc = launch(normal_lse_mfma_kernel<1>); break;
        case 2: rc = launch(normal_lse_mfma_kernel<2>); break;
        case 3: rc = launch(normal_lse_mfma_kernel<3>); break;
        case 4: rc = launch(normal_lse_mfma_kernel<4>); break;
        case 5: rc = launch(normal_lse_mfma_kernel<5>); break;
        case 6: rc = launch(normal_lse_mfma_kernel<6>); break;
        case 7: rc = launch(normal_lse_mfma_kernel<7>); break;
        case 8: rc = launch(normal_lse_mfma_kernel<8>); break;
        case 9: rc = launch(normal_lse_mfma_kernel<9>); break;
        case 10: rc = launch(normal_lse_mfma_kernel<10>); break;
        case 11: rc = launch(normal_lse_mfma_kernel<11>); break;
        case 12: rc = launch(normal_lse_mfma_kernel<12>); break;
        case 13: rc = launch(normal_lse_mfma_kernel<13>); break;
        case 14: rc = launch(normal_lse_mfma_kernel<14>); break;
        case 15: rc = launch(normal_lse_mfma_kernel<15>); break;
        case 16: rc = launch(normal_lse_mfma_kernel<16>); break;
        default: rc = launch(normal_lse_mfma_kernel<17>);
    }
    if (rc != ALAN_OK) return rc;

    // ---- second stage: out[l, s] = sum_chunk part[chunk, l, s] + add_const
    Canon s2;
    s2.nf = 1;
    s2.dominant = 0;
    s2.f[0].p = workspace;
    s2.f[0].dtype = ALAN_F32;
    s2.f[0].scale = 1.f;
    s2.w.p = nullptr;
    s2.l.p = nullptr;
    s2.o.p = a->out;
    s2.o.dtype = a->out_dtype;
    s2.o.scale = 1.f;
    for (int j = 0; j < MAXD; ++j) s2.f[0].ks[j] = s2.f[0].rs[j] = s2.o.ks[j] = 0;
    s2.nk = 0, s2.nr = 0, s2.n_out = 1, s2.n_red = 1;
    auto keep_dim = [&](int64_t size, int64_t src, int64_t dst) {
        if (size <= 1) return;
        s2.ksize[s2.nk] = size, s2.f[0].ks[s2.nk] = src, s2.o.ks[s2.nk] = dst, s2.kplate[s2.nk] = false;
        ++s2.nk;
        s2.n_out *= size;
    };
    if (p.n_chunks > 1) {
        s2.rsize[0] = p.n_chunks;
        s2.f[0].rs[0] = a->NL * a->NS;
        s2.nr = 1;
        s2.n_red = p.n_chunks;
    }
    keep_dim(a->NL, a->NS, a->o_sl);
    keep_dim(a->NS, 1, a->o_ss);
    s2.red_contig = false;
    s2.keep_contig = true;
    GroupDesc gd;
    GroupLaunch gl;
    rc = plan_group(s2, a->out_dtype, a->add_const, gd, gl);
    if (rc != ALAN_OK) return rc;
    rc = try_launch_small(s2, gd, gl, ALAN_MODE_SUM, ALAN_F32, stream, EvPair());
    if (rc == ALAN_ERR_UNSUPPORTED) rc = launch_group(gd, gl, ALAN_MODE_SUM, ALAN_F32, stream);
    return rc;
}
