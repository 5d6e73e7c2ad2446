// alan_normal_lse: the S-ML plate step with the factor producer fused in --
//
//   out[l, s] = sum_m  LSE_k( log N(value[m,k,:]; loc[l,:], scale[s,:]) + sum_f small_f[m,k] )  + add_const
//
// i.e. movielens' plate_1:  z[plate_1, K_z, d] ~ N(mu_z[K_mu, d], exp(psi_z)[K_psi, d]) with the -(log Q + log K)
// and data-likelihood terms as the small factors, log-sum-exp over K_z, sum over plate_1
// (TorchDimDist.py:127-162 + utils.py:147-152 + reduce_Ks.py:249-251 + utils.py:218-220 + logpq.py:149).
// The factor F[plate_1, K_mu, K_psi, K_z] (32 MB at K=30, 1.2 GB at K=100) is never written or read: the kernel is
// bound by its FMAs and LDS reads, not by HBM.
//
// Mapping: a half-wave (32 lanes) owns one (l, block of 32 scale rows) pair -- lane = scale row s, its
// w[s,:] = 1/(2 scale^2) in registers -- and walks the plate elements of its chunk; per plate element the
// workgroup stages value[m,:,:] in LDS, each half-wave turns it into d2[k,:] = (value[m,k,:] - loc[l,:])^2 in
// its own LDS tile, then per k: 5 broadcast 16-byte reads + 20 FMAs + a one-exp online log-sum-exp update.
// Per-chunk partial sums go to the workspace; a small second stage adds the chunks (no float atomics).
#include <algorithm>
#include <cstring>

#include "plan.h"

namespace alan {

struct NLDesc {
    const float *val, *loc, *scl;
    float *part;                      // [n_chunks][NL][NS]
    int32_t M, NK, NL, NS, E, nsb, m_chunk, n_small, log_scale;
    int64_t v_sm, v_sk, v_se, l_sl, l_se, s_ss, s_se;
    const float *small[4];
    int64_t small_sm[4], small_sk[4];
};

constexpr int NL_UNITS = 8;   // half-waves per workgroup

// SPL = scale rows per lane (lane's rows: s0, s0 + 32, ...): every broadcast read of the d2 tile feeds 4*SPL FMAs.
template <int EMAX, int SPL>
__global__ __launch_bounds__(256) void normal_lse_kernel(const NLDesc d) {
    extern __shared__ __align__(16) float lds[];
    constexpr int EP = EMAX;
    const int tid = threadIdx.x, lane = tid & 31, u = tid >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS;
    float *wt = lds;                                  // [NS][EP]   1/(2 scale^2), pads 0
    float *lgt = wt + (size_t)NS * EP;                // [NS]       sum_e log scale + E log sqrt(2 pi)
    float *mus = lgt + ((NS + 3) & ~3);               // [8][EP]    loc rows of the units
    float *zs = mus + NL_UNITS * EP;                  // [NK][EP]   value[m,:,:], pads 0
    float *hs = zs + (size_t)NK * EP;                 // [NK]       sum of the small factors at m
    float *dds = hs + ((NK + 3) & ~3);                // [8][NK][EP]

    const int q = blockIdx.x * NL_UNITS + u;          // (l, s-block) pair of this half-wave
    const int l = q / d.nsb, sb = q - l * d.nsb;
    const bool unit_ok = l < d.NL;
    const int s0 = sb * (32 * SPL) + lane;           // this lane's scale rows: s0 + 32 j
    const int lc = min(l, d.NL - 1);
    const int m0 = blockIdx.y * d.m_chunk, m1 = min(d.M, m0 + d.m_chunk);

    // ---- tables
    for (int i = tid; i < NS * EP; i += 256) {
        const int is = i / EP, e = i - is * EP;
        float w = 0.f;
        if (e < E) {
            const float x = d.scl[(int64_t)is * d.s_ss + (int64_t)e * d.s_se];
            w = d.log_scale ? 0.5f * expf(-2.f * x) : 0.5f / (x * x);
        }
        wt[i] = w;
    }
    for (int is = tid; is < NS; is += 256) {
        float a = 0.f;
        for (int e = 0; e < E; ++e) {
            const float x = d.scl[(int64_t)is * d.s_ss + (int64_t)e * d.s_se];
            a += d.log_scale ? x : logf(x);
        }
        lgt[is] = a + (float)E * 0.91893853320467274178f;
    }
    if (lane < EP) mus[u * EP + lane] = (lane < E) ? d.loc[(int64_t)lc * d.l_sl + (int64_t)lane * d.l_se] : 0.f;
    if (EP > 32 && lane + 32 < EP)
        mus[u * EP + lane + 32] = (lane + 32 < E) ? d.loc[(int64_t)lc * d.l_sl + (int64_t)(lane + 32) * d.l_se] : 0.f;
    __syncthreads();
    float4 w4[SPL][EMAX / 4];
    float lgs[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int sc = min(s0 + 32 * j, NS - 1);
#pragma unroll
        for (int qd = 0; qd < EMAX / 4; ++qd) w4[j][qd] = reinterpret_cast<const float4 *>(wt + (size_t)sc * EP)[qd];
        lgs[j] = lgt[sc];
    }

    float *dd = dds + (size_t)u * NK * EP;
    float accm[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) accm[j] = 0.f;
    for (int m = m0; m < m1; ++m) {
        __syncthreads();                              // everyone is done with zs / hs / dds of the previous m
        for (int i = tid; i < NK * EP; i += 256) {
            const int k = i / EP, e = i - k * EP;
            zs[i] = e < E ? d.val[(int64_t)m * d.v_sm + (int64_t)k * d.v_sk + (int64_t)e * d.v_se] : 0.f;
        }
        for (int k = tid; k < NK; k += 256) {
            float h = 0.f;
            for (int f = 0; f < d.n_small; ++f) h += d.small[f][(int64_t)m * d.small_sm[f] + (int64_t)k * d.small_sk[f]];
            hs[k] = h;
        }
        __syncthreads();
        for (int i = lane; i < NK * EP; i += 32) {    // this half-wave's d2 tile (pads: (0 - 0)^2 = 0)
            const int e = i % EP;
            const float df = zs[i] - mus[u * EP + e];
            dd[i] = df * df;
        }
        __syncthreads();
        float mx[SPL], sm[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) mx[j] = -__builtin_huge_valf(), sm[j] = 0.f;
        for (int k = 0; k < NK; ++k) {
            const float4 *d4 = reinterpret_cast<const float4 *>(dd + (size_t)k * EP);
            float4 dv[EMAX / 4];
#pragma unroll
            for (int qd = 0; qd < EMAX / 4; ++qd) dv[qd] = d4[qd];
            const float hk = hs[k];
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int qd = 0; qd < EMAX / 4; ++qd) {
                    acc = fmaf(dv[qd].x, w4[j][qd].x, acc);
                    acc = fmaf(dv[qd].y, w4[j][qd].y, acc);
                    acc = fmaf(dv[qd].z, w4[j][qd].z, acc);
                    acc = fmaf(dv[qd].w, w4[j][qd].w, acc);
                }
                lse_push(mx[j], sm[j], (-acc - lgs[j]) + hk);     // (packed v_pk_fma_f32 over row pairs was slower)
            }
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) accm[j] += lse_finish(mx[j], sm[j]);
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = s0 + 32 * j;
        if (unit_ok && s < NS) d.part[((int64_t)blockIdx.y * d.NL + l) * NS + s] = accm[j];
    }
}

}  // namespace alan

using namespace alan;

namespace {

struct NLPlan {
    int em = 0, nsb = 1, spl = 1, m_chunk = 1, n_chunks = 1;
    size_t lds = 0, part_bytes = 0;
    dim3 grid;
};

int plan_nl(const alan_normal_lse_desc_t &a, NLPlan &p) {
    if (!a.value || !a.loc || !a.scale || !a.out) return ALAN_ERR_BAD_DESC;
    if (a.M < 1 || a.NK < 1 || a.NL < 1 || a.NS < 1 || a.E < 1) return ALAN_ERR_BAD_DESC;
    if (a.n_small < 0 || a.n_small > 4) return ALAN_ERR_BAD_DESC;
    for (int f = 0; f < a.n_small; ++f)
        if (!a.small[f]) return ALAN_ERR_BAD_DESC;
    if (a.E > 32 || a.NK > 4096 || a.NS > 4096 || a.NL > (1 << 20) || a.M > (1 << 24)) return ALAN_ERR_UNSUPPORTED;
    static const int kEmax[] = {4, 8, 12, 16, 20, 24, 28, 32};
    for (int em : kEmax)
        if (a.E <= em) {
            p.em = em;
            break;
        }
    // scale rows per lane: with one row per lane and > 32 rows every (m, l) pair would be walked by several
    // half-waves, each rebuilding and re-reading the d2 tile (measured at K=100: 110 us per 38-user chunk against
    // 49 + 32 us for the producer + rows kernels) -- block 2 or 4 rows per lane instead (65 us)
    p.spl = a.NS <= 32 ? 1 : a.NS <= 64 ? 2 : 4;
    if (a.NS > 128) return ALAN_ERR_UNSUPPORTED;
    p.nsb = 1;
    const int64_t gx = (a.NL * p.nsb + NL_UNITS - 1) / NL_UNITS;
    int64_t nch = std::max<int64_t>(1, std::min<int64_t>(a.M, 2048 / std::max<int64_t>(1, gx)));
    p.m_chunk = (int)((a.M + nch - 1) / nch);
    p.n_chunks = (int)((a.M + p.m_chunk - 1) / p.m_chunk);
    const size_t fl = (size_t)a.NS * p.em + ((a.NS + 3) & ~3) + (size_t)NL_UNITS * p.em + (size_t)a.NK * p.em +
                      ((a.NK + 3) & ~3) + (size_t)NL_UNITS * a.NK * p.em + 8;
    p.lds = fl * sizeof(float);
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
    d.M = (int)a->M, d.NK = (int)a->NK, d.NL = (int)a->NL, d.NS = (int)a->NS, d.E = (int)a->E;
    d.nsb = p.nsb, d.m_chunk = p.m_chunk, d.n_small = a->n_small, d.log_scale = a->log_scale;
    d.v_sm = a->v_sm, d.v_sk = a->v_sk, d.v_se = a->v_se;
    d.l_sl = a->l_sl, d.l_se = a->l_se, d.s_ss = a->s_ss, d.s_se = a->s_se;
    for (int f = 0; f < a->n_small; ++f) {
        d.small[f] = (const float *)a->small[f];
        d.small_sm[f] = a->small_sm[f];
        d.small_sk[f] = a->small_sk[f];
    }
    auto launch = [&](auto kern) {
        if (p.lds > 64 * 1024)
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) !=
                hipSuccess)
                return ALAN_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, p.grid, dim3(256), p.lds, stream, d);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
#define ALAN_NL_CASE(EM)                                                    \
    case EM:                                                                \
        rc = p.spl == 1 ? launch(normal_lse_kernel<EM, 1>)                  \
             : p.spl == 2 ? launch(normal_lse_kernel<EM, 2>)                \
                          : launch(normal_lse_kernel<EM, 4>);               \
        break
    switch (p.em) {
        ALAN_NL_CASE(4);
        ALAN_NL_CASE(8);
        ALAN_NL_CASE(12);
        ALAN_NL_CASE(16);
        ALAN_NL_CASE(20);
        ALAN_NL_CASE(24);
        ALAN_NL_CASE(28);
        default: rc = p.spl == 1 ? launch(normal_lse_kernel<32, 1>)
                      : p.spl == 2 ? launch(normal_lse_kernel<32, 2>) : launch(normal_lse_kernel<32, 4>);
    }
#undef ALAN_NL_CASE
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
    s2.o.dtype = ALAN_F32;
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
    rc = plan_group(s2, ALAN_F32, a->add_const, gd, gl);
    if (rc != ALAN_OK) return rc;
    rc = try_launch_small(s2, gd, gl, ALAN_MODE_SUM, ALAN_F32, stream, EvPair());
    if (rc == ALAN_ERR_UNSUPPORTED) rc = launch_group(gd, gl, ALAN_MODE_SUM, ALAN_F32, stream);
    return rc;
}
