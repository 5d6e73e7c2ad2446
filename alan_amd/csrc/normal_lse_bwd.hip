// alan_normal_lse_backward: every gradient of the fused plate step (normal_lse.hip) in one pass that never writes
// the [M, NL, NS, NK] factor or its gradient -- what autograd derives from TorchDimDist.py:127-162
// (torch.distributions.Normal.log_prob) + utils.py:147-152 + reduce_Ks.py:249-251 + utils.py:218-220 + logpq.py:149.
//
//   X[m,l,s,k] = G[l,s] * exp(log N(value[m,k,:]; loc[l,:], scale[s,:]) + small[m,k] - lse[m,l,s])     (never stored)
//   d small[m,k]   = sum_{l,s}   X
//   d value[m,k,e] = sum_{l,s}   X * -2 (v - mu) w          w[s,e] = 1 / (2 scale^2)
//   d loc[l,e]     = sum_{m,k,s} X *  2 (v - mu) w
//   d log scale[s,e] = sum_{m,k,l} X * (2 w (v - mu)^2 - 1)          (d scale = that / scale)
//
// One 32x32 tile of X per (m, k tile, l, s tile), recomputed on the matrix cores exactly as the forward does
// (D[k,s] = -log-prob, C operand = the log-normaliser), then three uses of it, all as v_mfma_f32_32x32x2_f32:
//   V[s,e] += sum_k X[k,s] * d2[k,e]        X's ROW index is summed: the accumulator registers ARE the A operand
//   U[k,e]  = sum_s X[k,s] * w'[s,e]        X's COLUMN (lane) index is summed: one transpose through LDS (wave-private,
//                                           32 x 36 floats, written by ds_write_b32, read back as 4 ds_read_b128)
//   T[k,e]  = (v - mu)[k,e] * U[k,e]:  d value -= 2 T (summed over l, s in registers),  d loc[l,e] = 2 sum_k T
// Column e = E of both products multiplies by ones instead: V[s,E] = sum_k X (the "-1" term of d log scale) and
// U[k,E] = sum_s X = d small.  A workgroup owns (m, k tile) and a share of the loc rows, so d value / d small
// leave complete; d loc / d scale leave as per-workgroup partials that a column-sum kernel adds (no float atomics).
#include <algorithm>
#include <cstring>

#include "plan.h"
#include "normal_lse_x3.h"

namespace alan {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NLBDesc {
    const float *val, *loc, *scl, *lse, *gout;
    float *dval;      // [gy][M][NK][E]   (gy == 1: the caller's grad_value itself)
    float *dsm;       // [gy][M][NK]
    float *ploc;      // [M * nkt * nst][NL][E]
    float *pscl;      // [M * nkt * gy][NS][E]
    int32_t M, NK, NL, NS, E, n_small, log_scale;
    int64_t v_sm, v_sk, v_se, l_sl, l_se, s_ss, s_se, g_sl, g_ss;
    const float *small[4];
    int64_t small_sm[4], small_sk[4];
};

constexpr int NLB_TS = 36;                 // row stride (floats) of a wave's 32 x 32 transpose tile: 144 B, 16-byte aligned
constexpr int NLB_SCR = 32 * NLB_TS;       // floats per wave

struct NLBLds {
    int eps, nsp, tss, o_lgt, o_wtT, o_mu, o_scr, total;
};
__host__ __device__ inline NLBLds nlb_lds(int EH, int NS, int NL, int E) {
    NLBLds L;
    L.eps = 2 * EH + 1;                    // odd row stride: the per-lane reads of a scale row are conflict-free
    L.nsp = ((NS + 31) / 32) * 32;
    L.tss = L.nsp + 4;
    L.o_lgt = (L.nsp * L.eps + 3) & ~3;
    L.o_wtT = L.o_lgt + L.nsp;
    L.o_mu = L.o_wtT + (E + 1) * L.tss;
    L.o_scr = (L.o_mu + NL * E + 3) & ~3;
    // scratch: the four waves' transpose tiles; reused for the cross-wave sums (4 x 16 x 64) and, while the tables
    // are built, for the log-scale values (nsp x eps)
    int scr = 4 * NLB_SCR > 4 * 16 * 64 ? 4 * NLB_SCR : 4 * 16 * 64;
    if (L.nsp * L.eps > scr) scr = L.nsp * L.eps;
    L.total = L.o_scr + scr;
    return L;
}

// SMALL_ONLY: only d small is wanted (elbo_rws: the sample is detached, log Q carries the gradient) -- no V, no U.
// FLAT (as the forward's flat row tiling; NK > 32, NK % 4 == 0, plate elements contiguous): the k rows of all plate
// elements are ONE run of M NK rows cut into tiles of 32, so only the run's last tile is padded (K = 100: 938 tiles
// where a tile grid per plate element has 1,200).  A tile then ends one plate element (rows below `bnd`) and begins the
// next: the forward's log-sum-exp of either is held per lane and chosen per group of four accumulator rows; everything
// else -- V, U, d value, d small -- is indifferent to which plate element a row belongs to.
// X2 (round 3): the V and U products -- 32 of a tile's 42 matrix steps -- on v_mfma_f32_32x32x16_bf16 with 2-way split
// operands (x = hi + lo, both bf16, round-to-nearest: 16 mantissa bits; the product as hi hi + hi lo + lo hi, every
// bf16 x bf16 product exact in the fp32 accumulator: relative error <= 2^-15 per term where the fp32 chain has 2^-24, far
// inside what a gradient is compared at).  Six instructions of 32 cycles replace sixteen of 64 per product, and -- unlike
// the fp32 matrix instructions -- leave the vector unit running beside them; the operands are the registers the fp32 form
// used, eight per instruction: a lane half's rows (r = 8 t .. 8 t + 7) on both sides of the contraction.  The tile X itself
// (D chain, exps) stays fp32.
__device__ __forceinline__ void nlb_split8(const float (&x)[8], u32x4v &hi, u32x4v &lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h2 = nl_cvt_pk(x[2 * q], x[2 * q + 1]);
        hi[q] = h2;
        lo[q] = nl_cvt_pk(x[2 * q] - __uint_as_float(h2 << 16), x[2 * q + 1] - __uint_as_float(h2 & 0xffff0000u));
    }
}
__device__ __forceinline__ f32x16 nlb_mfma_x2(const u32x4v &ah, const u32x4v &al, const u32x4v &bh, const u32x4v &bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, ah), __builtin_bit_cast(bf16x8v, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, ah), __builtin_bit_cast(bf16x8v, bl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, al), __builtin_bit_cast(bf16x8v, bh), c, 0, 0, 0);
    return c;
}

template <int EH, bool SMALL_ONLY, bool FLAT = false, bool X2 = false>
__global__ __launch_bounds__(256) void normal_lse_bwd_kernel(const NLBDesc d) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS, NL = d.NL;
    const NLBLds L = nlb_lds(EH, NS, NL, E);
    float *wt = lds, *lgt = lds + L.o_lgt, *wtT = lds + L.o_wtT, *mu = lds + L.o_mu, *scr = lds + L.o_scr;
    const int nkt = (NK + 31) >> 5, nst = L.nsp >> 5;
    // the tile's rows: (m, 32 kt ..) of one plate element, or rows 32 blockIdx.x .. of the flat run
    const int64_t rows_total = (int64_t)d.M * NK, row0 = FLAT ? 32ll * blockIdx.x : 0;
    const int m = FLAT ? (int)(row0 / NK) : blockIdx.x / nkt, kt = FLAT ? 0 : blockIdx.x - m * nkt;
    const int bnd = FLAT ? (int)std::min<int64_t>(32, (int64_t)(m + 1) * NK - row0) : 32;   // rows of the tile in element m
    const int ys = blockIdx.y, gy = gridDim.y;
    const int sE = E >> 1;                                  // step / half of event slot E (the small factors; ones)
    const bool slotE = (E & 1) == h;

    // ---- this lane's share of the tile, requested before the tables are built
    const int kA = 32 * kt + j;
    const bool k_ok = FLAT ? row0 + j < rows_total : kA < NK;
    int mj = m, kj = min(kA, NK - 1);                       // this lane's row: (plate element, k)
    if (FLAT) {
        const int64_t rj = std::min<int64_t>(row0 + j, rows_total - 1);
        mj = (int)(rj / NK), kj = (int)(rj - (int64_t)mj * NK);
    }
    const float *vrow = d.val + (int64_t)mj * d.v_sm + (int64_t)kj * d.v_sk;
    float vA[EH];
#pragma unroll
    for (int step = 0; step < EH; ++step) vA[step] = vrow[(int64_t)min(2 * step + h, E - 1) * d.v_se];
    float hsum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const float x = d.small[f][(int64_t)mj * d.small_sm[f] + (int64_t)kj * d.small_sk[f]];
        hsum += f < d.n_small ? x : 0.f;
    }
    // the same rows with the event index on the lanes: vT[r] = value[row r(h) of the tile, e = j], 1 in column E, 0 beyond
    float vT[16];
    if (!SMALL_ONLY) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            int64_t roff;
            if (FLAT) {
                const int64_t rf = std::min<int64_t>(row0 + rr, rows_total - 1);
                const int mr = (int)(rf / NK);
                roff = (int64_t)mr * d.v_sm + (rf - (int64_t)mr * NK) * d.v_sk;
            } else {
                roff = (int64_t)m * d.v_sm + (int64_t)min(32 * kt + rr, NK - 1) * d.v_sk;
            }
            const float x = d.val[roff + (int64_t)min(j, E - 1) * d.v_se];
            vT[r] = j < E ? x : (j == E ? 1.f : 0.f);
        }
    }
    // ---- tables (one scale element per thread and pass): w, log scale (in the transpose scratch for now), w transposed
    for (int i = tid; i < L.nsp * L.eps; i += 256) {
        const int is = i / L.eps, e = i - is * L.eps;
        float w = 0.f, lg = 0.f;
        if (is < NS && e < E) {
            const float x = d.scl[(int64_t)is * d.s_ss + (int64_t)e * d.s_se];
            w = d.log_scale ? 0.5f * expf(-2.f * x) : 0.5f / (x * x);
            lg = d.log_scale ? x : logf(x);
        } else if (is < NS && e == E) {
            w = 1.f;                                        // the small-factor slot of D / the ones column of U
        }
        wt[i] = w;
        scr[i] = lg;
        if (e <= E) wtT[e * L.tss + is] = w;
    }
    for (int i = tid; i < NL * E; i += 256) {
        const int il = i / E, e = i - il * E;
        mu[i] = d.loc[(int64_t)il * d.l_sl + (int64_t)e * d.l_se];
    }
    __syncthreads();
    for (int is = tid; is < L.nsp; is += 256) {
        float a = 0.f;
        for (int e = 0; e < 2 * EH; ++e) a += scr[is * L.eps + e];          // (same order as the forward's table)
        lgt[is] = a + (float)E * 0.91893853320467274178f;
    }
    __syncthreads();

    float *tile = scr + wave * NLB_SCR;
    f32x16 dvacc, vacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dvacc[r] = 0.f, vacc[r] = 0.f;

    for (int st = 0; st < nst; ++st) {
        const int s = 32 * st + j;
        const bool s_ok = s < NS;
        float breg[EH];
#pragma unroll
        for (int step = 0; step < EH; ++step) breg[step] = wt[s * L.eps + 2 * step + h];
        f32x16 cinit;
        {
            const float lg = s_ok ? lgt[s] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[r] = lg;
        }
        f32x4 wU[4];                                        // w'[32 st + 16 h + r', e = j] (0 beyond column E)
        if (!SMALL_ONLY) {
            const f32x4 *p = (const f32x4 *)(wtT + min(j, E) * L.tss + 32 * st + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wU[q] = p[q];
                if (j > E) wU[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (st > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) vacc[r] = 0.f;
            }
        }
        u32x4v wUh[2], wUl[2];                              // X2: w' of this scale tile, split once
        if (X2 && !SMALL_ONLY) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float w8[8] = {wU[2 * t2][0], wU[2 * t2][1], wU[2 * t2][2], wU[2 * t2][3],
                                     wU[2 * t2 + 1][0], wU[2 * t2 + 1][1], wU[2 * t2 + 1][2], wU[2 * t2 + 1][3]};
                nlb_split8(w8, wUh[t2], wUl[t2]);
            }
        }
        for (int l = ys * 4 + wave; l < NL; l += 4 * gy) {
            // per-lane scalars of this tile: upstream gradient and the forward's log-sum-exp at (m, l, s)
            const float Gs = s_ok ? d.gout[(int64_t)l * d.g_sl + (int64_t)s * d.g_ss] : 0.f;
            const float nls = s_ok ? -d.lse[((int64_t)m * NL + l) * NS + s] : 0.f;
            // (FLAT: the rows from `bnd` on belong to the next plate element)
            const float nls2 = (FLAT && bnd < 32 && s_ok && m + 1 < d.M) ? -d.lse[((int64_t)(m + 1) * NL + l) * NS + s] : 0.f;
            float a[EH];
#pragma unroll
            for (int step = 0; step < EH; ++step) {
                const int e = 2 * step + h;
                const float mr = e < E ? mu[l * E + min(e, E - 1)] : 0.f;
                const float df = vA[step] - mr;
                a[step] = df * df;
                if (step == sE && slotE) a[step] = k_ok ? -hsum : __builtin_huge_valf();
            }
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], breg[0], cinit, 0, 0, 0);
#pragma unroll
            for (int step = 1; step < EH; ++step)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step], breg[step], acc, 0, 0, 0);
            // acc[r] = -(log-prob + small) of row k_r(h), column s: X = G * exp(-lse - acc)
            // (lanes beyond the last scale row hold 0, not inf * 0: their columns enter U's sum over s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float nl = (FLAT && 8 * (r >> 2) + 4 * h >= bnd) ? nls2 : nls;     // (NK % 4 == 0: four rows of a register group go together)
                acc[r] = s_ok ? Gs * __expf(nl - acc[r]) : 0.f;
            }
            if (SMALL_ONLY) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dvacc[r] += acc[r];
                continue;
            }
            float dfT[16];
            {
                const float muT = j < E ? mu[l * E + min(j, E - 1)] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dfT[r] = vT[r] - muT;
            }
            // V[s, e] += sum_k X[k, s] d2[k, e]: step r pairs rows k_r(0), k_r(1) -- exactly what register r holds
            if (X2) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float x8[8], d8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x8[i] = acc[8 * t2 + i], d8[i] = dfT[8 * t2 + i] * dfT[8 * t2 + i];
                    u32x4v xh, xl, dh, dl;
                    nlb_split8(x8, xh, xl);
                    nlb_split8(d8, dh, dl);
                    vacc = nlb_mfma_x2(xh, xl, dh, dl, vacc);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    vacc = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[r], dfT[r] * dfT[r], vacc, 0, 0, 0);
            }
            // transpose X through the wave's LDS tile: [k][s] image, then 16 consecutive s per lane
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * h) * NLB_TS + j] = acc[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 xt[4];
            {
                const f32x4 *p = (const f32x4 *)(tile + j * NLB_TS + 16 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) xt[q] = p[q];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // U[k, e] = sum_s X[k, s] w'[s, e]: step r' pairs columns s = r' and 16 + r'
            f32x16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
            if (X2) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const float x8[8] = {xt[2 * t2][0], xt[2 * t2][1], xt[2 * t2][2], xt[2 * t2][3],
                                         xt[2 * t2 + 1][0], xt[2 * t2 + 1][1], xt[2 * t2 + 1][2], xt[2 * t2 + 1][3]};
                    u32x4v xh, xl;
                    nlb_split8(x8, xh, xl);
                    u = nlb_mfma_x2(xh, xl, wUh[t2], wUl[t2], u);
                    // (the first instruction of a chain from C = 0 may be given a destination over a dead operand under
                    // -amdgpu-mfma-vgpr-form: keep the operands alive past it, as the forward does)
                    if (t2 == 0) asm volatile("" ::"v"(u[0]), "v"(xh), "v"(xl), "v"(wUh[0]), "v"(wUl[0]));
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        u = __builtin_amdgcn_mfma_f32_32x32x2f32(xt[q][c], wU[q][c], u, 0, 0, 0);
            }
            // T = (v - mu) * U  (U itself in column E): d value / d small accumulate, d loc leaves as a partial
            float p = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tv = dfT[r] * u[r];
                dvacc[r] += tv;
                p += tv;
            }
            p += __shfl_xor(p, 32);
            if (d.ploc && h == 0 && j < E)
                d.ploc[(((int64_t)blockIdx.x * nst + st) * NL + l) * E + j] = 2.f * p;
        }
        if (!SMALL_ONLY && d.pscl) {
            // d (log) scale rows of this scale tile: the four waves' V summed through LDS, then 2 w V[s,e] - V[s,E]
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[(wave * 16 + r) * 64 + lane] = vacc[r];
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * wave + rr;
                const float tot = scr[(0 * 16 + r) * 64 + lane] + scr[(1 * 16 + r) * 64 + lane] +
                                  scr[(2 * 16 + r) * 64 + lane] + scr[(3 * 16 + r) * 64 + lane];
                const float vE = __shfl(tot, (lane & 32) + min(E, 31));
                const int sr = 32 * st + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (sr < NS && j < E) {
                    const float wv = wtT[j * L.tss + sr];
                    float g = 2.f * wv * tot - vE;
                    if (!d.log_scale) g *= sqrtf(2.f * wv);                 // / scale
                    d.pscl[(((int64_t)blockIdx.x * gy + ys) * NS + sr) * E + j] = g;
                }
            }
            __syncthreads();
        }
    }

    // ---- d value / d small: the four waves' sums through LDS; wave w finishes registers 4 w .. 4 w + 3
    __syncthreads();
    if (SMALL_ONLY) {
        // dvacc[r] = sum over this wave's (l, s tiles) of X[k_r(h), s = lane]: add up the 32 lanes of each half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = dvacc[r];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) x += __shfl_xor(x, o);
            if (j == 0) scr[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = x;
        }
        __syncthreads();
        if (FLAT) {
            if (tid < 32 && row0 + tid < rows_total)
                d.dsm[(int64_t)ys * rows_total + row0 + tid] = scr[tid] + scr[32 + tid] + scr[64 + tid] + scr[96 + tid];
        } else if (tid < 32 && 32 * kt + tid < NK) {
            d.dsm[((int64_t)ys * d.M + m) * NK + 32 * kt + tid] = scr[tid] + scr[32 + tid] + scr[64 + tid] + scr[96 + tid];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[(wave * 16 + r) * 64 + lane] = dvacc[r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * wave + rr;
        const float tot = scr[(0 * 16 + r) * 64 + lane] + scr[(1 * 16 + r) * 64 + lane] +
                          scr[(2 * 16 + r) * 64 + lane] + scr[(3 * 16 + r) * 64 + lane];
        const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (FLAT ? row0 + k < rows_total : k < NK) {
            const int64_t row = FLAT ? (int64_t)ys * rows_total + row0 + k : ((int64_t)ys * d.M + m) * NK + k;
            if (j < E && d.dval) d.dval[row * E + j] = -2.f * tot;
            if (j == E && d.dsm) d.dsm[row] = tot;
        }
    }
}

// out[c] = sum_p part[p][c]: a 16-column x 16-row-lane tile per workgroup (P rows of a few hundred, C of a few thousand)
struct ColSeg {
    const float *part;
    float *out;
    int32_t P, C;
};
__global__ __launch_bounds__(256) void nlb_colsum_kernel(ColSeg a, ColSeg b) {
    __shared__ float red[16][17];
    const ColSeg sg = blockIdx.y == 0 ? a : b;
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    if (blockIdx.x * 16 >= sg.C) return;
    float acc = 0.f;
    if (c < sg.C) {
        int p = ry;
        for (; p + 16 * 7 < sg.P; p += 16 * 8) {               // eight loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sg.part[(int64_t)(p + 16 * u) * sg.C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < sg.P; p += 16) acc += sg.part[(int64_t)p * sg.C + c];
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < sg.C) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) t += red[y][cx];
        sg.out[c] = t;
    }
}

// out[c] = sum_{y < P} part[y][c] for a handful of rows and many columns (the loc-row shares of d value / d small)
__global__ __launch_bounds__(256) void nlb_addrows_kernel(ColSeg a, ColSeg b) {
    const ColSeg sg = blockIdx.y == 0 ? a : b;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < sg.C; c += (int64_t)gridDim.x * 256) {
        float t = 0.f;
        for (int y = 0; y < sg.P; ++y) t += sg.part[(int64_t)y * sg.C + c];
        sg.out[c] = t;
    }
}

}  // namespace alan

using namespace alan;

namespace {

struct NLBPlan {
    int eh = 0, nkt = 1, nst = 1, gy = 1;
    int64_t tiles = 0;           // workgroups along x: M * nkt, or ceil(M NK / 32) with flat row tiling
    bool small_only = false, flat = false;
    size_t lds = 0;
    size_t o_ploc = 0, o_pscl = 0, o_dvp = 0, o_dsp = 0, bytes = 0;   // workspace offsets (bytes)
    int64_t rows_loc = 0, rows_scl = 0;
};

int plan_nlb(const alan_normal_lse_backward_desc_t &b, NLBPlan &p) {
    const alan_normal_lse_desc_t &a = b.fwd;
    if (!a.value || !a.loc || !a.scale || !b.lse || !b.grad_out) return ALAN_ERR_BAD_DESC;
    if (a.M < 1 || a.NK < 1 || a.NL < 1 || a.NS < 1 || a.E < 1) return ALAN_ERR_BAD_DESC;
    if (a.n_small < 0 || a.n_small > 4) return ALAN_ERR_BAD_DESC;
    for (int f = 0; f < a.n_small; ++f)
        if (!a.small[f]) return ALAN_ERR_BAD_DESC;
    if (a.E > 31 || a.NK > 4096 || a.NS > 4096 || a.NL > (1 << 20) || a.M > (1 << 22)) return ALAN_ERR_UNSUPPORTED;
    const int need = (int)(a.E + 2) / 2;                     // MFMA steps: the event dim plus the small-factor slot
    p.eh = need <= 4 ? 4 : need <= 8 ? 8 : need <= 10 ? 10 : need <= 12 ? 12 : 16;
    p.nkt = (int)((a.NK + 31) / 32);
    p.nst = (int)((a.NS + 31) / 32);
    p.small_only = !b.grad_value && !b.grad_loc && !b.grad_scale;
    const NLBLds L = nlb_lds(p.eh, (int)a.NS, (int)a.NL, (int)a.E);
    p.lds = (size_t)L.total * sizeof(float);
    if (p.lds > 150 * 1024) return ALAN_ERR_UNSUPPORTED;
    // flat row tiling (see the kernel): the plate elements' k rows as one run
    // (not for SMALL_ONLY: with no V and no U a tile is short, and the second log-sum-exp value per lane and the
    // selects cost more than the saved tiles -- K = 100: 334 against 315 us)
    p.flat = a.NK > 32 && (a.NK & 31) != 0 && (a.NK & 3) == 0 && a.v_sm == a.NK * a.v_sk &&
             (b.grad_value || b.grad_loc || b.grad_scale);
    // workgroups: one per (plate element, k tile), times gy shares of the loc rows so that the chip is covered
    const int64_t base = p.flat ? (a.M * a.NK + 31) / 32 : a.M * p.nkt;
    p.tiles = base;
    if (base * 64 >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    // (one share unless the plate is short: every share adds a row of partials to the column sums and the loc-share
    // add launch -- at K=30, M=300 one share is 39 + 4 us, three are 37 + 10 + 4)
    int64_t gy = std::max<int64_t>(1, (256 + base - 1) / base);
    gy = std::min<int64_t>(gy, (a.NL + 3) / 4);
    p.gy = (int)std::min<int64_t>(gy, 65535);
    p.rows_loc = base * p.nst;
    p.rows_scl = base * p.gy;
    size_t off = 0;
    auto take = [&](size_t floats) {
        const size_t o = off;
        off += (floats * sizeof(float) + 255) & ~(size_t)255;
        return o;
    };
    if (!p.small_only) {
        p.o_ploc = take(b.grad_loc ? (size_t)p.rows_loc * a.NL * a.E : 0);
        p.o_pscl = take(b.grad_scale ? (size_t)p.rows_scl * a.NS * a.E : 0);
    }
    if (p.gy > 1) {
        p.o_dvp = take(b.grad_value ? (size_t)p.gy * a.M * a.NK * a.E : 0);
        p.o_dsp = take(b.grad_small ? (size_t)p.gy * a.M * a.NK : 0);
    }
    p.bytes = std::max<size_t>(off, 256);
    return ALAN_OK;
}

}  // namespace

extern "C" size_t alan_normal_lse_backward_workspace_bytes(const alan_normal_lse_backward_desc_t *b) {
    if (!b) return 0;
    NLBPlan p;
    if (plan_nlb(*b, p) != ALAN_OK) return 0;
    return p.bytes;
}

extern "C" int alan_normal_lse_backward(const alan_normal_lse_backward_desc_t *b, void *workspace,
                                        size_t workspace_bytes, void *stream_) {
    if (!b) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    NLBPlan p;
    int rc = plan_nlb(*b, p);
    if (rc != ALAN_OK) return rc;
    if (!workspace || workspace_bytes < p.bytes) return ALAN_ERR_WORKSPACE;
    const alan_normal_lse_desc_t &a = b->fwd;
    if (p.small_only && !b->grad_small) return ALAN_OK;                       // nothing wanted
    static const int x2_knob = env_knob("ALAN_NLB_X2");                               // ablation knob: 0 = fp32 V and U products
    const bool x2 = x2_knob != 0;
    char *ws = (char *)workspace;
    NLBDesc d;
    std::memset(&d, 0, sizeof(d));
    d.val = (const float *)a.value, d.loc = (const float *)a.loc, d.scl = (const float *)a.scale;
    d.lse = (const float *)b->lse, d.gout = (const float *)b->grad_out;
    d.M = (int)a.M, d.NK = (int)a.NK, d.NL = (int)a.NL, d.NS = (int)a.NS, d.E = (int)a.E;
    d.n_small = a.n_small, d.log_scale = a.log_scale;
    d.v_sm = a.v_sm, d.v_sk = a.v_sk, d.v_se = a.v_se;
    d.l_sl = a.l_sl, d.l_se = a.l_se, d.s_ss = a.s_ss, d.s_se = a.s_se;
    d.g_sl = b->g_sl, d.g_ss = b->g_ss;
    for (int f = 0; f < 4; ++f) {
        const bool used = f < a.n_small;
        d.small[f] = used ? (const float *)a.small[f] : (const float *)a.value;   // (unused: any valid address)
        d.small_sm[f] = used ? a.small_sm[f] : 0;
        d.small_sk[f] = used ? a.small_sk[f] : 0;
    }
    const bool split = p.gy > 1;
    d.dval = b->grad_value ? (split ? (float *)(ws + p.o_dvp) : (float *)b->grad_value) : nullptr;
    d.dsm = b->grad_small ? (split ? (float *)(ws + p.o_dsp) : (float *)b->grad_small) : nullptr;
    d.ploc = (!p.small_only && b->grad_loc) ? (float *)(ws + p.o_ploc) : nullptr;
    d.pscl = (!p.small_only && b->grad_scale) ? (float *)(ws + p.o_pscl) : nullptr;

    const dim3 grid((uint32_t)p.tiles, (uint32_t)p.gy);
    auto launch = [&](auto kern) {
        if (p.lds > 64 * 1024)
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) !=
                hipSuccess)
                return ALAN_ERR_LAUNCH;
        ALAN_LAUNCH_EXT(kern, grid, dim3(256), p.lds, stream, (hipEvent_t)a.ev_start, (hipEvent_t)a.ev_stop, 0, d);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
#define NLB_CASE(EHV)                                                                        \
    case EHV:                                                                                \
        rc = p.flat ? (p.small_only ? launch(normal_lse_bwd_kernel<EHV, true, true>)                                   \
                                    : x2 ? launch(normal_lse_bwd_kernel<EHV, false, true, true>)                       \
                                         : launch(normal_lse_bwd_kernel<EHV, false, true>))                            \
                    : (p.small_only ? launch(normal_lse_bwd_kernel<EHV, true>)                                         \
                                    : x2 ? launch(normal_lse_bwd_kernel<EHV, false, false, true>)                      \
                                         : launch(normal_lse_bwd_kernel<EHV, false>));                                 \
        break;
    switch (p.eh) {
        NLB_CASE(4)
        NLB_CASE(8)
        NLB_CASE(10)
        NLB_CASE(12)
        default:
            NLB_CASE(16)
    }
#undef NLB_CASE
    if (rc != ALAN_OK) return rc;

    // ---- second stage: the per-workgroup partials of d loc / d scale, and the loc-row shares of d value / d small
    ColSeg s0{nullptr, nullptr, 0, 0}, s1{nullptr, nullptr, 0, 0};
    if (d.ploc) s0 = ColSeg{d.ploc, (float *)b->grad_loc, (int32_t)p.rows_loc, (int32_t)(a.NL * a.E)};
    if (d.pscl) s1 = ColSeg{d.pscl, (float *)b->grad_scale, (int32_t)p.rows_scl, (int32_t)(a.NS * a.E)};
    if (s0.C || s1.C) {
        const uint32_t gx = (uint32_t)((std::max(s0.C, s1.C) + 15) / 16);
        ALAN_LAUNCH(nlb_colsum_kernel, dim3(gx, 2), dim3(256), 0, stream, s0, s1);
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    }
    if (split && (d.dval || d.dsm)) {
        ColSeg v0{nullptr, nullptr, 0, 0}, v1{nullptr, nullptr, 0, 0};
        const int64_t cv = a.M * a.NK * a.E, cs = a.M * a.NK;
        if (cv >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
        if (d.dval) v0 = ColSeg{d.dval, (float *)b->grad_value, p.gy, (int32_t)cv};
        if (d.dsm) v1 = ColSeg{d.dsm, (float *)b->grad_small, p.gy, (int32_t)cs};
        const uint32_t gx = (uint32_t)std::min<int64_t>(2048, (std::max<int64_t>(v0.C, v1.C) + 255) / 256);
        ALAN_LAUNCH(nlb_addrows_kernel, dim3(gx, 2), dim3(256), 0, stream, v0, v1);
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    }
    return ALAN_OK;
}
