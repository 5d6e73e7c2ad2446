// The fused plate step with the square EXPANDED (round 4): the kernel of alan_normal_lse for one tile of child particles
// and one tile of scale rows (NK <= 32, NS <= 32: movielens at K <= 32) and event lengths >= 8.  gfx950 only.
//
//   out[l, s] = sum_m LSE_k( log N(value[m,k,:]; loc[l,:], scale[s,:]) + sum_f small_f[m,k] )
//   (TorchDimDist.py:127-162 + utils.py:147-152 + reduce_Ks.py:249-251 + utils.py:218-220 + logpq.py:149)
//
// normal_lse_x3.h builds, for EVERY (plate element m, loc row l), the operand A = split3((v - mu_l)^2): 11 vector
// instructions per element, 110 of the ~230 a 32 x 32 tile costs at K = 30, where an A operand serves ONE tile -- the
// kernel is bound by the vector unit, the matrix pipe is 29 % busy (profiles/r3_fused_forward_sq_counters_K30.md).  Here,
// with v' = v - c and mu' = mu_l - c (c = the workgroup's first loc row: a common shift changes nothing),
//     D[k, (l, s)] = sum_e v'^2 w[s]  -  2 sum_e v' (mu'_l w[s])  +  sum_e mu'_l^2 w[s],         w = log2(e) / (2 sigma^2)
//   * T1[k, s]  = sum_e v'^2 w[s] + small[k]     one chain of matrix instructions per (m, workgroup), operand split3(v'^2);
//   * the cross term: A2 = split3(v') built ONCE per plate element and shared by every loc row of the workgroup; its B
//     operand split3(-2 mu'_l w[s]) depends on (l, s) only: a table the workgroup builds once into LDS, in the lanes' own
//     operand layout (one ds_read_b128 per matrix instruction); the chain starts from C = T1;
//   * the last term does not depend on k: it leaves the log-sum-exp and is added to the per-(l, s) constant with the
//     log-normaliser.
// Per (m, l) tile that leaves the log-sum-exp (~100 vector instructions) beside 7-8 matrix instructions.
//
// ACCURACY.  The three terms are each as accurate as fp32 (3-way split operands, exact bf16 products, fp32 accumulate) but
// their SUM cancels when v' and mu' are both large against |v - mu|: the absolute error of D is ~2^-23 (T1 + cross + const)
// where the difference form's is ~2^-23 D.  With the shift c the operands are the SPREAD of the loc rows (and the values'
// distance from them) in units of sigma: error 2^-23 (spread / sigma)^2 per event.  For the long event dims this kernel
// takes (E >= 8) a value row within sigma of a loc row in every event while the rows are spread over many sigma does not
// happen by chance, and D is large wherever its terms are (relative error stays ~1e-6: tests/test_gpu_fused_plate_step.py
// at rtol 3e-5 incl. adversarial shapes; measured 5e-6 worst on movielens at initialisation, 3e-8 on a concentrated
// posterior).  Short event dims (bus_breakdown's scalars), where such matches are common, stay on the difference form, and
// ALAN_NLSE_EXPANDED=0 / alan_normal_lse_desc_t.exact_difference keeps everything there.
#pragma once
#include "normal_lse_x3.h"

namespace alan {

struct XEDesc {
    const float *val, *loc, *scl;
    float *part, *lse;
    const float *small[4];
    int32_t M, NK, NL, NS, E, n_small, log_scale;
    int32_t v_sm, l_sl, l_se, s_ss, s_se;
    int32_t small_sm[4], small_sk[4];
    int32_t LG, n_mg;                 // loc rows per workgroup; groups of the plate (gridDim.y)
    uint32_t rcp_e;                   // ceil(2^16 / E)
};

// LDS, in floats (xe_lds_floats): cross table | T1's B operand | staged w' | staged log sigma | loc rows | constants |
// per wave: value tile [32][ES], accumulators [LG][32]
template <int EQ, int EQC, int NW>
struct XELayout {
    static constexpr int NSTEP = (3 * EQ + 3) / 4, NSTEPC = (3 * EQC + 3) / 4;
    static constexpr int ES = 33;                     // staged rows: odd stride
    __host__ __device__ static constexpr int btab() { return 0; }
    __host__ __device__ static constexpr int wtab(int LG) { return LG * NSTEPC * 256; }
    __host__ __device__ static constexpr int wst(int LG) { return wtab(LG) + NSTEP * 256; }
    __host__ __device__ static constexpr int lst(int LG) { return wst(LG) + 32 * ES; }
    __host__ __device__ static constexpr int locs(int LG) { return lst(LG) + 32 * ES; }
    __host__ __device__ static constexpr int cst(int LG) { return locs(LG) + LG * 36; }
    __host__ __device__ static constexpr int waves(int LG) { return cst(LG) + LG * 32; }
    __host__ __device__ static constexpr int per_wave(int LG) { return 32 * ES + LG * 32; }
    __host__ __device__ static constexpr int total(int LG) { return waves(LG) + NW * per_wave(LG); }
};

// EQ: events per lane half of T1's operand incl. the small-factor slot (2 EQ >= E + 1); EQC: of the cross term's (2 EQC
// >= E); NW: waves per workgroup.  Grid: x = group of LG loc rows, y = group of the plate; the workgroup's waves take the
// group's plate elements in turn (m0 + wave, m0 + wave + NW, ...), one 32-row tile each.
template <int EQ, int EQC, int NW>
__global__ __launch_bounds__(64 * NW, 2) void normal_lse_xe_kernel(const XEDesc d) {
    typedef XELayout<EQ, EQC, NW> LY;
    constexpr int NT = 64 * NW;
    constexpr int NSTEP = LY::NSTEP, NSTEPC = LY::NSTEPC, ES = LY::ES;
    constexpr int NV = 4 * NSTEP, NVC = 4 * NSTEPC;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS, LG = d.LG;
    const int l0 = blockIdx.x * LG, LGc = min(LG, d.NL - l0);
    const int m0 = (int)((int64_t)blockIdx.y * d.M / d.n_mg), m1 = (int)((int64_t)(blockIdx.y + 1) * d.M / d.n_mg);
    const int slot_h = E > 2 * (EQ - 1) ? 1 : 0;                    // the small-factor slot: event pair EQ - 1, this half
    const float inf = __builtin_huge_valf();
#ifdef ALAN_TIMELINE
    unsigned long long tl[NL_TL_SLOTS] = {};
    const unsigned long long tl_real = __builtin_amdgcn_s_memrealtime();
    int tl_units = 0;
    NL_STAMP(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(NK), "s"(E), "s"(NS) : "memory");       // (the kernel arguments have arrived)
    NL_STAMP(13);
#endif
    u32x4v *btab = reinterpret_cast<u32x4v *>(lds + LY::btab());
    u32x4v *wtab = reinterpret_cast<u32x4v *>(lds + LY::wtab(LG));
    float *wst = lds + LY::wst(LG), *lst = lds + LY::lst(LG), *locs = lds + LY::locs(LG), *cst = lds + LY::cst(LG);
    float *tile = lds + LY::waves(LG) + wave * LY::per_wave(LG);
    float *accw = tile + 32 * ES;                                   // [LG][32] this wave's plate sums
    // ---- the wave's first value tile and its small factors: requested before anything waits
    constexpr int NX = EQ;
    uint32_t soff[NX];
#pragma unroll
    for (int qq = 0; qq < NX; ++qq) {
        const uint32_t f = lane + 64 * qq, row = (f * d.rcp_e) >> 16;
        soff[qq] = row < 32 ? f + row * (uint32_t)(ES - E) : 32 * ES - 1;               // (beyond: a slot nobody reads)
    }
    const uint32_t lane4 = lane * 4;
    auto load_run = [&](int m, float (&x)[NX]) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void *)(d.val + (int64_t)m * d.v_sm), 0, NK * E * 4, 0x00020000);
#pragma unroll
        for (int qq = 0; qq < NX; ++qq)
            x[qq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + 256 * qq, 0, 0));
    };
    auto load_small = [&](int m, float (&hs)[4]) {
        const uint32_t kk = (uint32_t)min(j, NK - 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {                 // (the launcher points unused slots at valid memory, stride 0)
            const float *sp = d.small[f] + (int64_t)m * d.small_sm[f];                       // (scalar)
            hs[f] = sp[kk * (uint32_t)d.small_sk[f]];
        }
        asm volatile("" ::: "memory");
    };
    float zc[NX], zn[NX], hc[4], hn[4];
    int m = m0 + wave;
    if (m < m1) {
        load_run(m, zc);
        load_small(m, hc);
    }
    // ---- phase 1: w' = log2(e) / (2 sigma^2) and log sigma of the scale rows, the group's loc rows -> LDS
    {
        const bool lsc = d.log_scale != 0;
        constexpr int R1 = (32 * 2 * EQ + NT - 1) / NT, R2 = (512 + NT - 1) / NT;     // rounds over the (s, e) / (l, e) elements
        float xs[R1], xl[R2];
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            const uint32_t idx = tid + NT * r, s = (idx * d.rcp_e) >> 16, e = idx - s * E;
            xs[r] = d.scl[(uint32_t)(min((int)s, NS - 1) * d.s_ss + (int)e * d.s_se)];
        }
#pragma unroll
        for (int r = 0; r < R2; ++r) {                // (LG E <= 512)
            const uint32_t idx = tid + NT * r, l = (idx * d.rcp_e) >> 16, e = idx - l * E;
            xl[r] = d.loc[(uint32_t)(min(l0 + (int)l, d.NL - 1) * d.l_sl + (int)e * d.l_se)];
        }
#ifdef ALAN_TIMELINE
        NL_STAMP(14);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NL_STAMP(15);
#endif
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            const uint32_t idx = tid + NT * r, s = (idx * d.rcp_e) >> 16, e = idx - s * E;
            if (s < 32) {
                const float x = xs[r];
                const bool ok = (int)s < NS;
                wst[s * ES + e] = !ok ? 0.f
                                  : lsc ? __builtin_amdgcn_exp2f(-2.f * NL_LOG2E * x) * (0.5f * NL_LOG2E)
                                        : (0.5f * NL_LOG2E) * __builtin_amdgcn_rcpf(x * x);
                lst[s * ES + e] = !ok ? 0.f : lsc ? x : __builtin_amdgcn_logf(x) * NL_LN2;
            }
        }
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const uint32_t idx = tid + NT * r, l = (idx * d.rcp_e) >> 16, e = idx - l * E;
            if ((int)l < LG) locs[l * 36 + e] = xl[r];
        }
        for (int i = lane; i < LG * 32; i += 64) accw[i] = 0.f;
    }
    __syncthreads();
    NL_STAMP(1);                                      // (staged operands visible: the first barrier passed)
    // ---- phase 2: the tables, in the lanes' own operand layout.  Lane (j = scale row, h) holds the events 2 q + h:
    // registers 3 q .. 3 q + 2 of its operand, dword (3 q + i) & 3 of step (3 q + i) >> 2.  Items = the cross tables of
    // the group's loc rows, then T1's operand (split3(w') with log2(e) in the small-factor slot), one per wave in turn
    // (with eight waves and LG <= 7 no wave builds two), the event pairs unrolled: a wave needs that many independent
    // chains in flight -- the same items walked event pair by event pair in a rolled loop took 660 cycles apiece.
    {
        float wv[EQ];
#pragma unroll
        for (int q = 0; q < EQ; ++q) wv[q] = 2 * q + h < E ? wst[j * ES + 2 * q + h] : 0.f;
        for (int it = wave; it < LGc + 1; it += NW) {                         // (scalar)
            if (it == LGc) {
                unsigned r[NV];
#pragma unroll
                for (int q = 0; q < EQ; ++q) {
                    const float bval = 2 * q + h < E ? wv[q] : (j < NS && q == EQ - 1 && h == slot_h) ? NL_LOG2E : 0.f;
                    nl_split_b(bval, r[3 * q], r[3 * q + 1], r[3 * q + 2]);
                }
#pragma unroll
                for (int v = 3 * EQ; v < NV; ++v) r[v] = 0u;
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) wtab[st * 64 + lane] = u32x4v{r[4 * st], r[4 * st + 1], r[4 * st + 2], r[4 * st + 3]};
                continue;
            }
            const int l = it;
            unsigned r[NVC];
            float cc = 0.f, lg = 0.f;
#pragma unroll
            for (int q = 0; q < EQC; ++q) {
                const bool ev = 2 * q + h < E;
                const float mp = ev ? locs[l * 36 + 2 * q + h] - locs[2 * q + h] : 0.f;
                const float mw = mp * wv[q];
                cc = fmaf(mp, mw, cc);
                lg += ev ? lst[j * ES + 2 * q + h] : 0.f;
                nl_split_b(-2.f * mw, r[3 * q], r[3 * q + 1], r[3 * q + 2]);
            }
#pragma unroll
            for (int v = 3 * EQC; v < NVC; ++v) r[v] = 0u;
#pragma unroll
            for (int st = 0; st < NSTEPC; ++st)
                btab[(l * NSTEPC + st) * 64 + lane] = u32x4v{r[4 * st], r[4 * st + 1], r[4 * st + 2], r[4 * st + 3]};
            // the k-free terms: sum_e mu'^2 w' (base 2 -> natural units) + the log-normaliser sum_e log sigma + E log sqrt(2 pi)
            // -- what is subtracted from (log2(sum + eps) - min) ln 2; both halves add their events, then meet
            const float part = cc * NL_LN2 + lg;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
            if (h == 0) cst[l * 32 + j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]) + (float)E * 0.91893853320467274178f;
        }
    }
    // the centre c = the group's first loc row, this lane's events
    float cen[EQ];
#pragma unroll
    for (int q = 0; q < EQ; ++q) cen[q] = locs[min(2 * q + h, E - 1)];
    NL_STAMP(2);                                      // (tables written)
    __syncthreads();
    NL_STAMP(3);
    const float n_small_mask[4] = {d.n_small > 0 ? 1.f : 0.f, d.n_small > 1 ? 1.f : 0.f, d.n_small > 2 ? 1.f : 0.f,
                                   d.n_small > 3 ? 1.f : 0.f};
    for (; m < m1; m += NW) {
        const int mn_ = m + NW;
        if (mn_ < m1) {
            load_run(mn_, zn);
            load_small(mn_, hn);
        }
        float hsum = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) hsum += n_small_mask[f] != 0.f ? hc[f] : 0.f;
        // the run of NK rows x E floats into the wave's tile, each lane its row's events back
#pragma unroll
        for (int qq = 0; qq < NX; ++qq) tile[soff[qq]] = zc[qq];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float zv[EQ];
#pragma unroll
        for (int q = 0; q < EQ; ++q) zv[q] = tile[j * ES + min(2 * q + h, E - 1)] - cen[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float nh = -hsum;
        const float slot = j < NK ? (nh > NL_BIG ? NL_BIG : nh) : NL_BIG;      // (a NaN small factor stays a NaN)
        // ---- T1 = sum_e v'^2 w + small: its A operand, then the chain against the table of split3(w')
        f32x16 t1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            unsigned a1[NV];
#pragma unroll
            for (int q = 0; q < EQ; ++q) {
                float a = zv[q] * zv[q];
                if (q == EQ - 1) a = h == slot_h ? slot : a;
                nl_split_a(a, a1[3 * q], a1[3 * q + 1], a1[3 * q + 2]);
            }
#pragma unroll
            for (int v = 3 * EQ; v < NV; ++v) a1[v] = 0u;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const u32x4v av = {a1[4 * st], a1[4 * st + 1], a1[4 * st + 2], a1[4 * st + 3]};
                const u32x4v bv = wtab[st * 64 + lane];
                t1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, av), __builtin_bit_cast(bf16x8v, bv), t1, 0, 0, 0);
                if (st == 0) asm volatile("" ::"v"(t1[0]), "v"(av), "v"(bv));       // (see normal_lse_x3.h: the chain's first destination)
            }
        }
        // ---- the cross term's A operand: split3(v'), shared by every loc row of the group (pad events meet zeros in B)
        unsigned a2[NVC];
#pragma unroll
        for (int q = 0; q < EQC; ++q) nl_split_a(zv[q], a2[3 * q], a2[3 * q + 1], a2[3 * q + 2]);
#pragma unroll
        for (int v = 3 * EQC; v < NVC; ++v) a2[v] = 0u;
#ifdef ALAN_TIMELINE
        if (m == m0 + wave) NL_STAMP(4);              // (xe: the first plate element's operands built, T1 issued)
#endif
        auto chain = [&](int l) {
            f32x16 acc = t1;
#pragma unroll
            for (int st = 0; st < NSTEPC; ++st) {
                const u32x4v av = {a2[4 * st], a2[4 * st + 1], a2[4 * st + 2], a2[4 * st + 3]};
                const u32x4v bv = btab[(l * NSTEPC + st) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, av), __builtin_bit_cast(bf16x8v, bv), acc, 0, 0, 0);
            }
            return acc;
        };
        // acc[r] = -log2(e) (log-prob + small) of row (r & 3) + 8 (r >> 2) + 4 h, the k-free terms apart
        // Loc rows, software-pipelined: the matrix instructions of row l + 1 are issued in the same basic block as the
        // log-sum-exp of row l, interleaved (one matrix instruction, the LDS read that feeds the next, a share of the vector
        // work): an in-order wave that issues its dependent chain back to back stalls through all of it.
        constexpr int VPG = (64 + NSTEPC - 1) / NSTEPC;
        f32x16 nxt = chain(0);
        for (int l = 0; l < LGc; ++l) {
            const f32x16 cur = nxt;
            nxt = chain(min(l + 1, LGc - 1));             // (the last row once more: no branch in the block)
            float tmin = cur[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmin = fminf(tmin, cur[r]);
            const f32x2v mf2 = {tmin, tmin};
            f32x2v part = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2v c2 = {cur[r], cur[r + 1]};
                const f32x2v d2 = mf2 - c2;
                const f32x2v e2 = {__builtin_amdgcn_exp2f(d2[0]), __builtin_amdgcn_exp2f(d2[1])};
                part += e2;
            }
            const float ssum = part[0] + part[1];
            // join the two half-waves (rows 4 h .. of every group of eight)
            const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmin), __float_as_uint(tmin), false, false);
            const auto ps = __builtin_amdgcn_permlane32_swap(__float_as_uint(ssum), __float_as_uint(ssum), false, false);
            const float mn1 = __uint_as_float(pm[0]), mn2 = __uint_as_float(pm[1]);
            const float sm1 = __uint_as_float(ps[0]), sm2 = __uint_as_float(ps[1]);
            const float mm = fminf(mn1, mn2);
            const float tot = sm1 * __builtin_amdgcn_exp2f(mm - mn1) + sm2 * __builtin_amdgcn_exp2f(mm - mn2);
            // log(tot + eps) + max, in base 2 until the end (utils.py:218-220); every row masked / a +inf term: NaN
            float lse_m = (__builtin_amdgcn_logf(tot + Num<float>::eps) - mm) * NL_LN2 - cst[l * 32 + j];
            if (mm >= 1e29f || mm == -inf) lse_m = __builtin_nanf("");
            accw[l * 32 + j] += lse_m;                    // (both halves hold it: two lanes write the same value)
#pragma unroll
            for (int i = 0; i < NSTEPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
            }
            if (d.lse && h == 0 && j < NS) d.lse[((int64_t)m * d.NL + l0 + l) * NS + j] = lse_m;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) zc[i] = zn[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) hc[f] = hn[f];
#ifdef ALAN_TIMELINE
        if (m == m0 + wave) NL_STAMP(5);
        tl_units += LGc;
#endif
    }
    NL_STAMP(6);
    // ---- the four waves' plate sums, added in wave order
    __syncthreads();
    const float *a0 = lds + LY::waves(LG) + 32 * ES;
    for (int i = tid; i < LGc * 32; i += NT) {
        const int l = i >> 5, sr = i & 31;
        const int pw = LY::per_wave(LG);
        float tot = a0[i];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += a0[w * pw + i];
        if (sr < NS) d.part[((int64_t)blockIdx.y * d.NL + l0 + l) * NS + sr] = tot;
    }
#ifdef ALAN_TIMELINE
    NL_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NL_STAMP(8);
    const int wid = (blockIdx.y * gridDim.x + blockIdx.x) * NW + wave;
    if (lane == 0 && wid < NL_TL_WAVES) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl[9] = tl_real, tl[10] = ((unsigned long long)xcc << 32) | hwid, tl[11] = (unsigned long long)tl_units;
        tl[12] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < NL_TL_SLOTS; ++i) nl_timeline[wid * NL_TL_SLOTS + i] = tl[i];
    }
#endif
}

}  // namespace alan
