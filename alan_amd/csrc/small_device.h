// Device code of the small-problem kernels (reduce.hip).  gfx950 only.
#pragma once
#include <cstddef>

#include "common.h"
#include "plan.h"
#include "normal_lse_table.h"

namespace alan {

__device__ __forceinline__ void small_store(float *p, float v) { *p = v; }

// Diagnostic build (make TIMELINE=1; tools/small_timeline.py): every workgroup of a multi-problem launch stamps s_memtime at
// the phases of its life -- 0 entry, 1 its problem found and the descriptor's head read, 2 the first round of loads landed,
// 3 the walk over the reduced dim done, 4 lanes combined and the result stored, 5 exit -- and leaves them, with
// s_memrealtime of entry and exit, its problem and where it ran, in a buffer of the library's own ("show
// where a 10 us kernel with < 1 MB of input spends its life").  In the default build no stamp executes.
#ifdef ALAN_TIMELINE
constexpr int SM_TL_SLOTS = 12, SM_TL_WGS = 4096;
__device__ unsigned long long sm_timeline[SM_TL_WGS * SM_TL_SLOTS];
__shared__ unsigned long long sm_tl[SM_TL_SLOTS];
#define SM_STAMP(i) do { if (threadIdx.x == 0) sm_tl[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define SM_STAMP_LANDED(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SM_STAMP(i); } while (0)
#else
#define SM_STAMP(i) ((void)0)
#define SM_STAMP_LANDED(i) ((void)0)
#endif

// ------------------------------------------------------------------------------------------
// Generated noise (alan_noise_t): Philox4x32-10 (Salmon et al., SC'11: the counter-based generator torch's own CUDA /
// HIP sampling kernels use) keyed by the seed, counter = the element's index / 4; its four 32-bit words make four
// standard normals by Box-Muller (u in (0, 1]: x 2^-32 + 2^-33, as curand's uniform), element i takes the (i & 3)th.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (one 32 x 32 -> 64 multiply per product: v_mad_u64_u32, quarter rate like each of a mul_hi / mul_lo pair)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0, c1 = lo1, c2 = hi0 ^ c3 ^ k1, c3 = lo0;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    r[0] = c0, r[1] = c1, r[2] = c2, r[3] = c3;
}

__device__ __forceinline__ float noise_at(const uint64_t seed, const uint64_t i) {
    uint32_t r[4];
    const uint64_t c = i >> 2;
    philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 0x414c414eu, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int lane = (int)(i & 3u);
    const uint32_t a = lane & 2 ? r[2] : r[0], b = lane & 2 ? r[3] : r[1];
    const float u1 = (float)a * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
    const float u2 = (float)b * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // sqrt(-2 ln u1)
    // (v_sin_f32 / v_cos_f32 take their argument in turns)
    return rad * (lane & 1 ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
}

// Behind a launch's problems: its first workgroup leaves the receipt and hands the generator on -- {counter + advance_by,
// seed} written to the slot whose ADDRESS the word `advance` holds: never the slot this launch reads (its other
// workgroups may not have read it yet), so nothing has to wait for anything.
__device__ __forceinline__ void noise_finish(const NoiseLaunch &n, const uint64_t cell_value, const uint64_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n.receipt) {
        __hip_atomic_store(n.receipt, cell_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(n.receipt + 1, seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (n.advance) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(
            __hip_atomic_load(n.advance, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store(dst, cell_value + n.advance_by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------
// One element of the reduce index: the mode's term from the loaded factor values (shared by both kernels).
template <typename T, int MODE>
__device__ __forceinline__ void accumulate(T &m, T &s, const T (&val)[MAXF], T wv, const float (&scale)[MAXF],
                                           int nf, bool ok) {
    if (MODE == ALAN_MODE_NORMAL || MODE == ALAN_MODE_NORMAL_LOGSCALE) {
        // torch.distributions.Normal.log_prob: -(v-loc)^2/(2 var) - log(scale) - log(sqrt(2 pi));
        // one term per (value, loc, scale) triple, weighted by the value factor's scale field
        T lp = T(0);
#pragma unroll
        for (int tm = 0; tm < MAXF / 3; ++tm) {
            if (3 * tm < nf) {
                const T z = val[3 * tm] - (T)scale[3 * tm + 1] * val[3 * tm + 1];      // (loc's scale field: loc = c * raw)
                const T sc = val[3 * tm + 2];
                const bool logsc = MODE == ALAN_MODE_NORMAL_LOGSCALE || scale[3 * tm + 2] == 2.f;
                const T one = logsc
                    ? -(z * z) * (T(0.5) * Num<T>::exp_acc(T(-2) * sc)) - sc - T(0.91893853320467274178)
                    : -(z * z) / (T(2) * sc * sc) - Num<T>::log(sc) - T(0.91893853320467274178);
                lp += (T)scale[3 * tm] * one;
            }
        }
        s += ok ? lp : T(0);
    } else if (MODE == ALAN_MODE_PRODUCER_GRAD) {
        // G * d log-prob / d (one argument); which argument is a launch-uniform switch (factor 0's scale field)
        const int kind = (int)scale[0];
        const T g = val[0];
        T r;
        if (kind == 4) {
            const T xl = val[2];
            const T sg = T(1) / (T(1) + Num<T>::exp_acc(-xl));             // d/dx [logsigmoid(x) - (1 - y) x] = y - sigmoid(x)
            r = g * (val[1] - sg);
        } else {
            const T z = val[1] - val[2], sc = val[3];
            const bool logsc = scale[3] == 2.f;
            const T w = logsc ? Num<T>::exp_acc(T(-2) * sc) : T(1) / (sc * sc);  // 1 / scale^2
            if (kind == 1)
                r = -g * z * w;
            else if (kind == 2)
                r = g * z * w;
            else
                r = logsc ? g * (z * z * w - T(1)) : g * (z * z * w - T(1)) / sc;
        }
        s += ok ? r : T(0);
    } else if (MODE == ALAN_MODE_BERNOULLI) {
        // torch.distributions.Bernoulli.log_prob = -BCE_with_logits = logsigmoid(x) - (1 - y) x,
        // logsigmoid(x) = min(x, 0) - log1p(exp(-|x|))
        const T y = val[0], xl = val[1];
        const T ls = (xl < T(0) ? xl : T(0)) - Num<T>::log1p(Num<T>::exp_acc(xl < T(0) ? xl : -xl));
        s += ok ? ls - (T(1) - y) * xl : T(0);
    } else if (MODE == ALAN_MODE_DOT) {
        // (a third factor multiplies in, through exp when its scale field says it holds a logarithm)
        const T third = nf > 2 ? (scale[2] == 2.f ? Num<T>::exp_acc(val[2]) : val[2]) : T(1);
        // (a fourth factor is ADDED, weighted by its scale field: the other contribution to the same gradient)
        // and a fifth joins the first: (f0 + f4) f1 g(f2) -- a gradient that arrives in two pieces
        const T lead = val[0] + (nf > 4 ? val[4] : T(0));
        s += ok ? lead * val[1] * third + ((nf > 3 && scale[3] != 0.f) ? (T)scale[3] * val[3] : T(0)) : T(0);
    } else if (MODE == ALAN_MODE_AFFINE) {
        const T sc = scale[2] == 2.f ? Num<T>::exp_acc(val[2]) : val[2];
        s += ok ? val[0] + val[1] * sc : T(0);
    } else {
        T x = T(0);
#pragma unroll
        for (int f = 0; f < MAXF; ++f)
            if (f < nf) x += (T)scale[f] * val[f];
        if (MODE == ALAN_MODE_LSE) {
            if (ok) lse_push(m, s, x);
        } else if (MODE == ALAN_MODE_SUM) {
            s += ok ? x : T(0);
        } else {
            s += ok ? wv * Num<T>::exp(x) : T(0);
        }
    }
}

// Lanes of a group (or the NW waves of a block) -> one value.  Log-sum-exp in two passes -- the maximum over the lanes first
// (compares only), every lane's sum rescaled to it ONCE, then a plain sum -- where merging (max, sum) pairs step by step took
// two exponentials per step and lane: the 1024-thread launch that ends an evaluation spent 3.5 of its 6.1 us there (six
// shuffle steps, then fifteen dependent merges by one thread; tools/small_timeline.py).
template <typename T>
__device__ __forceinline__ void lse_rescale(T &m, T &s, T M) {
    s = s * ((m == Num<T>::ninf()) ? T(0) : Num<T>::exp(m - M));      // (a NaN sum survives the multiply by 0, as it must)
    m = M;
}
template <typename T, int MODE, bool BLOCK, int NW = 4>
__device__ __forceinline__ void combine_lanes(T &m, T &s, uint32_t G) {
    const uint32_t WG = BLOCK ? 64u : G;  // lanes combined by shuffles
    if (MODE == ALAN_MODE_LSE) {
        T M = m;
        for (uint32_t ofs = WG >> 1; ofs > 0; ofs >>= 1) {
            const T m2 = __shfl_xor(M, (int)ofs);
            M = M > m2 ? M : m2;
        }
        lse_rescale(m, s, M);
    }
    for (uint32_t ofs = WG >> 1; ofs > 0; ofs >>= 1) s += __shfl_xor(s, (int)ofs);
    if (BLOCK) {
        __shared__ T sm[NW], ss[NW];
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (ln == 0) {
            sm[wv] = m;
            ss[wv] = s;
        }
        __syncthreads();
        if (wv == 0) {                                   // the waves' values on the first wave's lanes, the same two passes
            static_assert(NW <= 64 && (NW & (NW - 1)) == 0, "waves of a block: a power of two");
            m = ln < NW ? sm[ln] : Num<T>::ninf();
            s = ln < NW ? ss[ln] : T(0);
            if (MODE == ALAN_MODE_LSE) {
                T M = m;
#pragma unroll
                for (int ofs = NW >> 1; ofs > 0; ofs >>= 1) {
                    const T m2 = __shfl_xor(M, ofs);
                    M = M > m2 ? M : m2;
                }
                lse_rescale(m, s, M);
            }
#pragma unroll
            for (int ofs = NW >> 1; ofs > 0; ofs >>= 1) s += __shfl_xor(s, ofs);
        }
    }
}

// ------------------------------------------------------------------------------------------
// NW: waves of the workgroup (BLOCK: all of them on one output); UNR elements of the reduce index per thread and round, PF
// slices of a partial-sum factor (role PRESUM) per element and round -- UNR x PF loads in flight.
template <int MODE, bool BLOCK, int NW = 4, int UNR = 4, int PF = 8>
__device__ __forceinline__ void small_body(const SmallDesc &d, const int logG, const uint32_t block_id,
                                           const uint64_t noise_seed = 0, const uint64_t noise_cell = 0) {
    typedef float T;
    constexpr bool NOISY = MODE == ALAN_MODE_AFFINE || MODE == ALAN_MODE_DOT;
    const bool noise = NOISY && d.noise_on != 0;                       // (uniform)
    const uint64_t noise_base = d.noise_off + noise_cell;
    const uint32_t G = BLOCK ? 64u * NW : (1u << logG);
    uint32_t grp, gl;
    if (BLOCK) {
        grp = block_id;
        gl = threadIdx.x;
    } else {
        const uint32_t gid = block_id * (64u * NW) + threadIdx.x;
        grp = gid >> logG;
        gl = gid & (G - 1u);
    }
    const bool active = grp < d.n_out;
    uint32_t o = active ? grp : d.n_out - 1u;

    int32_t base[MAXF], wbase = 0, obase = 0;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) base[f] = 0;
#pragma unroll
    for (int k = SMALL_NK - 1; k >= 0; --k) {
        const uint32_t q = fd_div(o, d.kdiv[k]);
        const int32_t idx = (int32_t)(o - q * d.kdiv[k].d);
        o = q;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) base[f] += idx * d.fks[f][k];
        if (MODE == ALAN_MODE_WEXPSUM) wbase += idx * d.wks[k];
        obase += idx * d.oks[k];
    }
    float sc[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) sc[f] = d.fscale[f];

    T m = Num<T>::ninf(), s = T(0);
    for (uint32_t r0 = gl; r0 < d.n_red; r0 += UNR * G) {
        T val[UNR][MAXF];
        T wv[UNR];
        int32_t off0[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t ru = r0 + (uint32_t)u * G;
            uint32_t rr = ru < d.n_red ? ru : r0;     // clamped: the slot is masked in accumulate()
            int32_t off[MAXF], woff = wbase;
#pragma unroll
            for (int f = 0; f < MAXF; ++f) off[f] = base[f];
#pragma unroll
            for (int k = SMALL_NR - 1; k >= 0; --k) {
                const uint32_t q = fd_div(rr, d.rdiv[k]);
                const int32_t idx = (int32_t)(rr - q * d.rdiv[k].d);
                rr = q;
#pragma unroll
                for (int f = 0; f < MAXF; ++f) off[f] += idx * d.frs[f][k];
                if (MODE == ALAN_MODE_WEXPSUM) woff += idx * d.wrs[k];
            }
            // (unused factor slots alias factor 0 with zero strides -- harmless to load, but a Normal producer's three factors
            // were six loads per element, 24 per lane and round: the slots beyond nf are skipped, a scalar branch each)
#pragma unroll
            for (int f = 0; f < MAXF; ++f) val[u][f] = f < 2 || f < d.nf ? d.f[f][off[f]] : T(0);
            if (NOISY) {
                if (noise) val[u][1] = noise_at(noise_seed, noise_base + (uint64_t)(int64_t)off[1]);
            }
            wv[u] = MODE == ALAN_MODE_WEXPSUM ? d.w[woff] : 0.f;
            off0[u] = off[0];
        }
#ifdef ALAN_TIMELINE
        if (r0 == gl) SM_STAMP_LANDED(2);
#endif
        if ((MODE == ALAN_MODE_LSE || MODE == ALAN_MODE_SUM) && d.presum_n > 1) {
            // factor 0 is the sum of presum_n slices (role ALAN_PRESUM): the other slices, eight loads per element in
            // flight, added in slice order
            // (PF = 8 in the 256-thread kernels: more costs every one of them registers -- 16 took them to 5 waves per SIMD)
            for (int32_t c0 = 1; c0 < d.presum_n; c0 += PF) {
                T part[UNR][PF];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int32_t c = c0 + i < d.presum_n ? c0 + i : 0;
                        part[u][i] = d.f[0][off0[u] + c * d.presum_stride];
                    }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int i = 0; i < PF; ++i) val[u][0] += c0 + i < d.presum_n ? part[u][i] : T(0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            accumulate<T, MODE>(m, s, val[u], wv[u], sc, d.nf, r0 + (uint32_t)u * G < d.n_red);
    }
    SM_STAMP(3);
    combine_lanes<T, MODE, BLOCK, NW>(m, s, G);
    if (active && gl == 0) {
        T v = (MODE == ALAN_MODE_LSE) ? lse_finish(m, s) : s;
        if (MODE == ALAN_MODE_NORMAL || MODE == ALAN_MODE_NORMAL_LOGSCALE || MODE == ALAN_MODE_BERNOULLI ||
            MODE == ALAN_MODE_PRODUCER_GRAD)
            v *= d.out_scale;
        if (d.ring_n) {
            // one workgroup, one value (try_launch_small checked): deliver it to this replay's slot and move on
            const int32_t slot = *d.ring_counter;
            small_store(d.ring_slots[slot], v + d.add_const);
            *d.ring_counter = slot + 1 == d.ring_n ? 0 : slot + 1;
            if (d.ring_and_out) small_store(d.out + obase, v + d.add_const);
        } else {
            small_store(d.out + obase, v + d.add_const);
        }
    }
}

// ------------------------------------------------------------------------------------------
// ALAN_MODE_BERNOULLI_LINEAR: the Bernoulli producer with its logits computed on the fly,
//   l = sum_t ( a_t  |  sum_e a_t[e] * b_t[e] ),     out = out_scale * sum_R [ logsigmoid(l) - (1 - value) * l ] + add_const
// (what the model's lambda -- `z @ x` -- and td.Bernoulli.log_prob evaluate as a batched GEMM, adds and a producer launch,
// TorchDimDist.py:127-162).  A lane group per output element, lanes along the summed dims; each lane walks the dot
// products of its element serially, all loads of up to 32 events in flight.
// sum_e a[e] b[e] over `len` events: loads issued N at a time, in groups of four that are skipped (a scalar branch) when
// they lie wholly beyond `len` -- an 18-event dot is 20 loads per operand in two round trips.
template <int N>
__device__ __forceinline__ float lin_dot(const float *pa, const float *pb, int len, int as, int bs) {
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int e0 = 0; e0 < len; e0 += N) {
        float av[N], bv[N];
#pragma unroll
        for (int g = 0; g < N / 4; ++g) {
            if (e0 + 4 * g < len) {
#pragma unroll
                for (int i = 4 * g; i < 4 * g + 4; ++i) {
                    const int e = min(e0 + i, len - 1);
                    av[i] = pa[e * as], bv[i] = pb[e * bs];
                }
            } else {
#pragma unroll
                for (int i = 4 * g; i < 4 * g + 4; ++i) av[i] = 0.f, bv[i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            acc0 = fmaf(e0 + i + 0 < len ? av[i + 0] : 0.f, bv[i + 0], acc0);
            acc1 = fmaf(e0 + i + 1 < len ? av[i + 1] : 0.f, bv[i + 1], acc1);
            acc2 = fmaf(e0 + i + 2 < len ? av[i + 2] : 0.f, bv[i + 2], acc2);
            acc3 = fmaf(e0 + i + 3 < len ? av[i + 3] : 0.f, bv[i + 3], acc3);
        }
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

template <bool BLOCK>
__device__ __forceinline__ void lin_body(const LinDesc &d, const int logG, const uint32_t block_id) {
    const uint32_t G = BLOCK ? 256u : (1u << logG);
    uint32_t grp, gl;
    if (BLOCK) {
        grp = block_id;
        gl = threadIdx.x;
    } else {
        const uint32_t gid = block_id * 256u + threadIdx.x;
        grp = gid >> logG;
        gl = gid & (G - 1u);
    }
    const bool active = grp < d.n_out;
    uint32_t o = active ? grp : d.n_out - 1u;
    int32_t abase[LIN_T], bbase[LIN_T], vbase = 0, obase = 0;
#pragma unroll
    for (int tm = 0; tm < LIN_T; ++tm) abase[tm] = bbase[tm] = 0;
#pragma unroll
    for (int k = LIN_NK - 1; k >= 0; --k) {
        const uint32_t q = fd_div(o, d.kdiv[k]);
        const int32_t idx = (int32_t)(o - q * d.kdiv[k].d);
        o = q;
#pragma unroll
        for (int tm = 0; tm < LIN_T; ++tm) {
            abase[tm] += idx * d.aks[tm][k];
            bbase[tm] += idx * d.bks[tm][k];
        }
        vbase += idx * d.vks[k];
        obase += idx * d.oks[k];
    }
    float s = 0.f, m = 0.f;
    // one summed dim (bus_breakdown's plate of 150 observations): offsets are linear in r -- no index decomposition per
    // element (it was a third of the instructions of an element at K = 100: 9 M elements)
    const bool one_dim = d.rdiv[0].d == 1;
    // Every term a plain summand (dot products over dims the output lacks were evaluated beforehand: bus_breakdown at
    // K = 100, 9 M elements of 3 loads + a softplus): eight elements per round, their loads issued together -- one element
    // per round waits out a load latency per element (38 of them in a row per lane: 31.7 us; the arithmetic is 6)
    const bool all_plain = one_dim && d.b[0] == nullptr && d.b[1] == nullptr && d.b[2] == nullptr;
    if (all_plain) {
        constexpr int LU = 8;
        const int32_t vst = d.vrs[LIN_NR - 1];
        int32_t ast[LIN_T];
#pragma unroll
        for (int tm = 0; tm < LIN_T; ++tm) ast[tm] = d.ars[tm][LIN_NR - 1];
        for (uint32_t r0 = gl; r0 < d.n_red; r0 += LU * G) {
            float y[LU], x[LU][LIN_T];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const uint32_t ru = r0 + (uint32_t)u * G;
                const int32_t r = (int32_t)(ru < d.n_red ? ru : r0);          // (clamped: masked below)
                y[u] = d.val[vbase + r * vst];
#pragma unroll
                for (int tm = 0; tm < LIN_T; ++tm) x[u][tm] = tm < d.nt ? d.a[tm][abase[tm] + r * ast[tm]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                float xl = 0.f;
#pragma unroll
                for (int tm = 0; tm < LIN_T; ++tm) xl += x[u][tm];
                const float e = __builtin_amdgcn_exp2f(-fabsf(xl) * 1.44269504088896340736f);
                const float ls = fminf(xl, 0.f) - __builtin_amdgcn_logf(1.f + e) * 0.69314718055994530942f;
                s += r0 + (uint32_t)u * G < d.n_red ? ls - (1.f - y[u]) * xl : 0.f;
            }
        }
    } else
    for (uint32_t r = gl; r < d.n_red; r += G) {
        int32_t aoff[LIN_T], boff[LIN_T], voff = vbase;
#pragma unroll
        for (int tm = 0; tm < LIN_T; ++tm) {
            aoff[tm] = abase[tm];
            boff[tm] = bbase[tm];
        }
        if (one_dim) {
#pragma unroll
            for (int tm = 0; tm < LIN_T; ++tm) {
                aoff[tm] += (int32_t)r * d.ars[tm][LIN_NR - 1];
                boff[tm] += (int32_t)r * d.brs[tm][LIN_NR - 1];
            }
            voff += (int32_t)r * d.vrs[LIN_NR - 1];
        } else {
            uint32_t rr = r;
#pragma unroll
            for (int k = LIN_NR - 1; k >= 0; --k) {
                const uint32_t q = fd_div(rr, d.rdiv[k]);
                const int32_t idx = (int32_t)(rr - q * d.rdiv[k].d);
                rr = q;
#pragma unroll
                for (int tm = 0; tm < LIN_T; ++tm) {
                    aoff[tm] += idx * d.ars[tm][k];
                    boff[tm] += idx * d.brs[tm][k];
                }
                voff += idx * d.vrs[k];
            }
        }
        const float y = d.val[voff];
#ifdef ALAN_TIMELINE
        if (r == gl) SM_STAMP_LANDED(2);
#endif
        float xl = 0.f;
#pragma unroll
        for (int tm = 0; tm < LIN_T; ++tm) {
            if (tm >= d.nt) continue;
            if (d.b[tm] == nullptr) {
                xl += d.a[tm][aoff[tm]];
                continue;
            }
            const float *pa = d.a[tm] + aoff[tm], *pb = d.b[tm] + boff[tm];
            const int len = d.len[tm], as = d.ads[tm], bs = d.bds[tm];
            // (every load of a chunk issued before the first product: the kernel is a chain of load latencies -- four
            // products in flight made movielens' 18-event dot five round trips.  16 at a time (groups of four beyond the
            // length skipped): a chunk of 32 cost the kernel half its waves per SIMD in registers)
            // (17-20 events -- movielens' 18 -- in ONE round trip: 20 loads per operand in flight; the sums' order is the same)
            xl += len > 16 && len <= 20 ? lin_dot<20>(pa, pb, len, as, bs) : lin_dot<16>(pa, pb, len, as, bs);
        }
        // logsigmoid(x) = min(x, 0) - log(1 + exp(-|x|)) on the fast transcendental instructions (1 ulp each; 1 + e in
        // (1, 2] is rounded as the reference's log1p argument is): the accurate expf / log1pf were two thirds of an element
        const float e = __builtin_amdgcn_exp2f(-fabsf(xl) * 1.44269504088896340736f);
        const float ls = fminf(xl, 0.f) - __builtin_amdgcn_logf(1.f + e) * 0.69314718055994530942f;
        s += ls - (1.f - y) * xl;
    }
    SM_STAMP(3);
    combine_lanes<float, ALAN_MODE_SUM, BLOCK>(m, s, G);
    if (active && gl == 0) small_store(d.out + obase, s * d.out_scale + d.add_const);
}

template <int MODE>
__device__ __forceinline__ void small_either(const SmallDesc &d, int logG, bool block, uint32_t bid,
                                             const uint64_t noise_seed = 0, const uint64_t noise_cell = 0) {
    if (block)
        small_body<MODE, true>(d, 8, bid, noise_seed, noise_cell);
    else
        small_body<MODE, false>(d, logG, bid, noise_seed, noise_cell);
}

// Virtual workgroup `vb` of a SmallMulti that sits in the kernel-argument segment at byte offset `arg_off` (its
// descriptors are read through the segment itself -- scalar loads at a uniform offset: indexing the by-value struct with a
// run-time p makes the compiler copy all of it to scratch first).  Every thread of the workgroup calls it (barriers inside).
template <int NP = SMALL_MULTI>
__device__ __forceinline__ void small_multi_block(const size_t arg_off, const uint32_t vb) {
    typedef SmallMultiT<NP> SmallMulti;
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    const char *base = (const char *)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + arg_off);
    const SmallMulti &m = *reinterpret_cast<const SmallMulti *>(base);
#ifdef ALAN_TIMELINE
    const unsigned long long sm_real0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
        for (int i = 0; i < SM_TL_SLOTS; ++i) sm_tl[i] = 0;
    SM_STAMP(0);
#endif
    int p = 0;                                                            // workgroup-uniform
#pragma unroll
    for (int i = 1; i < NP; ++i) p += vb >= m.first_block[i] ? 1 : 0;     // (constant offsets: one wide scalar load, no branch)
    // (every problem's head fetched beside first_block[] -- constant offsets, the same round of scalar loads -- and picked by
    // p afterwards: a fetch at an offset that depends on p was a third dependent round trip in front of the descriptor's)
    typename SmallMulti::Head hd = m.head[0];
#pragma unroll
    for (int i = 1; i < NP; ++i) {
        const typename SmallMulti::Head hi = m.head[i];
        hd.mode = p == i ? hi.mode : hd.mode, hd.logG = p == i ? hi.logG : hd.logG;
        hd.block = p == i ? hi.block : hd.block, hd.first_block = p == i ? hi.first_block : hd.first_block;
    }
    const uint32_t bid = vb - hd.first_block;
    const SmallDesc &d = *reinterpret_cast<const SmallDesc *>(base + offsetof(SmallMulti, d) + (size_t)p * sizeof(SmallDesc));
    const bool block = hd.block != 0;
    const int logG = hd.logG;
#ifdef ALAN_TIMELINE
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(logG), "s"(bid) : "memory");
    SM_STAMP(1);
#endif
    const NoiseLaunch &nz = *reinterpret_cast<const NoiseLaunch *>(base + offsetof(SmallMulti, noise));
    // (the cell: {counter, seed}, written by the previous launch that drew -- read past the scalar and vector L1 caches)
    uint64_t cellv = 0ull, nseed = nz.seed;
    if (nz.cell) {
        cellv = __hip_atomic_load(nz.cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nseed = __hip_atomic_load(nz.cell + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    switch (hd.mode) {
        case ALAN_MODE_LSE: small_either<ALAN_MODE_LSE>(d, logG, block, bid); break;
        case ALAN_MODE_SUM: small_either<ALAN_MODE_SUM>(d, logG, block, bid); break;
        case ALAN_MODE_NORMAL: small_either<ALAN_MODE_NORMAL>(d, logG, block, bid); break;
        case ALAN_MODE_NORMAL_LOGSCALE: small_either<ALAN_MODE_NORMAL_LOGSCALE>(d, logG, block, bid); break;
        case ALAN_MODE_BERNOULLI: small_either<ALAN_MODE_BERNOULLI>(d, logG, block, bid); break;
        case ALAN_MODE_PRODUCER_GRAD: small_either<ALAN_MODE_PRODUCER_GRAD>(d, logG, block, bid); break;
        case ALAN_MODE_WEXPSUM: small_either<ALAN_MODE_WEXPSUM>(d, logG, block, bid); break;   // (per-factor backward launches)
        case ALAN_MODE_DOT: small_either<ALAN_MODE_DOT>(d, logG, block, bid, nseed, cellv); break;
        case ALAN_MODE_AFFINE: small_either<ALAN_MODE_AFFINE>(d, logG, block, bid, nseed, cellv); break;
        case ALAN_MODE_BERNOULLI_LINEAR: {
            const LinDesc &ld = *reinterpret_cast<const LinDesc *>(base + offsetof(SmallMulti, lin));
            if (block)
                lin_body<true>(ld, 8, bid);
            else
                lin_body<false>(ld, logG, bid);
            break;
        }
        case ALAN_MODE_NORMAL_TABLE:                  // (one workgroup: table_prepare, plan.hip)
            nl_table_block(d.f[0], d.fks[0][SMALL_NK - 1], d.frs[0][SMALL_NR - 1], (int)d.n_out, (int)d.n_red, d.fscale[0] == 2.f,
                           reinterpret_cast<unsigned *>(d.out));
            break;
        default: break;
    }
    if (nz.receipt || nz.advance) noise_finish(nz, cellv, nseed);
#ifdef ALAN_TIMELINE
    SM_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SM_STAMP(5);
    if (threadIdx.x == 0 && vb < SM_TL_WGS) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        sm_tl[6] = sm_real0, sm_tl[7] = __builtin_amdgcn_s_memrealtime();
        sm_tl[8] = ((unsigned long long)xcc << 32) | hwid;
        sm_tl[9] = ((unsigned long long)(unsigned)hd.mode << 32) | (unsigned)p;
        sm_tl[10] = gridDim.x;
        for (int i = 0; i < SM_TL_SLOTS; ++i) sm_timeline[vb * SM_TL_SLOTS + i] = sm_tl[i];
    }
#endif
}

}  // namespace alan
