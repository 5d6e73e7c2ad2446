// alan_chain_logmmexp: the timeseries plate.  Replaces utils.py:478-510 (chain_reduce / logmmexp /
// chain_logmmexp) and the trailing t.logsumexp(lp, -1) of logpq.py:139.
//
// The reference reduces [T,K,K] by a pairwise tree (ceil(log2 T) rounds, ~9 torch ops per round; an odd leftover
// matrix is carried to the end of the next round, utils.py:488-495).  The same tree runs here, one launch per
// round, one workgroup per pair, each logmmexp entirely in LDS with the normalisation and eps-in-log of
// utils.py:503-507.  Log-matrix multiplication is associative, so longer left-to-right segments per workgroup
// would need fewer launches (T=1000, K=30: 113 us per ELBO against 127 us) -- but only up to the +eps: on sharply
// peaked transition matrices most entries sit on the eps floor and the result then depends on the bracketing, and
// the reference's gradient is the gradient THROUGH that floor.  Keeping the reference's bracketing keeps both
// (tests/golden/chain_peaked.pt).  Every round's output stays in the workspace (the "tree"): the backward walks it
// top-down, one launch per round.  At T=1000, K=30 the whole input is 3.6 MB: this path is latency-bound.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace alan {

constexpr int CHAIN_THREADS = 256;

template <typename T>
struct alignas(4 * sizeof(T)) Vec4 {
    T x, y, z, w;
};

// Further terms of the chain's input: the factors of a timeseries plate are ADDED into one [T, K, K] tensor before the
// chain (reduce_Ks with no K to sum, logpq.py:128) -- here the first round adds them on load instead (strides may be
// 0: a factor without the K_init dim).  n = 0 on every later round.
// One of the terms may be the TRANSITION log-prob itself, computed on load instead of read (norm != 0): log N(value;
// lmul * loc, scale) with value / loc / scale each a [B, T, K_init, K] view (stride 0 where it lacks a dim) -- for a
// timeseries `ts ~ Normal(lambda prev: c * prev, sigma)` (Timeseries.py:205-245 + TorchDimDist.py:127-162) the
// [T, K_init, K] factor is then never written: value = x[t, k], loc = prev[t, k_init].  norm = 2: `scale` holds log(scale).
template <typename T>
struct ChainAdd {
    const T *p[2];
    int64_t sB[2], sT[2], sR[2], sC[2];
    int32_t n;
    int32_t norm;
    const T *nv, *nl, *ns;
    int64_t nvs[4], nls[4], nss[4];         // (sB, sT, sRow, sCol) of value / loc / scale
    T lmul;
    const T *nl0;                           // optional location of step 0 (then step t >= 1 reads nl at t - 1)
    int64_t nl0s[4];
};

template <typename T>
__device__ __forceinline__ T chain_normal_term(const ChainAdd<T> &ad, int64_t b, int64_t t, int i, int j) {
    const T v = ad.nv[b * ad.nvs[0] + t * ad.nvs[1] + i * ad.nvs[2] + j * ad.nvs[3]];
    const T l = (ad.nl0 && t == 0) ? ad.nl0[b * ad.nl0s[0] + i * ad.nl0s[2] + j * ad.nl0s[3]]
                                   : ad.nl[b * ad.nls[0] + (t - (ad.nl0 ? 1 : 0)) * ad.nls[1] + i * ad.nls[2] + j * ad.nls[3]];
    const T sc = ad.ns[b * ad.nss[0] + t * ad.nss[1] + i * ad.nss[2] + j * ad.nss[3]];
    const T z = v - ad.lmul * l;
    // torch.distributions.Normal.log_prob, as the producer kernels evaluate it (reduce.hip accumulate<NORMAL>)
    return ad.norm == 2 ? -(z * z) * (T(0.5) * Num<T>::exp_acc(T(-2) * sc)) - sc - T(0.91893853320467274178)
                        : -(z * z) / (T(2) * sc * sc) - Num<T>::log(sc) - T(0.91893853320467274178);
}

template <typename T>
__device__ __forceinline__ T chain_in(const T *ms, int64_t off, const ChainAdd<T> &ad, int64_t b, int64_t t, int i, int j) {
    T v = ms[off];
    if (ad.n > 0) v += ad.p[0][b * ad.sB[0] + t * ad.sT[0] + i * ad.sR[0] + j * ad.sC[0]];
    if (ad.n > 1) v += ad.p[1][b * ad.sB[1] + t * ad.sT[1] + i * ad.sR[1] + j * ad.sC[1]];
    if (ad.norm) v += chain_normal_term(ad, b, t, i, j);
    return v;
}

template <typename T>
__device__ __forceinline__ void lds_max(T *addr, T v) {   // ds_max_f32 / ds_max_f64: NaN-ignoring, like fmax
    __hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One workgroup per segment: out[seg] = ms[t0] (x) ms[t0+1] (x) ... (x) ms[t1-1]   (log-space)
// If vec_out != nullptr (final level, one segment) also writes logsumexp(out, -1).
//
// A segment is a chain of DEPENDENT logmmexp steps (utils.py:503-507), so what matters is the latency of
// one step.  Per step: 3 barriers; the row maxima of P and column maxima of C are gathered with LDS
// float-max atomics while those matrices are being written (no separate max pass); the next matrix is
// prefetched from global memory into registers during the step; the [K,K]x[K,K] product runs on 1x4
// register tiles with 16-byte LDS reads (5 reads per 16 FMAs; rows padded to a multiple of 4, pad = 0).
// ME / NT: matrix elements / 1x4 tiles per thread (compile-time, so the per-thread loops carry no dead, predicated
// copies): (4, 1) for K <= 32, (16, 4) for K <= 64, (40, 10) for K <= 100.
template <typename T, int ME, int NT>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_segment_kernel(
    const T *ms, int64_t sB, int64_t sT, int64_t sRow, int64_t sCol, int T_total, int seg_len, int K,
    T *out, T *vec_out, const ChainAdd<T> ad) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // blockIdx.y = which chain of the batch (timeseries plates nested under other plate / K dims: the reference's
    // lp.order(T, K_init, K_curr) leaves those as torchdim batch dims of the matmuls, logpq.py:133-135)
    ms += (int64_t)blockIdx.y * sB;
    if (out) out += (int64_t)blockIdx.y * gridDim.x * K * K;
    if (vec_out) vec_out += (int64_t)blockIdx.y * K;
    const int KP = (K + 3) & ~3, NJG = KP >> 2;
    const int PS = KP + 4;                       // row stride of P: rows on different banks (C rows are read by whole lanes-groups at once)
    T *P = reinterpret_cast<T *>(smem_raw);     // [K][PS]   log-space between steps, exp-space inside one
    T *C = P + (size_t)KP * PS;                  // [KP][KP]
    T *pm = C + (size_t)KP * KP;                 // [2][KP] row maxima of P (double buffered)
    T *cm = pm + 2 * KP;                         // [2][KP] column maxima of C
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * seg_len;
    const int t1 = min(T_total, t0 + seg_len);
    const T NINF = Num<T>::ninf();
    const int KK = K * K;

    for (int i = tid; i < 2 * KP; i += CHAIN_THREADS) {
        pm[i] = NINF;
        cm[i] = NINF;
    }
    __syncthreads();
    // P <- matrix t0, C <- matrix t0+1 (log space), maxima by atomics
    for (int e = tid; e < KK; e += CHAIN_THREADS) {
        const int i = e / K, j = e - i * K;
        const T v = chain_in(ms, (int64_t)t0 * sT + i * sRow + j * sCol, ad, (int64_t)blockIdx.y, (int64_t)t0, i, j);
        P[i * PS + j] = v;
        lds_max(&pm[i], v);
        if (t0 + 1 < t1) {
            const T c = chain_in(ms, (int64_t)(t0 + 1) * sT + i * sRow + j * sCol, ad, (int64_t)blockIdx.y, (int64_t)(t0 + 1), i, j);
            C[i * KP + j] = c;
            lds_max(&cm[j], c);
        }
    }
    __syncthreads();

    int cur = 0;
    T creg[ME];
    auto prefetch = [&](int tt) {   // matrix tt -> registers; consumed one whole step later
        if (tt < t1) {
#pragma unroll
            for (int q = 0; q < ME; ++q) {
                const int e = tid + q * CHAIN_THREADS;
                if (e < KK) {
                    const int i = e / K, j = e - i * K;
                    creg[q] = chain_in(ms, (int64_t)tt * sT + i * sRow + j * sCol, ad, (int64_t)blockIdx.y, (int64_t)tt, i, j);
                }
            }
        }
    };
    prefetch(t0 + 2);
    for (int t = t0 + 1; t < t1; ++t) {
        const T *pmc = pm + cur * KP, *cmc = cm + cur * KP;
        T *pmn = pm + (cur ^ 1) * KP, *cmn = cm + (cur ^ 1) * KP;
        const bool more = t + 1 < t1;
        // ---- exp pass (utils.py:503-505), pads -> 0; the other max buffers are reset for this step's writes
#pragma unroll 4
        for (int e = tid; e < KP * KP; e += CHAIN_THREADS) {
            const int i = e / KP, j = e - i * KP;
            const bool in = i < K && j < K;
            if (i < K) P[i * PS + j] = in ? Num<T>::exp(P[i * PS + j] - pmc[i]) : T(0);
            C[e] = in ? Num<T>::exp(C[e] - cmc[j]) : T(0);
        }
        for (int i = tid; i < KP; i += CHAIN_THREADS) {
            pmn[i] = NINF;
            cmn[i] = NINF;
        }
        __syncthreads();
        // ---- R = Pe @ Ce on 1x4 register tiles
        Vec4<T> acc[NT];
        const Vec4<T> *P4 = reinterpret_cast<const Vec4<T> *>(P);
        const Vec4<T> *C4 = reinterpret_cast<const Vec4<T> *>(C);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tile = tid + q * CHAIN_THREADS;
            Vec4<T> a = {T(0), T(0), T(0), T(0)};
            if (tile < K * NJG) {
                const int i = tile / NJG, jg = tile - i * NJG;
#pragma unroll 4
                for (int k4 = 0; k4 < NJG; ++k4) {
                    const Vec4<T> p = P4[i * (NJG + 1) + k4];
                    const Vec4<T> c0 = C4[(4 * k4) * NJG + jg], c1 = C4[(4 * k4 + 1) * NJG + jg],
                                  c2 = C4[(4 * k4 + 2) * NJG + jg], c3 = C4[(4 * k4 + 3) * NJG + jg];
                    a.x += p.x * c0.x + p.y * c1.x + p.z * c2.x + p.w * c3.x;
                    a.y += p.x * c0.y + p.y * c1.y + p.z * c2.y + p.w * c3.y;
                    a.z += p.x * c0.z + p.y * c1.z + p.z * c2.z + p.w * c3.z;
                    a.w += p.x * c0.w + p.y * c1.w + p.z * c2.w + p.w * c3.w;
                }
            }
            acc[q] = a;
        }
        __syncthreads();
        // ---- P <- log(R + eps) + pm + cm (utils.py:506-507) with its new row maxima; C <- matrix t+1
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tile = tid + q * CHAIN_THREADS;
            if (tile < K * NJG) {
                const int i = tile / NJG, jg = tile - i * NJG;
                const T r[4] = {acc[q].x, acc[q].y, acc[q].z, acc[q].w};
                T mx = NINF;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * jg + c;
                    if (j < K) {
                        const T v = Num<T>::log(r[c] + Num<T>::eps) + pmc[i] + cmc[j];
                        P[i * PS + j] = v;
                        mx = fmax(mx, v);
                    }
                }
                lds_max(&pmn[i], mx);
            }
        }
        if (more) {
#pragma unroll
            for (int q = 0; q < ME; ++q) {
                const int e = tid + q * CHAIN_THREADS;
                if (e < KK) {
                    const int i = e / K, j = e - i * K;
                    C[i * KP + j] = creg[q];
                    lds_max(&cmn[j], creg[q]);
                }
            }
            prefetch(t + 2);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (out)
        for (int e = tid; e < KK; e += CHAIN_THREADS) {
            const int i = e / K, j = e - i * K;
            out[(int64_t)blockIdx.x * KK + e] = P[i * PS + j];
        }
    if (vec_out) {
        // torch.logsumexp(lp, -1)  (logpq.py:139): no eps; -inf rows stay -inf
        for (int i = tid; i < K; i += CHAIN_THREADS) {
            T mx = NINF;
            for (int j = 0; j < K; ++j) mx = fmax(mx, P[i * PS + j]);
            T s = T(0);
            const T mref = (mx == NINF || mx == -NINF) ? T(0) : mx;
            for (int j = 0; j < K; ++j) s += Num<T>::exp(P[i * PS + j] - mref);
            vec_out[i] = Num<T>::log(s) + mref;
        }
    }
}

// Up to THREE rounds of the tree in one launch (K <= 32): a workgroup of 4 x 256 threads takes 8 consecutive nodes;
// its four quarter-groups multiply the four pairs of the first round side by side, two of them the second round's
// pairs, one the third's -- operands of every round in their own LDS slots, so a product's result goes straight into
// its parent's P (left child: with row maxima) or C (right child: with column maxima) slot and, fire-and-forget, into
// the tree in global memory for the backward.  Same arithmetic as chain_segment_kernel, product for product; T = 1000
// is 4 launches of 3 + 3 + 3 + 1 rounds instead of 10.
constexpr int TREE_THREADS = 1024;

template <typename T>
__global__ __launch_bounds__(TREE_THREADS) void chain_tree_kernel(
    const T *ms, int64_t sB, int64_t sT, int64_t sRow, int64_t sCol, int n_in, int K, int rounds,
    T *out1, T *out2, T *out3, int n1, int n2, int n3, T *vec_out, const ChainAdd<T> ad) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int KP = (K + 3) & ~3, NJG = KP >> 2, PS = KP + 4, KK = K * K;
    const int SLOT = KP * PS + KP * KP + 2 * KP;        // P [KP][PS], C [KP][KP], row maxima of P, column maxima of C
    T *base = reinterpret_cast<T *>(smem_raw);
    auto P_of = [&](int s) { return base + (size_t)s * SLOT; };
    auto C_of = [&](int s) { return base + (size_t)s * SLOT + KP * PS; };
    auto pm_of = [&](int s) { return base + (size_t)s * SLOT + KP * PS + KP * KP; };
    auto cm_of = [&](int s) { return base + (size_t)s * SLOT + KP * PS + KP * KP + KP; };
    const int first_slot[4] = {0, 4, 6, 7};             // round 0: slots 0-3, round 1: 4-5, round 2: 6, result: 7
    const int tid = threadIdx.x, g = tid >> 8, gt = tid & 255;
    const int64_t b = blockIdx.y;
    const int seg = blockIdx.x, W = 1 << rounds;        // input nodes per workgroup
    const T NINF = Num<T>::ninf();
    T *const outs[3] = {out1, out2, out3};
    const int ns[3] = {n1, n2, n3};
    ms += b * sB;

    for (int i = tid; i < 8 * 2 * KP; i += TREE_THREADS) pm_of(i / (2 * KP))[i % (2 * KP)] = NINF;
    __syncthreads();
    int m = min(W, n_in - seg * W);                     // nodes entering the current round, in this workgroup
    {
        const int a = 2 * g;
        T *P = P_of(g), *C = C_of(g), *pm = pm_of(g), *cm = cm_of(g);
        for (int e = gt; e < KK; e += 256) {
            const int i = e / K, j = e - i * K;
            if (a < m) {
                const T v = chain_in(ms, (int64_t)(seg * W + a) * sT + i * sRow + j * sCol, ad, b, (int64_t)(seg * W + a), i, j);
                P[i * PS + j] = v;
                lds_max(&pm[i], v);
            }
            if (a + 1 < m) {
                const T c = chain_in(ms, (int64_t)(seg * W + a + 1) * sT + i * sRow + j * sCol, ad, b, (int64_t)(seg * W + a + 1), i, j);
                C[i * KP + j] = c;
                lds_max(&cm[j], c);
            }
        }
    }
    __syncthreads();
    for (int l = 0; l < rounds; ++l) {
        const bool exists = g < (4 >> l) && 2 * g < m, pair = g < (4 >> l) && 2 * g + 1 < m;
        const int s = first_slot[l] + (g < (4 >> l) ? g : 0);
        T *P = P_of(s), *C = C_of(s);
        const T *pm = pm_of(s), *cm = cm_of(s);
        if (pair) {                                     // utils.py:503-505, pads -> 0
            for (int e = gt; e < KP * KP; e += 256) {
                const int i = e / KP, j = e - i * KP;
                const bool in = i < K && j < K;
                if (i < K) P[i * PS + j] = in ? Num<T>::exp(P[i * PS + j] - pm[i]) : T(0);
                C[e] = in ? Num<T>::exp(C[e] - cm[j]) : T(0);
            }
        }
        __syncthreads();
        const int ds = first_slot[l + 1] + (g >> 1);
        const bool asC = g & 1;                         // right child: the parent's C operand
        T *D = asC ? C_of(ds) : P_of(ds);
        T *dmax = asC ? cm_of(ds) : pm_of(ds);
        const int DS = asC ? KP : PS;
        T *gout = (exists && outs[l]) ? outs[l] + ((b * ns[l]) + (int64_t)seg * (W >> (l + 1)) + g) * KK : nullptr;
        if (pair) {
            if (gt < K * NJG) {
                const int i = gt / NJG, jg = gt - i * NJG;
                const Vec4<T> *P4 = reinterpret_cast<const Vec4<T> *>(P);
                const Vec4<T> *C4 = reinterpret_cast<const Vec4<T> *>(C);
                Vec4<T> a = {T(0), T(0), T(0), T(0)};
#pragma unroll 4
                for (int k4 = 0; k4 < NJG; ++k4) {
                    const Vec4<T> p = P4[i * (NJG + 1) + k4];
                    const Vec4<T> c0 = C4[(4 * k4) * NJG + jg], c1 = C4[(4 * k4 + 1) * NJG + jg],
                                  c2 = C4[(4 * k4 + 2) * NJG + jg], c3 = C4[(4 * k4 + 3) * NJG + jg];
                    a.x += p.x * c0.x + p.y * c1.x + p.z * c2.x + p.w * c3.x;
                    a.y += p.x * c0.y + p.y * c1.y + p.z * c2.y + p.w * c3.y;
                    a.z += p.x * c0.z + p.y * c1.z + p.z * c2.z + p.w * c3.z;
                    a.w += p.x * c0.w + p.y * c1.w + p.z * c2.w + p.w * c3.w;
                }
                const T r[4] = {a.x, a.y, a.z, a.w};
                T mx = NINF;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 4 * jg + c;
                    if (j < K) {
                        const T v = Num<T>::log(r[c] + Num<T>::eps) + pm[i] + cm[j];     // utils.py:506-507
                        D[i * DS + j] = v;
                        if (gout) gout[i * K + j] = v;
                        if (asC) lds_max(&dmax[j], v);
                        mx = fmax(mx, v);
                    }
                }
                if (!asC) lds_max(&dmax[i], mx);
            }
        } else if (exists) {                            // the leftover of this round passes through (utils.py:488-495)
            for (int e = gt; e < KK; e += 256) {
                const int i = e / K, j = e - i * K;
                const T v = P[i * PS + j];
                D[i * DS + j] = v;
                if (gout) gout[e] = v;
                lds_max(&dmax[asC ? j : i], v);
            }
        }
        __syncthreads();
        m = (m + 1) >> 1;
    }
    if (vec_out) {                                      // torch.logsumexp(lp, -1)  (logpq.py:139): no eps
        const T *R = P_of(first_slot[rounds]);
        for (int i = tid; i < K; i += TREE_THREADS) {
            T mx = NINF;
            for (int j = 0; j < K; ++j) mx = fmax(mx, R[i * PS + j]);
            T sum = T(0);
            const T mref = (mx == NINF || mx == -NINF) ? T(0) : mx;
            for (int j = 0; j < K; ++j) sum += Num<T>::exp(R[i * PS + j] - mref);
            vec_out[b * K + i] = Num<T>::log(sum) + mref;
        }
    }
}

// One product of the tree for 32 < K <= 128 (fp32) with the K x K x K contraction on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact k-ordered fp32 fmaf chains, the arithmetic of the vector loop).  What the matrix
// instruction buys here is LDS traffic, not FLOPs (fp32 MFMA runs at the vector rate): the 1x4 register tiles of
// chain_segment_kernel read 5 x 16 bytes per 16 FMAs -- 5.2 MB through the LDS per K = 100 product, 20 us of a 40 us
// workgroup -- the 32x32 tiles read 8 bytes per 4096 FMAs.  A workgroup of KT*KT/NTW waves takes one pair (P, C);
// a wave owns NTW output tiles of one tile row: per four k values one 16-byte read of its P rows (shared by its
// tiles) and four dword reads per tile down C's rows, then four MFMAs per tile (lane half h takes k = 4q + h and
// 4q + 2 + h: the order of k inside a step is free as long as A and B agree).
// Same normalisation, eps and bracketing as utils.py:503-507; an odd leftover passes through (utils.py:488-495).
typedef float chain_f32x16 __attribute__((ext_vector_type(16)));
typedef float chain_f32x4 __attribute__((ext_vector_type(4)));

template <int KT, int NTW, bool ADD = true>     // ADD: further terms on load (the first round only: every load then carries their branches)
__global__ __launch_bounds__(64 * KT * KT / NTW) void chain_pair_mfma_kernel(
    const float *ms, int64_t sB, int64_t sT, int64_t sRow, int64_t sCol, int n_src, int K,
    float *out, float *vec_out, const ChainAdd<float> ad) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int KP = 32 * KT, S = KP + 4, NTH = 64 * KT * KT / NTW;
    float *Pe = reinterpret_cast<float *>(smem_raw);      // [KP][S]  log-space, then exp-space, then the result R
    float *Ce = Pe + KP * S;                              // [KP][S]
    float *pm = Ce + KP * S, *cm = pm + KP;               // row maxima of P, column maxima of C
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.y, node = blockIdx.x;
    const int t0 = 2 * (int)node, t1 = t0 + 1, KK = K * K;
    const float NINF = -__builtin_huge_valf();
    ms += b * sB;
    float *o = out ? out + (b * gridDim.x + node) * (int64_t)KK : nullptr;
    const bool pair = t1 < n_src;

    for (int i = tid; i < 2 * KP; i += NTH) pm[i] = NINF;
    __syncthreads();
    // A wave per row, lanes across the columns (coalesced): P's row maximum by a wave reduction, C's column maxima
    // kept per lane over the wave's rows and merged across the waves at the end.  Every load of the wave is issued
    // before the first one is used: a product is a chain of dependent launches reading what the previous one wrote,
    // and a loop of load -> reduce -> next row pays that latency a dozen times.
    {
        constexpr int NW = NTH / 64, NC = KP / 64, RPW = KP / NW;
        float pv[RPW][NC], cv[RPW][NC];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int i = wave + NW * rr;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int jj = lane + 64 * c;
                const bool in = i < K && jj < K;
                const int ic = min(i, K - 1), jc = min(jj, K - 1);
                const int64_t o0 = (int64_t)t0 * sT + ic * sRow + jc * sCol, o1 = (int64_t)t1 * sT + ic * sRow + jc * sCol;
                pv[rr][c] = ADD ? chain_in(ms, o0, ad, b, (int64_t)t0, ic, jc) : ms[o0];
                cv[rr][c] = !pair ? NINF : ADD ? chain_in(ms, o1, ad, b, (int64_t)t1, ic, jc) : ms[o1];
                if (!in) pv[rr][c] = NINF, cv[rr][c] = NINF;
            }
        }
        float cmax[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) cmax[c] = NINF;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int i = wave + NW * rr;
            float rmax = NINF;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int jj = lane + 64 * c;
                if (i < K && jj < K) {
                    Pe[i * S + jj] = pv[rr][c];
                    if (pair) Ce[i * S + jj] = cv[rr][c];
                }
                rmax = fmaxf(rmax, pv[rr][c]);
                cmax[c] = fmaxf(cmax[c], cv[rr][c]);
            }
#pragma unroll
            for (int o_ = 32; o_ >= 1; o_ >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o_));
            if (lane == 0 && i < K) pm[i] = rmax;
        }
        if (pair) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (lane + 64 * c < K) lds_max(&cm[lane + 64 * c], cmax[c]);
        }
    }
    __syncthreads();
    if (pair) {
        for (int e = tid; e < KP * KP; e += NTH) {        // utils.py:503-505, pads -> 0
            const int i = e / KP, jj = e - i * KP;
            const bool in = i < K && jj < K;
            Pe[i * S + jj] = in ? Num<float>::exp(Pe[i * S + jj] - pm[i]) : 0.f;
            Ce[i * S + jj] = in ? Num<float>::exp(Ce[i * S + jj] - cm[jj]) : 0.f;
        }
        __syncthreads();
        const int tr = (wave * NTW) / KT, tc0 = (wave * NTW) - tr * KT;
        chain_f32x16 acc[NTW];
#pragma unroll
        for (int q = 0; q < NTW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        const float *prow = Pe + (32 * tr + j) * S;
        const int nq4 = (K + 3) >> 2;
        for (int q4 = 0; q4 < nq4; ++q4) {
            const chain_f32x4 a4 = *reinterpret_cast<const chain_f32x4 *>(prow + 4 * q4);
            const float a_lo = h ? a4[1] : a4[0], a_hi = h ? a4[3] : a4[2];
            const float *c_lo = Ce + (4 * q4 + h) * S + 32 * tc0 + j, *c_hi = c_lo + 2 * S;
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, c_lo[32 * q], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, c_hi[32 * q], acc[q], 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wave is done reading Pe / Ce: R goes into Pe
        // ---- R = log(Pe @ Ce + eps) + pm + cm (utils.py:506-507)
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            const int col = 32 * (tc0 + q) + j;
            const float cmj = col < K ? cm[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * tr + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K && col < K) {
                    const float v = Num<float>::log(acc[q][r] + Num<float>::eps) + pm[row] + cmj;
                    Pe[row * S + col] = v;
                    if (o) o[row * K + col] = v;
                }
            }
        }
    } else if (o) {                                        // the leftover of this round passes through
        for (int e = tid; e < KK; e += NTH) o[e] = Pe[(e / K) * S + (e % K)];
    }
    if (vec_out) {                                         // torch.logsumexp(lp, -1)  (logpq.py:139): no eps
        __syncthreads();
        for (int i = tid; i < K; i += NTH) {
            float mx = NINF;
            for (int jj = 0; jj < K; ++jj) mx = fmaxf(mx, Pe[i * S + jj]);
            float sum = 0.f;
            const float mref = (mx == NINF || mx == -NINF) ? 0.f : mx;
            for (int jj = 0; jj < K; ++jj) sum += Num<float>::exp(Pe[i * S + jj] - mref);
            vec_out[b * K + i] = Num<float>::log(sum) + mref;
        }
    }
}

// The same product split over KT workgroups, one per tile ROW of the result (blockIdx.z): each loads its 32 rows of P
// and ALL of C (whose column maxima it needs anyway), so nothing is exchanged between workgroups.  For the late rounds
// of the tree, where a launch holds a handful of products and lasts as long as ONE of them: the exp pass over P, the
// MFMA phase and the log epilogue shrink by KT, the loads of C do not (early rounds keep one workgroup per product:
// there the launch is throughput-bound and reading C KT times would cost more than it saves).  No on-load terms.
template <int KT>
__global__ __launch_bounds__(64 * KT * 2) void chain_pair_stripe_kernel(const float *ms, int64_t sB, int64_t sT, int64_t sRow,
                                                                        int64_t sCol, int n_src, int K, float *out,
                                                                        float *vec_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int KP = 32 * KT, S = KP + 4, NTH = 64 * KT * 2, NW = NTH / 64, NC = KP / 64, RPW = KP / NW, RPS = 32 / NW;
    float *Pe = reinterpret_cast<float *>(smem_raw);      // [32][S]   this workgroup's rows of P
    float *Ce = Pe + 32 * S;                              // [KP][S]
    float *pm = Ce + KP * S, *cm = pm + 32;               // row maxima of the stripe, column maxima of C
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.y, node = blockIdx.x;
    const int sr = blockIdx.z, row0 = 32 * sr;
    const int t0 = 2 * (int)node, t1 = t0 + 1, KK = K * K;
    const float NINF = -__builtin_huge_valf();
    ms += b * sB;
    float *o = out ? out + (b * gridDim.x + node) * (int64_t)KK : nullptr;
    const bool pair = t1 < n_src;
    for (int i = tid; i < 32 + KP; i += NTH) pm[i] = NINF;
    __syncthreads();
    {
        float pv[RPS][NC], cv[RPW][NC];
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const int ic = min(row0 + wave + NW * rr, K - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) pv[rr][c] = ms[(int64_t)t0 * sT + ic * sRow + min(lane + 64 * c, K - 1) * sCol];
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int ic = min(wave + NW * rr, K - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c)
                cv[rr][c] = pair ? ms[(int64_t)t1 * sT + ic * sRow + min(lane + 64 * c, K - 1) * sCol] : NINF;
        }
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const int li = wave + NW * rr, i = row0 + li;
            float rmax = NINF;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int jj = lane + 64 * c;
                const float v = (i < K && jj < K) ? pv[rr][c] : NINF;
                if (i < K && jj < K) Pe[li * S + jj] = v;
                rmax = fmaxf(rmax, v);
            }
#pragma unroll
            for (int o_ = 32; o_ >= 1; o_ >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o_));
            if (lane == 0 && i < K) pm[li] = rmax;
        }
        if (pair) {
            float cmax[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) cmax[c] = NINF;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int i = wave + NW * rr;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int jj = lane + 64 * c;
                    if (i < K && jj < K) {
                        Ce[i * S + jj] = cv[rr][c];
                        cmax[c] = fmaxf(cmax[c], cv[rr][c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (lane + 64 * c < K) lds_max(&cm[lane + 64 * c], cmax[c]);
        }
    }
    __syncthreads();
    if (pair) {
        for (int e = tid; e < 32 * KP; e += NTH) {        // utils.py:503-505, pads -> 0
            const int li = e / KP, jj = e - li * KP;
            const bool in = row0 + li < K && jj < K;
            Pe[li * S + jj] = in ? Num<float>::exp(Pe[li * S + jj] - pm[li]) : 0.f;
        }
        for (int e = tid; e < KP * KP; e += NTH) {
            const int i = e / KP, jj = e - i * KP;
            const bool in = i < K && jj < K;
            Ce[i * S + jj] = in ? Num<float>::exp(Ce[i * S + jj] - cm[jj]) : 0.f;
        }
        __syncthreads();
        chain_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (wave < KT) {                                  // one tile of the stripe per wave: tile column = wave
            const float *prow = Pe + j * S;
            const int nq4 = (K + 3) >> 2;
            for (int q4 = 0; q4 < nq4; ++q4) {
                const chain_f32x4 a4 = *reinterpret_cast<const chain_f32x4 *>(prow + 4 * q4);
                const float a_lo = h ? a4[1] : a4[0], a_hi = h ? a4[3] : a4[2];
                const float *c_lo = Ce + (4 * q4 + h) * S + 32 * wave + j, *c_hi = c_lo + 2 * S;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, c_lo[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, c_hi[0], acc, 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wave is done reading Pe: R goes into Pe
        if (wave < KT) {                                   // R = log(Pe @ Ce + eps) + pm + cm (utils.py:506-507)
            const int col = 32 * wave + j;
            const float cmj = col < K ? cm[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = (r & 3) + 8 * (r >> 2) + 4 * h, row = row0 + li;
                if (row < K && col < K) {
                    const float v = Num<float>::log(acc[r] + Num<float>::eps) + pm[li] + cmj;
                    Pe[li * S + col] = v;
                    if (o) o[row * K + col] = v;
                }
            }
        }
    } else if (o) {                                        // the leftover of this round passes through
        for (int e = tid; e < 32 * K; e += NTH) {
            const int li = e / K, jj = e - li * K;
            if (row0 + li < K) o[(row0 + li) * K + jj] = Pe[li * S + jj];
        }
    }
    if (vec_out) {                                         // torch.logsumexp(lp, -1)  (logpq.py:139): no eps
        __syncthreads();
        if (tid < 32 && row0 + tid < K) {
            float mx = NINF;
            for (int jj = 0; jj < K; ++jj) mx = fmaxf(mx, Pe[tid * S + jj]);
            float sum = 0.f;
            const float mref = (mx == NINF || mx == -NINF) ? 0.f : mx;
            for (int jj = 0; jj < K; ++jj) sum += Num<float>::exp(Pe[tid * S + jj] - mref);
            vec_out[b * K + row0 + tid] = Num<float>::log(sum) + mref;
        }
    }
}

// (the launcher of the kernel above: fp32 only)
template <typename T>
static int launch_pair_mfma(int64_t, uint32_t, uint32_t, hipStream_t, const T *, int64_t, int64_t, int64_t, int64_t, int,
                            T *, T *, const ChainAdd<T> &) {
    return ALAN_ERR_UNSUPPORTED;
}
template <>
int launch_pair_mfma<float>(int64_t K, uint32_t n_out, uint32_t B, hipStream_t stream, const float *src, int64_t cB,
                            int64_t cT, int64_t cR, int64_t cC, int n_src, float *dst, float *vec_out,
                            const ChainAdd<float> &ad) {
    if (K <= 32 || K > 128) return ALAN_ERR_UNSUPPORTED;
    const int kt = K <= 64 ? 2 : 4;
    const size_t smem = (size_t)(2 * 32 * kt * (32 * kt + 4) + 2 * 32 * kt) * sizeof(float);
    auto go = [&](auto kern, int threads) {
        if (smem > 64 * 1024)
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return ALAN_ERR_LAUNCH;
        ALAN_LAUNCH(kern, dim3(n_out, B), dim3(threads), smem, stream, src, cB, cT, cR, cC, n_src, (int)K, dst,
                           vec_out, ad);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
    const bool add = ad.n > 0 || ad.norm != 0;
    const uint32_t stripe_below = 96u;                // products per launch below which a product is split (tuned: DESIGN.md)
    if (!add && (uint64_t)n_out * B < stripe_below) {
        // a launch of few products lasts as long as one of them: one workgroup per tile row of each
        const size_t ssm = (size_t)((32 + 32 * kt) * (32 * kt + 4) + 32 + 32 * kt) * sizeof(float);
        auto gos = [&](auto kern, int threads) {
            if (ssm > 64 * 1024)
                if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssm) != hipSuccess)
                    return ALAN_ERR_LAUNCH;
            ALAN_LAUNCH(kern, dim3(n_out, B, (uint32_t)kt), dim3(threads), ssm, stream, src, cB, cT, cR, cC, n_src,
                               (int)K, dst, vec_out);
            return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
        };
        return kt == 2 ? gos(chain_pair_stripe_kernel<2>, 256) : gos(chain_pair_stripe_kernel<4>, 512);
    }
    if (!add) return kt == 2 ? go(chain_pair_mfma_kernel<2, 1, false>, 256) : go(chain_pair_mfma_kernel<4, 2, false>, 512);
    return kt == 2 ? go(chain_pair_mfma_kernel<2, 1>, 256) : go(chain_pair_mfma_kernel<4, 2>, 512);
}

// ------------------------------------------------------------------------------------------------
// Up to FIVE rounds of the tree in one launch for K <= 32, fp32, one WAVE per product (the K x K x K contraction is 16
// v_mfma_f32_32x32x2_f32 on operands held in registers).  A workgroup of 16 waves takes 32 consecutive nodes: 16
// products side by side in its first round, then 8, 4, 2, 1; T = 1000 is 2 launches (5 + 5 rounds) instead of the 4 of
// chain_tree_kernel, whose rounds each cost several barriers and an LDS-operand vector product.
//   * operand layouts: lane (i, h) of the LEFT child holds P[i][k(r, h)], lane (j, h) of the RIGHT child holds
//     C[k(r, h)][j], r = 0..15, with k(r, h) = (r & 3) + 8 (r >> 2) + 4 h -- the row a lane's accumulator register r
//     stands for.  The order of k inside the contraction is free as long as A and B agree, so step r of the MFMA chain
//     takes the pair k(r, 0), k(r, 1).
//   * a product's result goes to an LDS tile [32][33] (read next round in either layout, conflict-free both ways) and,
//     fire-and-forget, into the tree in global memory for the backward; pads (row or column >= K) are kept at -inf.
// Same normalisation, eps and bracketing as utils.py:503-507; an odd leftover passes through (utils.py:488-495).
constexpr int WAVE_ROUNDS = 5, WAVE_TILE = 32 * 33;

struct WaveOuts {
    float *o[WAVE_ROUNDS];
    int n[WAVE_ROUNDS];
};

// The parent's contraction of a single chain, run by the last launch's one workgroup behind its last round
// (alan_chain_final_t): out = log(sum_k exp(x_k - max) + eps) + max + add_const, x_k = vec[k] + sum_f extra_f[k]
// (reduce_Ks.py:249-251 + utils.py:218-220 on [K_init] vectors: an evaluation's last launch, one fewer).
struct WaveFinal {
    const float *extra[3];
    int32_t stride[3];
    int32_t n_extra, on;
    float add_const;
    float *out;
    float *const *ring_slots;
    int32_t *ring_counter;
    int32_t ring_n;
};

// MAXT: the largest workgroup the instantiation is launched with (launches of <= 3 rounds: 256 threads -- one wave per
// SIMD, so the loads of the on-load terms have registers to be in flight in)
template <int NADD, bool NORM = false, int MAXT = 1024>   // further terms added to the input on load (ChainAdd): 0..2 tensors, the Normal term
__global__ __launch_bounds__(MAXT) void chain_wave_kernel(const float *ms, int64_t sB, int64_t sT, int64_t sRow,
                                                          int64_t sCol, int n_in, int K, int rounds, const WaveOuts outs,
                                                          float *vec_out, const ChainAdd<float> ad,
                                                          const WaveFinal fin = WaveFinal()) {
    extern __shared__ __align__(16) float wl[];
    const int NW = blockDim.x >> 6;                      // waves = products of the first round = 2^(rounds - 1)
    float *tiles = wl;                                   // NW tiles written by rounds 0, 2, 4, then NW / 2 by rounds 1, 3
    float *pmv = wl + (NW + NW / 2 + 1) * WAVE_TILE;     // [NW][32]: the row maxima of a wave's left operand
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h4 = (lane >> 5) * 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: branches on it are scalar, pointers SGPRs
    const int64_t b = blockIdx.y;
    const int seg = blockIdx.x, W = 1 << rounds;
    const float NINF = -__builtin_huge_valf();
    const uint32_t OOB = 0x80000000u;
    ms += b * sB;
    int m = min(W, n_in - seg * W);                      // nodes entering the current round, in this workgroup
#define KROW(r) (((r) & 3) + 8 * ((r) >> 2) + h4)

    for (int l = 0; l < rounds; ++l) {
        const int nprod = NW >> l;                       // waves with a node to make this round
        const bool exists = w < nprod && 2 * w < m, pair = w < nprod && 2 * w + 1 < m;
        const float *src_tiles = tiles + ((l & 1) ? 0 : NW * WAVE_TILE);      // what round l - 1 wrote
        float *dst = tiles + ((l & 1) ? NW * WAVE_TILE : 0) + w * WAVE_TILE;
        // this node in the tree (global memory, for the backward): a buffer of K * K floats; lanes outside the matrix
        // get an offset beyond it and the hardware drops their stores
        const bool store = exists && outs.o[l] != nullptr;
        const float *node = store ? outs.o[l] + ((b * outs.n[l]) + (int64_t)seg * (W >> (l + 1)) + w) * (int64_t)(K * K) : ms;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(node), 0, store ? K * K * 4 : 0, 0x00020000);
        const uint32_t lane_boff = (j < K) ? (uint32_t)((h4 * K + j) * 4) : OOB;
        float pa[16], pb[16];
        if (exists) {
            // ---- operands: round 0 from global memory (pads -> -inf), later rounds from the tiles
            if (l == 0) {
                // 32-bit element offsets from wave-uniform bases (the host checked that a matrix spans < 2^31 elements)
                auto load_node = [&](int64_t tn, bool rowwise, float(&dstv)[16]) {
                    const float *p0 = ms + tn * sT;
                    const float *p1 = NADD > 0 ? ad.p[0] + b * ad.sB[0] + tn * ad.sT[0] : p0;
                    const float *p2 = NADD > 1 ? ad.p[1] + b * ad.sB[1] + tn * ad.sT[1] : p0;
                    const int r0 = (int)sRow, c0 = (int)sCol, r1 = (int)ad.sR[0], c1 = (int)ad.sC[0], r2 = (int)ad.sR[1],
                              c2 = (int)ad.sC[1];
                    // the Normal term's operands, likewise: uniform bases, 32-bit offsets
                    const float *qv = NORM ? ad.nv + b * ad.nvs[0] + tn * ad.nvs[1] : p0;
                    const bool first = NORM && ad.nl0 && tn == 0;                 // (the previous state of step 0: its own source)
                    const float *ql = !NORM ? p0 : first ? ad.nl0 + b * ad.nl0s[0] : ad.nl + b * ad.nls[0] + (tn - (ad.nl0 ? 1 : 0)) * ad.nls[1];
                    const float *qs = NORM ? ad.ns + b * ad.nss[0] + tn * ad.nss[1] : p0;
                    // (host-checked: the value varies along the columns only, the location along the rows only -- a
                    // transition's x[t, k] and prev[t, k_init] -- so one of the two is fixed per lane)
                    const int vc = (int)ad.nvs[3], lr = first ? (int)ad.nl0s[2] : (int)ad.nls[2];
                    const int jc = j < K ? j : 0;
                    const float fixed = !NORM ? 0.f : rowwise ? ad.lmul * ql[jc * lr] : qv[jc * vc];
                    // (the scale does not vary inside a matrix -- the host checked: the usual scalar -- so its weight
                    // 1 / (2 scale^2) and log-normaliser are taken once per node)
                    float w_uni = 0.f, lg_uni = 0.f;
                    if (NORM) {
                        const float sc = qs[0];
                        w_uni = ad.norm == 2 ? 0.5f * expf(-2.f * sc) : 0.5f / (sc * sc);
                        lg_uni = (ad.norm == 2 ? sc : logf(sc)) + 0.91893853320467274178f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = KROW(r);
                        const bool ok = j < K && k < K;
                        const int ii = ok ? (rowwise ? j : k) : 0, jj = ok ? (rowwise ? k : j) : 0;
                        float v = p0[ii * r0 + jj * c0];
                        if (NADD > 0) v += p1[ii * r1 + jj * c1];
                        if (NADD > 1) v += p2[ii * r2 + jj * c2];
                        dstv[r] = ok ? v : NINF;
                    }
                    if (NORM) {
                        // a pass of its own (pads stay -inf: -inf + finite), its loads kept apart from the ones above:
                        // sixteen elements of four tensors in flight at once do not fit the 128 registers of a 16-wave group
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int k = KROW(r);
                            const int kc = k < K ? k : 0;
                            const float z = rowwise ? qv[kc * vc] - fixed : fixed - ad.lmul * ql[kc * lr];     // value - loc
                            dstv[r] += -(z * z) * w_uni - lg_uni;
                        }
                    }
                };
                const int64_t ta = (int64_t)seg * W + 2 * w;
                load_node(ta, pair, pa);                 // (a leftover is read like a right child: the tile layout)
                if (pair) load_node(ta + 1, false, pb);
            } else {
                const float *TL = src_tiles + (2 * w) * WAVE_TILE, *TR = TL + WAVE_TILE;
                const float *la = pair ? TL + j * 33 + h4 : TL + h4 * 33 + j;       // row-wise (left child) / tile layout
                const int stepa = pair ? 1 : 33;
                const float *lb = TR + h4 * 33 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kc = (r & 3) + 8 * (r >> 2);
                    pa[r] = la[kc * stepa];
                    if (pair) pb[r] = lb[kc * 33];
                }
            }
        }
        if (pair) {
            float pm = pa[0], cm = pb[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) pm = fmaxf(pm, pa[r]), cm = fmaxf(cm, pb[r]);
            pm = fmaxf(pm, __shfl_xor(pm, 32));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            chain_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)                  // utils.py:503-505 (pads: exp(-inf - finite) = 0)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Num<float>::exp(pa[r] - pm), Num<float>::exp(pb[r] - cm), acc, 0,
                                                           0, 0);
            if (h4 == 0) pmv[w * 32 + j] = pm;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const chain_f32x4 *pm4 = reinterpret_cast<const chain_f32x4 *>(pmv + w * 32 + h4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const chain_f32x4 pmq = pm4[2 * q];       // rows 8 q + h4 .. + 3
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int r = 4 * q + c, kc = c + 8 * q;
                    float v = __logf(acc[r] + Num<float>::eps) + pmq[c] + cm;     // utils.py:506-507 (acc + eps >= eps: v_log_f32 is exact to 1 ulp there)
                    const bool in = kc + h4 < K && j < K;
                    v = in ? v : NINF;
                    dst[(kc + h4) * 33 + j] = v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc,
                                                          kc + h4 < K ? lane_boff : OOB, kc * K * 4, 0);
                }
            }
        } else if (exists) {                              // the leftover of this round passes through (utils.py:488-495)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kc = (r & 3) + 8 * (r >> 2);
                dst[(kc + h4) * 33 + j] = pa[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, pa[r]), rsrc,
                                                      kc + h4 < K ? lane_boff : OOB, kc * K * 4, 0);
            }
        }
        __syncthreads();
        m = (m + 1) >> 1;
    }
#undef KROW
    if (vec_out && w == 0) {                             // torch.logsumexp(lp, -1)  (logpq.py:139): no eps
        // lane (i, h): row i, columns 16 h .. 16 h + 15; the halves are merged by one exchange
        const float *R = tiles + (((rounds - 1) & 1) ? NW * WAVE_TILE : 0) + j * 33 + 4 * h4;
        float x[16], mx = NINF;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            x[c] = R[c];                                  // (pads hold -inf)
            mx = fmaxf(mx, x[c]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mref = (mx == NINF || mx == -NINF) ? 0.f : mx;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sum += expf(x[c] - mref);
        sum += __shfl_xor(sum, 32);
        const float v = logf(sum) + mref;
        if (lane < K) vec_out[b * K + lane] = v;
        if (fin.on) {                                      // (one chain: b = 0)
            float x = lane < K ? v : NINF;
#pragma unroll
            for (int f = 0; f < 3; ++f)
                if (f < fin.n_extra) x += fin.extra[f][min(lane, K - 1) * fin.stride[f]];
            float fm = x, fs = lane < K ? 1.f : 0.f;        // (x - max, summed: the lane's own term is exp(0))
            // lanes >= K hold (-inf, 0): lse_merge leaves them out; a NaN term poisons the sum
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                const float m2 = __shfl_xor(fm, ofs), s2 = __shfl_xor(fs, ofs);
                lse_merge(fm, fs, m2, s2);
            }
            if (lane == 0) {
                const float res = lse_finish(fm, fs) + fin.add_const;
                if (fin.ring_n) {
                    const int32_t slot = *fin.ring_counter;
                    *fin.ring_slots[slot] = res;
                    *fin.ring_counter = slot + 1 == fin.ring_n ? 0 : slot + 1;
                } else {
                    *fin.out = res;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The tree in memory: round r = 1..L holds n_r = ceil(n_{r-1} / 2) matrices per chain ([B][n_r][K][K], each round
// 256-byte aligned), n_0 = T, n_L = 1 (the root = chain_logmmexp(ms)); T = 1 has one round holding a copy of ms[0].
struct TreeLayout {
    int L = 0;
    int64_t n[34];          // n[0] = T
    size_t off[34];         // byte offset of round r (off[0] unused)
    size_t bytes = 0;
};

static TreeLayout tree_layout(int64_t B, int64_t T, int64_t K, size_t elt) {
    TreeLayout t;
    t.n[0] = T;
    int64_t n = T;
    do {
        n = (n + 1) / 2;
        ++t.L;
        t.n[t.L] = n;
        t.off[t.L] = t.bytes;
        t.bytes += ((size_t)(B * n * K * K) * elt + 255) & ~(size_t)255;
    } while (n > 1);
    return t;
}

template <typename T>
static int chain_run(const void *ms_, int64_t B, int64_t Tn, int64_t K, int64_t sB, int64_t sT, int64_t sRow,
                     int64_t sCol, void *out_chain, void *out_vec, void *ws, size_t ws_bytes, hipStream_t stream,
                     const void *const *more = nullptr, const int64_t *more_strides = nullptr, int n_more = 0,
                     const alan_chain_normal_t *normal = nullptr, const alan_chain_final_t *final_ = nullptr) {
    ChainAdd<T> ad0, none;
    std::memset(&ad0, 0, sizeof(ad0));
    std::memset(&none, 0, sizeof(none));
    if (n_more < 0 || n_more > 2) return ALAN_ERR_UNSUPPORTED;
    if (normal) {
        if (!normal->value || !normal->loc || !normal->scale) return ALAN_ERR_BAD_DESC;
        ad0.norm = normal->log_scale ? 2 : 1;
        ad0.nv = (const T *)normal->value, ad0.nl = (const T *)normal->loc, ad0.ns = (const T *)normal->scale;
        for (int q = 0; q < 4; ++q)
            ad0.nvs[q] = normal->v_stride[q], ad0.nls[q] = normal->l_stride[q], ad0.nss[q] = normal->s_stride[q];
        ad0.lmul = (T)normal->loc_mul;
        ad0.nl0 = (const T *)normal->loc0;
        for (int q = 0; q < 4; ++q) ad0.nl0s[q] = normal->loc0 ? normal->l0_stride[q] : 0;
    }
    for (int q = 0; q < n_more; ++q) {
        if (!more || !more[q] || !more_strides) return ALAN_ERR_BAD_DESC;
        ad0.p[q] = (const T *)more[q];
        ad0.sB[q] = more_strides[4 * q], ad0.sT[q] = more_strides[4 * q + 1];
        ad0.sR[q] = more_strides[4 * q + 2], ad0.sC[q] = more_strides[4 * q + 3];
    }
    ad0.n = n_more;
    const size_t KP = (size_t)((K + 3) & ~3);
    const size_t smem = (KP * (KP + 4) + KP * KP + 4 * KP) * sizeof(T);
    if (K > 100 || smem > 160 * 1024) return ALAN_ERR_UNSUPPORTED;
    auto kern = K <= 32 ? chain_segment_kernel<T, 4, 1> : K <= 64 ? chain_segment_kernel<T, 16, 4>
                                                                  : chain_segment_kernel<T, 40, 10>;
    if (smem > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    const TreeLayout tl = tree_layout(B, Tn, K, sizeof(T));
    if (!ws || ws_bytes < tl.bytes) return ALAN_ERR_WORKSPACE;
    const T *src = (const T *)ms_;
    int64_t cB = sB, cT = sT, cR = sRow, cC = sCol;
    static const bool no_wave = env_knob("ALAN_CHAIN_WAVE") == 0;                     // ablation: chain_tree_kernel instead
    bool fits32 = true;                                   // the wave kernel's 32-bit offsets inside one matrix
    {
        auto span = [&](int64_t r, int64_t c) { return ((r < 0 ? -r : r) + (c < 0 ? -c : c)) * (K - 1) < (1ll << 31) - 1; };
        fits32 = span(sRow, sCol);
        for (int q = 0; q < n_more; ++q) fits32 = fits32 && span(ad0.sR[q], ad0.sC[q]);
        // (the wave kernel takes the Normal term's scale once per matrix: it must not vary inside one)
        // and its value along the columns only, its location along the rows only (one of them is then fixed per lane)
        if (normal)
            fits32 = fits32 && span(0, ad0.nvs[3]) && span(ad0.nls[2], 0) && ad0.nss[2] == 0 && ad0.nss[3] == 0 &&
                     ad0.nvs[2] == 0 && ad0.nls[3] == 0 && (!ad0.nl0 || (span(ad0.nl0s[2], 0) && ad0.nl0s[3] == 0));
    }
    // (below K ~ 12 the vector-unit tree kernel is as fast or faster: a 32 x 32 MFMA tile is mostly padding there)
    if (K > 12 && K <= 32 && !no_wave && fits32 && std::is_same<T, float>::value) {
        const size_t wsmem = (size_t)(25 * WAVE_TILE + 16 * 32) * sizeof(float);     // the largest launch (5 rounds)
        for (auto fn : {(const void *)chain_wave_kernel<0>, (const void *)chain_wave_kernel<1>, (const void *)chain_wave_kernel<2>,
                        (const void *)chain_wave_kernel<0, true, 256>, (const void *)chain_wave_kernel<1, true, 256>,
                        (const void *)chain_wave_kernel<2, true, 256>})
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsmem) != hipSuccess)
                return ALAN_ERR_LAUNCH;
        ChainAdd<float> adf, nonef;
        std::memcpy(&adf, &ad0, sizeof(adf));             // (T is float here)
        std::memset(&nonef, 0, sizeof(nonef));
        // rounds per launch: 2^(rounds - 1) waves per workgroup, so few rounds while there are many nodes (the first
        // round of a launch runs 2^(rounds - 3) products per SIMD) and all the rest in the last launch
        for (int r = 0; r < tl.L;) {
            int rounds = std::min(WAVE_ROUNDS, tl.L - r);
            if (tl.n[r] * B > 64 || (r == 0 && normal)) {
                rounds = std::min(3, tl.L - r);           // (the Normal term's instantiation is built for <= 256 threads)
            }
            if (r == 0 && normal) rounds = std::min(rounds, 3);
            const int nw = 1 << (rounds - 1);
            const size_t lsmem = (size_t)((nw + nw / 2 + 1) * WAVE_TILE + nw * 32) * sizeof(float);
            const int64_t nseg = (tl.n[r] + (1 << rounds) - 1) >> rounds;
            WaveOuts wo;
            std::memset(&wo, 0, sizeof(wo));
            for (int q = 0; q < rounds; ++q) wo.o[q] = (float *)((char *)ws + tl.off[r + 1 + q]), wo.n[q] = (int)tl.n[r + 1 + q];
            const bool last = r + rounds == tl.L;
            const int nadd = r == 0 ? n_more : 0;
            auto wk = nadd == 0 ? chain_wave_kernel<0> : nadd == 1 ? chain_wave_kernel<1> : chain_wave_kernel<2>;
            if (r == 0 && normal)
                wk = nadd == 0 ? chain_wave_kernel<0, true, 256> : nadd == 1 ? chain_wave_kernel<1, true, 256>
                                                                              : chain_wave_kernel<2, true, 256>;
            WaveFinal wf;
            std::memset(&wf, 0, sizeof(wf));
            if (last && final_) {
                wf.on = 1, wf.n_extra = final_->n_extra, wf.add_const = (float)final_->add_const;
                for (int f = 0; f < final_->n_extra; ++f)
                    wf.extra[f] = (const float *)final_->extra[f], wf.stride[f] = (int32_t)final_->stride[f];
                wf.out = (float *)final_->out;
                wf.ring_slots = (float *const *)final_->ring_slots, wf.ring_counter = (int32_t *)final_->ring_counter;
                wf.ring_n = final_->ring_n;
            }
            ALAN_LAUNCH(wk, dim3((uint32_t)nseg, (uint32_t)B), dim3(64 * nw), lsmem, stream,
                               (const float *)src, cB, cT, cR, cC, (int)tl.n[r], (int)K, rounds, wo,
                               last ? (float *)out_vec : (float *)nullptr, r == 0 ? adf : nonef, wf);
            if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
            r += rounds;
            src = (const T *)wo.o[rounds - 1];
            cB = tl.n[r] * K * K;
            cT = K * K;
            cR = K;
            cC = 1;
        }
    } else if (final_) {
        return ALAN_ERR_UNSUPPORTED;                      // (only the one-wave-per-product kernel carries the parent's contraction)
    } else if (K <= 32) {
        const size_t slot = (KP * (KP + 4) + KP * KP + 2 * KP) * sizeof(T);
        auto tk = chain_tree_kernel<T>;
        if (hipFuncSetAttribute((const void *)tk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * slot)) != hipSuccess)
            return ALAN_ERR_LAUNCH;
        for (int r = 0; r < tl.L;) {
            const int rounds = std::min(3, tl.L - r);
            const int64_t nseg = (tl.n[r] + (1 << rounds) - 1) >> rounds;
            T *o[3] = {nullptr, nullptr, nullptr};
            int nn[3] = {0, 0, 0};
            for (int q = 0; q < rounds; ++q) o[q] = (T *)((char *)ws + tl.off[r + 1 + q]), nn[q] = (int)tl.n[r + 1 + q];
            const bool last = r + rounds == tl.L;
            ALAN_LAUNCH(tk, dim3((uint32_t)nseg, (uint32_t)B), dim3(TREE_THREADS), 8 * slot, stream, src, cB, cT,
                               cR, cC, (int)tl.n[r], (int)K, rounds, o[0], o[1], o[2], nn[0], nn[1], nn[2],
                               last ? (T *)out_vec : (T *)nullptr, r == 0 ? ad0 : none);
            if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
            r += rounds;
            src = o[rounds - 1];
            cB = tl.n[r] * K * K;
            cT = K * K;
            cR = K;
            cC = 1;
        }
    } else {
        constexpr bool no_mfma = false;
        for (int r = 1; r <= tl.L; ++r) {
            T *dst = (T *)((char *)ws + tl.off[r]);
            int rc = no_mfma ? (int)ALAN_ERR_UNSUPPORTED
                             : launch_pair_mfma<T>(K, (uint32_t)tl.n[r], (uint32_t)B, stream, src, cB, cT, cR, cC,
                                                   (int)tl.n[r - 1], dst, r == tl.L ? (T *)out_vec : (T *)nullptr,
                                                   r == 1 ? ad0 : none);
            if (rc == ALAN_ERR_LAUNCH) return rc;
            if (rc == ALAN_ERR_UNSUPPORTED) {
                ALAN_LAUNCH(kern, dim3((uint32_t)tl.n[r], (uint32_t)B), dim3(CHAIN_THREADS), smem, stream, src, cB,
                                   cT, cR, cC, (int)tl.n[r - 1], 2, (int)K, dst, r == tl.L ? (T *)out_vec : (T *)nullptr,
                                   r == 1 ? ad0 : none);
                if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
            }
            src = dst;
            cB = tl.n[r] * K * K;
            cT = K * K;
            cR = K;
            cC = 1;
        }
    }
    if (out_chain)
        if (hipMemcpyAsync(out_chain, (char *)ws + tl.off[tl.L], (size_t)(B * K * K) * sizeof(T),
                           hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    return ALAN_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward, one round of the tree per launch, one workgroup per node of that round.  For a pair
//   R = log(Pe @ Ce + eps) + pm + cm,   Pe = exp(P - pm), Ce = exp(C - cm),  pm = rowmax(P), cm = colmax(C)
// and an upstream gradient G of R, autograd through utils.py:503-507 gives, with G' = G / (Pe @ Ce + eps):
//   dP = Pe * (G' @ Ce^T)  +  [P == pm] * eps * rowsum(G') / ties      (the second term is the path through amax,
//   dC = Ce * (Pe^T @ G')  +  [C == cm] * eps * colsum(G') / ties       which only matters on the eps floor)
// A leftover node passes its gradient through.  At the root G comes from the caller:
//   G = grad_chain  +  grad_vec[i] * exp(R[i,j] - vec[i])               (t.logsumexp(., -1) of logpq.py:139)
// Round 3: the three K x K x K products (Pe @ Ce, G' @ Ce^T, Pe^T @ G') are register-tiled -- a thread owns 4 x 4 blocks
// of a product and reads its operands from LDS four at a time (rows padded to a multiple of four, zero beyond K) -- where
// round 2 walked one output element per thread with two LDS reads per multiply-add (T = 1000, K = 100: 2.77 ms against
// 0.21 ms for the forward).
template <typename T, int BS>
__device__ __forceinline__ void chain_ldn(const T *p, T (&v)[BS]) {
    if constexpr (BS == 2) {
        if constexpr (sizeof(T) == 4) {
            const float2 x = *reinterpret_cast<const float2 *>(p);
            v[0] = x.x, v[1] = x.y;
        } else {
            const double2 x = *reinterpret_cast<const double2 *>(p);
            v[0] = x.x, v[1] = x.y;
        }
    } else if constexpr (sizeof(T) == 4) {
        const float4 x = *reinterpret_cast<const float4 *>(p);
        v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
    } else {
        const double2 x = *reinterpret_cast<const double2 *>(p), y = *reinterpret_cast<const double2 *>(p + 2);
        v[0] = x.x, v[1] = x.y, v[2] = y.x, v[3] = y.y;
    }
}

// BS: side of a thread's block -- 4, or 2 where K <= 48 would leave most of the 256 threads without a 4 x 4 block.
template <typename T, int BS = 4>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_pair_backward_kernel(
    const T *src, int64_t sB, int64_t sT, int64_t sRow, int64_t sCol, int n_src, int K,
    const T *G,                                           // [B][gridDim.x][K][K], or nullptr at the root
    const T *root, int64_t rB, int64_t rRow, int64_t rCol, const T *vec, const T *grad_vec, const T *grad_chain,
    T *dsrc) {                                            // [B][n_src][K][K]
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int K4 = (K + 3) & ~3, KS = K4 + 4;            // rows of K4 (zero beyond K) + 4: 16-byte aligned, banks spread
    T *Pe = reinterpret_cast<T *>(smem_raw);
    T *Ce = Pe + (size_t)K4 * KS;
    T *Gp = Ce + (size_t)K4 * KS;
    T *pm = Gp + (size_t)K4 * KS, *cm = pm + K4, *pw = cm + K4, *cw = pw + K4;
    const int tid = threadIdx.x, KK = K * K;
    const int64_t b = blockIdx.y, node = blockIdx.x;
    const int t0 = 2 * (int)node, t1 = t0 + 1;
    const T NINF = Num<T>::ninf();

    auto upstream = [&](int i, int j) -> T {
        if (G) return G[((b * gridDim.x + node) * K + i) * K + j];
        T g = grad_chain ? grad_chain[(b * K + i) * K + j] : T(0);
        if (grad_vec) {
            const T v = vec[b * K + i];
            if (v != NINF) g += grad_vec[b * K + i] * Num<T>::exp_acc(root[b * rB + i * rRow + j * rCol] - v);
        }
        return g;
    };
    T *dP = dsrc + (b * n_src + t0) * (int64_t)KK;
    if (t1 >= n_src) {                                    // leftover of this round (utils.py:488-495)
        for (int e = tid; e < KK; e += CHAIN_THREADS) dP[e] = upstream(e / K, e % K);
        return;
    }
    T *dC = dP + KK;
    const T *Pg = src + b * sB + (int64_t)t0 * sT, *Cg = src + b * sB + (int64_t)t1 * sT;
    for (int e = tid; e < K4 * K4; e += CHAIN_THREADS) {
        const int i = e / K4, j = e - i * K4;
        const bool in = i < K && j < K;
        Pe[i * KS + j] = in ? Pg[i * sRow + j * sCol] : NINF;       // (-inf: exp gives the zero padding)
        Ce[i * KS + j] = in ? Cg[i * sRow + j * sCol] : NINF;
        Gp[i * KS + j] = T(0);
    }
    __syncthreads();
    for (int i = tid; i < 2 * K; i += CHAIN_THREADS) {
        T mx = NINF;
        if (i < K) {
            for (int j = 0; j < K; ++j) mx = fmax(mx, Pe[i * KS + j]);
            pm[i] = mx;
        } else {
            for (int k = 0; k < K; ++k) mx = fmax(mx, Ce[k * KS + (i - K)]);
            cm[i - K] = mx;
        }
    }
    __syncthreads();
    for (int e = tid; e < K4 * K4; e += CHAIN_THREADS) {
        const int i = e / K4, j = e - i * K4;
        const bool in = i < K && j < K;
        Pe[i * KS + j] = in ? Num<T>::exp_acc(Pe[i * KS + j] - pm[i]) : T(0);
        Ce[i * KS + j] = in ? Num<T>::exp_acc(Ce[i * KS + j] - cm[j]) : T(0);
    }
    __syncthreads();
    const int NB = K4 / BS;                               // blocks along a side
    // G' = G / (Pe @ Ce + eps): block (i0.., j0..) += Pe[i0 + r][k .. k + 3] . Ce[k + q][j0 .. j0 + 3]
    for (int blk = tid; blk < NB * NB; blk += CHAIN_THREADS) {
        const int i0 = BS * (blk / NB), j0 = BS * (blk % NB);
        T acc[BS][BS];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) acc[r][c] = T(0);
        for (int k = 0; k < K4; k += BS) {
            T a[BS][BS], c4[BS][BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) chain_ldn<T, BS>(Pe + (i0 + r) * KS + k, a[r]);
#pragma unroll
            for (int q = 0; q < BS; ++q) chain_ldn<T, BS>(Ce + (k + q) * KS + j0, c4[q]);
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int q = 0; q < BS; ++q)
#pragma unroll
                    for (int c = 0; c < BS; ++c) acc[r][c] += a[r][q] * c4[q][c];
        }
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c)
                if (i0 + r < K && j0 + c < K) Gp[(i0 + r) * KS + j0 + c] = upstream(i0 + r, j0 + c) / (acc[r][c] + Num<T>::eps);
    }
    __syncthreads();
    for (int i = tid; i < 2 * K; i += CHAIN_THREADS) {    // the amax paths: eps * sum(G') shared among the maxima
        T sum = T(0), ties = T(0);
        if (i < K) {
            for (int j = 0; j < K; ++j) sum += Gp[i * KS + j], ties += Pe[i * KS + j] == T(1) ? T(1) : T(0);
            pw[i] = Num<T>::eps * sum / ties;
        } else {
            const int j = i - K;
            for (int k = 0; k < K; ++k) sum += Gp[k * KS + j], ties += Ce[k * KS + j] == T(1) ? T(1) : T(0);
            cw[j] = Num<T>::eps * sum / ties;
        }
    }
    __syncthreads();
    for (int blk = tid; blk < NB * NB; blk += CHAIN_THREADS) {
        const int i0 = BS * (blk / NB), k0 = BS * (blk % NB);
        // dP[i, k] = Pe[i, k] * sum_j G'[i, j] Ce[k, j]: rows of G' against rows of Ce, four j at a time
        T acc[BS][BS];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) acc[r][c] = T(0);
        for (int j = 0; j < K4; j += BS) {
            T g4[BS][BS], c4[BS][BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) chain_ldn<T, BS>(Gp + (i0 + r) * KS + j, g4[r]);
#pragma unroll
            for (int c = 0; c < BS; ++c) chain_ldn<T, BS>(Ce + (k0 + c) * KS + j, c4[c]);
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int c = 0; c < BS; ++c)
#pragma unroll
                    for (int q = 0; q < BS; ++q) acc[r][c] += g4[r][q] * c4[c][q];
        }
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c)
                if (i0 + r < K && k0 + c < K) {
                    const T pe = Pe[(i0 + r) * KS + k0 + c];
                    dP[(i0 + r) * K + k0 + c] = pe * acc[r][c] + (pe == T(1) ? pw[i0 + r] : T(0));
                }
        // dC[k, j] = Ce[k, j] * sum_r Pe[r, k] G'[r, j]  (block rows k = i0.., columns j = k0..): one row r at a time
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) acc[r][c] = T(0);
        for (int r = 0; r < K; ++r) {
            T p4[BS], g4[BS];
            chain_ldn<T, BS>(Pe + r * KS + i0, p4);
            chain_ldn<T, BS>(Gp + r * KS + k0, g4);
#pragma unroll
            for (int q = 0; q < BS; ++q)
#pragma unroll
                for (int c = 0; c < BS; ++c) acc[q][c] += p4[q] * g4[c];
        }
#pragma unroll
        for (int q = 0; q < BS; ++q)
#pragma unroll
            for (int c = 0; c < BS; ++c)
                if (i0 + q < K && k0 + c < K) {
                    const T ce = Ce[(i0 + q) * KS + k0 + c];
                    dC[(i0 + q) * K + k0 + c] = ce * acc[q][c] + (ce == T(1) ? cw[k0 + c] : T(0));
                }
    }
}

// The same backward for fp32 with its three K x K x K products on the matrix cores and its phases rebuilt around the
// launch's latencies (round 3; T = 1000: K = 30 was 10 launches of 17 us whatever the number of pairs -- conditional
// loads each followed by its own wait, row maxima walked by 60 of the 256 threads --, K = 100 10 launches of 142 us of
// vector-unit multiply-adds):
//   loads      every element of P, C and G of a thread's share requested before the first is used (indices clamped, no
//              branch around a load), eight rounds at a time;
//   maxima / sums of a row or column: one wave per row, lanes along it, a shuffle tree (all four waves busy);
//   products   v_mfma_f32_32x32x2_f32 on 32 x 32 output tiles, operands read from LDS as the instruction wants them -- A
//              one (row, k) per lane, B one (k, column) -- out of matrices with an ODD row stride, so a row walk and a
//              column walk are both conflict-free and no transposed copy is needed; tiles dealt to the waves; rows and
//              columns beyond K are clamped reads times a zero mask (no padded copies: three K x K matrices are 121 KB at
//              K = 100).  The eps-floor paths (amax ties) are kept exactly as in the vector kernel.
// NT: threads of the workgroup -- 1024 above K = 48 (three K x K matrices fill most of a CU's LDS: one workgroup per CU
// whatever its size, so sixteen waves share the tiles and keep four times the loads in flight), 256 below.
template <int NT>
__global__ __launch_bounds__(NT) void chain_pair_backward_mfma_kernel(
    const float *src, int64_t sB, int64_t sT, int64_t sRow, int64_t sCol, int n_src, int K,
    const float *G,                                       // [B][gridDim.x][K][K], or nullptr at the root
    const float *root, int64_t rB, int64_t rRow, int64_t rCol, const float *vec, const float *grad_vec,
    const float *grad_chain, float *dsrc) {               // [B][n_src][K][K]
    typedef float T;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int KS = K | 1;
    T *Pe = reinterpret_cast<T *>(smem_raw);
    T *Ce = Pe + (size_t)K * KS;
    T *Gp = Ce + (size_t)K * KS;
    T *pm = Gp + (size_t)K * KS, *cm = pm + K, *pw = cm + K, *cw = pw + K;
    const int tid = threadIdx.x, KK = K * K, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.y, node = blockIdx.x;
    const int t0 = 2 * (int)node, t1 = t0 + 1;
    const T NINF = Num<T>::ninf();
    auto upstream_root = [&](int i, int j) -> T {
        T g = grad_chain ? grad_chain[(b * K + i) * K + j] : T(0);
        if (grad_vec) {
            const T v = vec[b * K + i];
            if (v != NINF) g += grad_vec[b * K + i] * Num<T>::exp_acc(root[b * rB + i * rRow + j * rCol] - v);
        }
        return g;
    };
    T *dP = dsrc + (b * n_src + t0) * (int64_t)KK;
    const T *Gn = G ? G + (b * gridDim.x + node) * (int64_t)KK : nullptr;
    if (t1 >= n_src) {                                    // leftover of this round (utils.py:488-495)
        for (int e = tid; e < KK; e += NT) dP[e] = Gn ? Gn[e] : upstream_root(e / K, e % K);
        return;
    }
    T *dC = dP + KK;
    const T *Pg = src + b * sB + (int64_t)t0 * sT, *Cg = src + b * sB + (int64_t)t1 * sT;
    // (e / K without the integer-division sequence: a float estimate and one correction each way, exact for e < 2^22)
    const float invK = 1.f / (float)K;
    auto row_of = [&](int e) {
        int i = (int)((float)e * invK);
        i -= (i * K > e) ? 1 : 0;
        i += ((i + 1) * K <= e) ? 1 : 0;
        return i;
    };
    // ---- loads
    constexpr int CH = 8;
    for (int e0 = tid; e0 < KK; e0 += NT * CH) {
        T pv[CH], cv[CH], gv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (e0 - tid + NT * u >= KK) {     // (uniform: a whole round beyond the matrices)
                pv[u] = cv[u] = gv[u] = T(0);
                continue;
            }
            const int e = min(e0 + NT * u, KK - 1), i = row_of(e), j = e - i * K;
            pv[u] = Pg[i * sRow + j * sCol];
            cv[u] = Cg[i * sRow + j * sCol];
            gv[u] = Gn ? Gn[e] : T(0);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int e = e0 + NT * u;
            if (e < KK) {
                const int i = row_of(e), j = e - i * K;
                Pe[i * KS + j] = pv[u], Ce[i * KS + j] = cv[u];
                Gp[i * KS + j] = Gn ? gv[u] : upstream_root(i, j);         // (the root: one launch of one pair)
            }
        }
    }
    __syncthreads();
    // ---- row maxima of P, column maxima of C: a thread per row / column walks it -- the reads of a walk do not depend on
    // each other (they pipeline), and with the odd row stride a column walk is as conflict-free as a row walk
    for (int rc = tid; rc < 2 * K; rc += NT) {
        const bool is_row = rc < K;
        const int idx = is_row ? rc : rc - K;
        const T *p0 = is_row ? Pe + idx * KS : Ce + idx;
        const int st = is_row ? 1 : KS;
        T m0 = NINF, m1 = NINF, m2 = NINF, m3 = NINF;
        int x = 0;
        for (; x + 4 <= K; x += 4) {
            m0 = fmaxf(m0, p0[x * st]), m1 = fmaxf(m1, p0[(x + 1) * st]);
            m2 = fmaxf(m2, p0[(x + 2) * st]), m3 = fmaxf(m3, p0[(x + 3) * st]);
        }
        for (; x < K; ++x) m0 = fmaxf(m0, p0[x * st]);
        (is_row ? pm : cm)[idx] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    }
    __syncthreads();
    for (int e = tid; e < KK; e += NT) {
        const int i = row_of(e), j = e - i * K;
        Pe[i * KS + j] = Num<T>::exp_acc(Pe[i * KS + j] - pm[i]);
        Ce[i * KS + j] = Num<T>::exp_acc(Ce[i * KS + j] - cm[j]);
    }
    __syncthreads();
    // ---- 32 x 32 tiles on the matrix cores.  Lane (c = lane & 31, h = lane >> 5): step s of a tile takes A[row c][k = 2 s +
    // h] and B[k = 2 s + h][column c]; accumulator register r holds row (r & 3) + 8 (r >> 2) + 4 h of column c.
    const int c = lane & 31, h = lane >> 5, nt = (K + 31) >> 5;
    auto tile = [&](auto a_at, auto b_at, int i0, int j0) {
        chain_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ra = min(i0 + c, K - 1), cb = min(j0 + c, K - 1);
        const bool oka = i0 + c < K, okb = j0 + c < K;
        // four steps' operands read before their four matrix instructions (a step's reads depend on nothing: an
        // un-unrolled loop would wait out an LDS latency per step)
        constexpr int SU = 4;
        for (int k0 = 0; k0 < K; k0 += 2 * SU) {
            float av[SU], bv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int kk = k0 + 2 * u + h, k = min(kk, K - 1);
                const bool okk = kk < K;                  // (selects, not products with 0: 0 x inf would be NaN)
                av[u] = oka && okk ? a_at(ra, k) : 0.f;
                bv[u] = okb && okk ? b_at(k, cb) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        return acc;
    };
    // G' = G / (Pe @ Ce + eps), in place over the staged G
    for (int tk = wave; tk < nt * nt; tk += NT / 64) {
        const int i0 = 32 * (tk / nt), j0 = 32 * (tk % nt);
        const chain_f32x16 acc = tile([&](int i, int k) { return Pe[i * KS + k]; }, [&](int k, int j) { return Ce[k * KS + j]; }, i0, j0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = j0 + c;
            if (i < K && j < K) Gp[i * KS + j] = Gp[i * KS + j] / (acc[r] + Num<T>::eps);
        }
    }
    __syncthreads();
    // ---- the amax paths: eps * sum(G') shared among the maxima of a row of P / a column of C
    for (int rc = tid; rc < 2 * K; rc += NT) {
        const bool is_row = rc < K;
        const int idx = is_row ? rc : rc - K;
        const T *g0 = is_row ? Gp + idx * KS : Gp + idx, *e0 = is_row ? Pe + idx * KS : Ce + idx;
        const int st = is_row ? 1 : KS;
        T s0 = T(0), s1 = T(0), n0 = T(0), n1 = T(0);
        int x = 0;
        for (; x + 2 <= K; x += 2) {
            s0 += g0[x * st], s1 += g0[(x + 1) * st];
            n0 += e0[x * st] == T(1) ? T(1) : T(0), n1 += e0[(x + 1) * st] == T(1) ? T(1) : T(0);
        }
        for (; x < K; ++x) s0 += g0[x * st], n0 += e0[x * st] == T(1) ? T(1) : T(0);
        (is_row ? pw : cw)[idx] = Num<T>::eps * (s0 + s1) / (n0 + n1);
    }
    __syncthreads();
    // ---- dP = Pe * (G' @ Ce^T) and dC = Ce * (Pe^T @ G'): 2 nt^2 tiles dealt to the waves
    for (int tk = wave; tk < 2 * nt * nt; tk += NT / 64) {
        const bool second = tk >= nt * nt;
        const int t2 = second ? tk - nt * nt : tk, i0 = 32 * (t2 / nt), j0 = 32 * (t2 % nt);
        chain_f32x16 acc;
        if (!second)        // dP[i, k] : A[i][j] = G'[i][j], B[j][k] = Ce[k][j]
            acc = tile([&](int i, int j) { return Gp[i * KS + j]; }, [&](int j, int k) { return Ce[k * KS + j]; }, i0, j0);
        else                // dC[k, j] : A[k][r] = Pe[r][k], B[r][j] = G'[r][j]
            acc = tile([&](int k, int r) { return Pe[r * KS + k]; }, [&](int r, int j) { return Gp[r * KS + j]; }, i0, j0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = j0 + c;
            if (i < K && j < K) {
                if (!second) {
                    const T pe = Pe[i * KS + j];
                    dP[i * K + j] = pe * acc[r] + (pe == T(1) ? pw[i] : T(0));
                } else {
                    const T ce = Ce[i * KS + j];
                    dC[i * K + j] = ce * acc[r] + (ce == T(1) ? cw[j] : T(0));
                }
            }
        }
    }
}

// The whole backward in ONE launch (round 4; T = 1000: ten launches of 12 us at K = 30, of 53 us at K = 100 -- a launch per
// round of the tree, every node of a round waiting for the slowest of the round above).  A workgroup per node of the tree,
// the root first: a node needs its parent's result only for the second half of its work, so it stages its pair, takes the
// maxima and the exponentials and multiplies Pe @ Ce at once -- every node of the tree does that side by side -- and then
// waits for the gradient its parent leaves in memory.  No flags: a launch in front of this one fills the workspace with a
// NaN of a payload no arithmetic produces (BWD_UNSET); the parent stores its result element by element with agent-scope
// relaxed atomics (past the non-coherent caches; release / acquire FENCES write back and invalidate a whole L2 per
// workgroup here: 181 us for the T = 1000, K = 30 tree against 123 with a launch per round) and the child reads until an
// element is no longer BWD_UNSET -- one thread watches the first element, then every thread checks its own.  What is
// left on the path from the root to the leaves is, per round: one store-to-load hand-over, a division, two products.
// Progress: workgroups start in the order of their index on each XCD (round-robin over the XCDs) and a node's parent has a
// lower index, so the lowest-indexed unfinished node always holds a slot and never waits for anyone behind it; should that
// ever fail, a wait gives up after `timeout` ticks of the 100 MHz clock and the element counts as NaN (nothing hangs;
// every gradient below it is NaN).
constexpr uint32_t BWD_UNSET = 0x7fedcba9u;
constexpr int BWD_MAX_LEVELS = 32;
struct BwdLevel {
    const float *src;                 // the round's inputs: [B][n_src] matrices with strides (cB, cT, cR, cC)
    const float *G;                   // upstream gradients [B][n_nodes][K][K]; nullptr at the root
    float *dsrc;                      // [B][n_src][K][K]
    int64_t cB, cT, cR, cC;
    int32_t n_src, n_nodes, leaf, pad_;   // leaf: dsrc is the caller's grad_ms (nobody in this launch reads it)
};
struct BwdTree {
    uint32_t first[BWD_MAX_LEVELS];   // first workgroup of level l (0 = the root's round); 0xffffffff beyond the last
    BwdLevel lv[BWD_MAX_LEVELS];
    uint32_t total, timeout;
    int32_t K, pad_;
    const float *root;
    int64_t rB, rRow, rCol;
    const float *vec, *grad_vec, *grad_chain;
};

__global__ __launch_bounds__(256) void chain_unset_kernel(uint4 *ws, uint32_t n16) {
    const uint4 u = {BWD_UNSET, BWD_UNSET, BWD_UNSET, BWD_UNSET};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) ws[i] = u;
}

template <int NT>
__global__ __launch_bounds__(NT) void chain_tree_backward_kernel(const BwdTree a) {
    typedef float T;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int K = a.K, KS = K | 1;
    T *Pe = reinterpret_cast<T *>(smem_raw);
    T *Ce = Pe + (size_t)K * KS;
    T *Gp = Ce + (size_t)K * KS;
    T *pm = Gp + (size_t)K * KS, *cm = pm + K, *pw = cm + K, *cw = pw + K, *pn = cw + K, *cn = pn + K;    // (pn, cn: ties)
    const int tid = threadIdx.x, KK = K * K, lane = tid & 63, wave = tid >> 6;
    const uint32_t vb = blockIdx.x;
    const int64_t b = blockIdx.y;
    int li = 0;                                           // (workgroup-uniform; constant offsets: wide scalar loads, no branch)
#pragma unroll
    for (int i = 1; i < BWD_MAX_LEVELS; ++i) li += vb >= a.first[i] ? 1 : 0;
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    const char *base = (const char *)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr());
    const BwdLevel &lv = *reinterpret_cast<const BwdLevel *>(base + offsetof(BwdTree, lv) + (size_t)li * sizeof(BwdLevel));
    const uint32_t first_here = *reinterpret_cast<const uint32_t *>(base + offsetof(BwdTree, first) + (size_t)li * 4);
    const int node = (int)(vb - first_here), n_src = lv.n_src;
    const int t0 = 2 * node, t1 = t0 + 1;
    const T NINF = Num<T>::ninf();
    const T *Gn = lv.G ? lv.G + (b * lv.n_nodes + node) * (int64_t)KK : nullptr;
    T *dP = lv.dsrc + (b * n_src + t0) * (int64_t)KK;
    T *dC = dP + KK;
    const bool pair = t1 < n_src;
    auto upstream_root = [&](int i, int j) -> T {
        T g = a.grad_chain ? a.grad_chain[(b * K + i) * K + j] : T(0);
        if (a.grad_vec) {
            const T v = a.vec[b * K + i];
            if (v != NINF) g += a.grad_vec[b * K + i] * Num<T>::exp_acc(a.root[b * a.rB + i * a.rRow + j * a.rCol] - v);
        }
        return g;
    };
    // what a node hands to its children: agent-scope relaxed atomics, element by element (the leaves' output: plain stores)
    const bool leaf_out = lv.leaf != 0;
    auto put = [&](T *p, T v) {
        if (leaf_out)
            *p = v;
        else
            __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the parent's gradient: one thread watches its first element (a node's elements arrive within a microsecond of each other;
    // hundreds of waiting workgroups polling all of theirs would flood the fabric) ...
    auto wait_parent = [&]() {
        if (Gn && tid == 0) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            while (__float_as_uint(__hip_atomic_load(Gn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == BWD_UNSET) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t_start > a.timeout) break;
            }
        }
        __syncthreads();
    };
    // ... then every thread reads its own until they are there
    auto get = [&](const T *p) {
        T v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__float_as_uint(v) == BWD_UNSET) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            do {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while (__float_as_uint(v) == BWD_UNSET && __builtin_amdgcn_s_memrealtime() - t_start <= a.timeout);
            if (__float_as_uint(v) == BWD_UNSET) v = __builtin_nanf("");
        }
        return v;
    };
    const float invK = 1.f / (float)K;
    auto row_of = [&](int e) {
        int i = (int)((float)e * invK);
        i -= (i * K > e) ? 1 : 0;
        i += ((i + 1) * K <= e) ? 1 : 0;
        return i;
    };
    if (!pair) {                                          // leftover of this round (utils.py:488-495): the gradient passes through
        wait_parent();
        for (int e = tid; e < KK; e += NT) put(dP + e, Gn ? get(Gn + e) : upstream_root(row_of(e), e - row_of(e) * K));
        return;
    }
    const T *Pg = lv.src + b * lv.cB + (int64_t)t0 * lv.cT, *Cg = lv.src + b * lv.cB + (int64_t)t1 * lv.cT;
    const int64_t sRow = lv.cR, sCol = lv.cC;
    // ---- before the parent is needed: the pair staged, maxima, exponentials, Pe @ Ce + eps (left in Gp)
    constexpr int CH = 8;
    for (int e0 = tid; e0 < KK; e0 += NT * CH) {
        T pv[CH], cv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (e0 - tid + NT * u >= KK) {     // (uniform: a whole round beyond the matrices)
                pv[u] = cv[u] = T(0);
                continue;
            }
            const int e = min(e0 + NT * u, KK - 1), i = row_of(e), j = e - i * K;
            pv[u] = Pg[i * sRow + j * sCol];
            cv[u] = Cg[i * sRow + j * sCol];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int e = e0 + NT * u;
            if (e < KK) {
                const int i = row_of(e), j = e - i * K;
                Pe[i * KS + j] = pv[u], Ce[i * KS + j] = cv[u];
            }
        }
    }
    __syncthreads();
    for (int rc = tid; rc < 2 * K; rc += NT) {
        const bool is_row = rc < K;
        const int idx = is_row ? rc : rc - K;
        const T *p0 = is_row ? Pe + idx * KS : Ce + idx;
        const int st = is_row ? 1 : KS;
        T m0 = NINF, m1 = NINF, m2 = NINF, m3 = NINF;
        int x = 0;
        for (; x + 4 <= K; x += 4) {
            m0 = fmaxf(m0, p0[x * st]), m1 = fmaxf(m1, p0[(x + 1) * st]);
            m2 = fmaxf(m2, p0[(x + 2) * st]), m3 = fmaxf(m3, p0[(x + 3) * st]);
        }
        for (; x < K; ++x) m0 = fmaxf(m0, p0[x * st]);
        (is_row ? pm : cm)[idx] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    }
    __syncthreads();
    for (int e = tid; e < KK; e += NT) {
        const int i = row_of(e), j = e - i * K;
        Pe[i * KS + j] = Num<T>::exp_acc(Pe[i * KS + j] - pm[i]);
        Ce[i * KS + j] = Num<T>::exp_acc(Ce[i * KS + j] - cm[j]);
    }
    __syncthreads();
    // rows of P / columns of C in four segments, a thread each (lanes 4 q .. 4 q + 3 of a wave: two shuffles add them up)
    auto walk4 = [&](auto f) {
        for (int q0 = 0; q0 < 2 * K; q0 += NT / 4) {      // (uniform trip count: the shuffles below need whole groups)
            const int rc = q0 + (tid >> 2), seg = tid & 3;
            const bool on = rc < 2 * K;
            const bool is_row = rc < K;
            const int idx = on ? (is_row ? rc : rc - K) : 0;
            const int per = (K + 3) >> 2, x0 = seg * per, x1 = min(K, x0 + per);
            float v = f(is_row, idx, x0, x1);
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (on && seg == 0) f(is_row, idx, v);
        }
    };
    struct Ties {
        const T *Pe, *Ce;
        T *pn, *cn;
        int KS;
        __device__ float operator()(bool is_row, int idx, int x0, int x1) const {
            const T *e0 = is_row ? Pe + idx * KS : Ce + idx;
            const int st = is_row ? 1 : KS;
            float n = 0.f;
            for (int x = x0; x < x1; ++x) n += e0[x * st] == T(1) ? 1.f : 0.f;
            return n;
        }
        __device__ void operator()(bool is_row, int idx, float v) const { (is_row ? pn : cn)[idx] = v; }
    };
    walk4(Ties{Pe, Ce, pn, cn, KS});
    const int c = lane & 31, h = lane >> 5, nt = (K + 31) >> 5;
    auto tile = [&](auto a_at, auto b_at, int i0, int j0) {
        chain_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ra = min(i0 + c, K - 1), cb = min(j0 + c, K - 1);
        const bool oka = i0 + c < K, okb = j0 + c < K;
        constexpr int SU = 8;
        for (int k0 = 0; k0 < K; k0 += 2 * SU) {
            float av[SU], bv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int kk = k0 + 2 * u + h, k = min(kk, K - 1);
                const bool okk = kk < K;                  // (selects, not products with 0: 0 x inf would be NaN)
                av[u] = oka && okk ? a_at(ra, k) : 0.f;
                bv[u] = okb && okk ? b_at(k, cb) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        return acc;
    };
    for (int tk = wave; tk < nt * nt; tk += NT / 64) {
        const int i0 = 32 * (tk / nt), j0 = 32 * (tk % nt);
        const chain_f32x16 acc = tile([&](int i, int k) { return Pe[i * KS + k]; }, [&](int k, int j) { return Ce[k * KS + j]; }, i0, j0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = j0 + c;
            if (i < K && j < K) Gp[i * KS + j] = acc[r] + Num<T>::eps;
        }
    }
    // ---- the parent's gradient: G' = G / (Pe @ Ce + eps)
    wait_parent();                                        // (its barrier also closes the products' writes to Gp)
    for (int e0 = tid; e0 < KK; e0 += NT * CH) {
        T gv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int e = min(e0 + NT * u, KK - 1);       // (every load requested before the first is looked at)
            gv[u] = Gn ? __hip_atomic_load(Gn + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int e = e0 + NT * u;
            if (e < KK) {
                const int i = row_of(e), j = e - i * K;
                if (Gn && __float_as_uint(gv[u]) == BWD_UNSET) gv[u] = get(Gn + e);
                Gp[i * KS + j] = (Gn ? gv[u] : upstream_root(i, j)) / Gp[i * KS + j];
            }
        }
    }
    __syncthreads();
    // ---- the amax paths: eps * sum(G') shared among the maxima of a row of P / a column of C (their number: counted above)
    struct Sums {
        const T *Gp;
        T *pw, *cw;
        const T *pn, *cn;
        int KS;
        __device__ float operator()(bool is_row, int idx, int x0, int x1) const {
            const T *g0 = is_row ? Gp + idx * KS : Gp + idx;
            const int st = is_row ? 1 : KS;
            float s = 0.f;
            for (int x = x0; x < x1; ++x) s += g0[x * st];
            return s;
        }
        __device__ void operator()(bool is_row, int idx, float v) const {
            (is_row ? pw : cw)[idx] = Num<T>::eps * v / (is_row ? pn : cn)[idx];
        }
    };
    walk4(Sums{Gp, pw, cw, pn, cn, KS});
    __syncthreads();
    // ---- dP = Pe * (G' @ Ce^T) and dC = Ce * (Pe^T @ G'): 2 nt^2 tiles dealt to the waves
    for (int tk = wave; tk < 2 * nt * nt; tk += NT / 64) {
        const bool second = tk >= nt * nt;
        const int t2 = second ? tk - nt * nt : tk, i0 = 32 * (t2 / nt), j0 = 32 * (t2 % nt);
        chain_f32x16 acc;
        if (!second)        // dP[i, k] : A[i][j] = G'[i][j], B[j][k] = Ce[k][j]
            acc = tile([&](int i, int j) { return Gp[i * KS + j]; }, [&](int j, int k) { return Ce[k * KS + j]; }, i0, j0);
        else                // dC[k, j] : A[k][r] = Pe[r][k], B[r][j] = G'[r][j]
            acc = tile([&](int k, int r) { return Pe[r * KS + k]; }, [&](int r, int j) { return Gp[r * KS + j]; }, i0, j0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h, j = j0 + c;
            if (i < K && j < K) {
                if (!second) {
                    const T pe = Pe[i * KS + j];
                    put(dP + i * K + j, pe * acc[r] + (pe == T(1) ? pw[i] : T(0)));
                } else {
                    const T ce = Ce[i * KS + j];
                    put(dC + i * K + j, ce * acc[r] + (ce == T(1) ? cw[j] : T(0)));
                }
            }
        }
    }
}

template <typename T>
static int chain_backward_run(const void *ms, int64_t B, int64_t Tn, int64_t K, int64_t sB, int64_t sT, int64_t sRow,
                              int64_t sCol, const void *tree, const void *out_vec, const void *grad_vec,
                              const void *grad_chain, void *grad_ms, void *ws, size_t ws_bytes, hipStream_t stream) {
    const size_t K4 = (size_t)((K + 3) & ~3), KS = K4 + 4;
    const size_t smem = (3 * K4 * KS + 4 * K4) * sizeof(T);
    if (smem > 160 * 1024) return ALAN_ERR_UNSUPPORTED;
    const TreeLayout tl = tree_layout(B, Tn, K, sizeof(T));
    if (tl.L > 1 && (!ws || ws_bytes < tl.bytes)) return ALAN_ERR_WORKSPACE;
    auto kern = K <= 48 ? chain_pair_backward_kernel<T, 2> : chain_pair_backward_kernel<T, 4>;
    static const int bwd_mfma_knob = env_knob("ALAN_CHAIN_BWD_MFMA");                 // ablation knob: 0 = the vector kernel
    bool mfma = false;
    size_t smem_m = 0;
    if constexpr (sizeof(T) == 4) {
        smem_m = (3 * (size_t)K * (size_t)(K | 1) + 6 * (size_t)K) * sizeof(float);
        mfma = bwd_mfma_knob != 0 && K >= 2 && smem_m <= 160 * 1024;
        if (mfma && smem_m > 64 * 1024 &&
            hipFuncSetAttribute((const void *)chain_pair_backward_mfma_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_m) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    }
    if (!mfma && smem > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    const T *root = (const T *)((const char *)tree + tl.off[tl.L]);
    if constexpr (sizeof(T) == 4) {
        static const int tree_knob = env_knob("ALAN_CHAIN_BWD_TREE");                 // ablation knob: 0 = a launch per round
        if (mfma && tree_knob != 0 && tl.L >= 2 && tl.L <= BWD_MAX_LEVELS) {
            // one launch for the whole tree (chain_tree_backward_kernel) behind one that marks the workspace as not yet written
            BwdTree a;
            std::memset(&a, 0, sizeof(a));
            uint32_t total = 0;
            for (int li = 0; li < BWD_MAX_LEVELS; ++li) a.first[li] = 0xffffffffu;
            for (int li = 0; li < tl.L; ++li) {
                const int r = tl.L - li;
                const bool top = r == tl.L, bottom = r == 1;
                BwdLevel &lv = a.lv[li];
                lv.src = bottom ? (const float *)ms : (const float *)((const char *)tree + tl.off[r - 1]);
                lv.cB = bottom ? sB : tl.n[r - 1] * K * K, lv.cT = bottom ? sT : K * K, lv.cR = bottom ? sRow : K,
                lv.cC = bottom ? sCol : 1;
                lv.G = top ? nullptr : (const float *)((const char *)ws + tl.off[r]);
                lv.dsrc = bottom ? (float *)grad_ms : (float *)((char *)ws + tl.off[r - 1]);
                lv.n_src = (int32_t)tl.n[r - 1], lv.n_nodes = (int32_t)tl.n[r], lv.leaf = bottom ? 1 : 0;
                a.first[li] = total;
                total += (uint32_t)tl.n[r];
            }
            if (!ws || ws_bytes < tl.bytes) return ALAN_ERR_WORKSPACE;
            a.total = total, a.timeout = 20000000u;                                    // 0.2 s of the 100 MHz clock
            a.K = (int32_t)K;
            a.root = (const float *)root, a.rB = K * K, a.rRow = K, a.rCol = 1;
            a.vec = (const float *)out_vec, a.grad_vec = (const float *)grad_vec, a.grad_chain = (const float *)grad_chain;
            const uint32_t n16 = (uint32_t)(tl.bytes / 16);                             // (rounds are 256-byte aligned)
            ALAN_LAUNCH(chain_unset_kernel, dim3(std::min<uint32_t>((n16 + 255) / 256, 2048u)), dim3(256), 0, stream, (uint4 *)ws, n16);
            if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
            if (K > 48) {
                if (smem_m > 64 * 1024 &&
                    hipFuncSetAttribute((const void *)chain_tree_backward_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)smem_m) != hipSuccess)
                    return ALAN_ERR_LAUNCH;
                ALAN_LAUNCH(chain_tree_backward_kernel<1024>, dim3(total, (uint32_t)B), dim3(1024), smem_m, stream, a);
            } else {
                ALAN_LAUNCH(chain_tree_backward_kernel<256>, dim3(total, (uint32_t)B), dim3(256), smem_m, stream, a);
            }
            return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
        }
    }
    for (int r = tl.L; r >= 1; --r) {
        const bool top = r == tl.L, bottom = r == 1;
        const T *src = bottom ? (const T *)ms : (const T *)((const char *)tree + tl.off[r - 1]);
        const int64_t cB = bottom ? sB : tl.n[r - 1] * K * K, cT = bottom ? sT : K * K, cR = bottom ? sRow : K,
                      cC = bottom ? sCol : 1;
        const T *G = top ? (const T *)nullptr : (const T *)((const char *)ws + tl.off[r]);
        T *dsrc = bottom ? (T *)grad_ms : (T *)((char *)ws + tl.off[r - 1]);
        if constexpr (sizeof(T) == 4) {
            if (mfma) {
                if (K > 48)
                    ALAN_LAUNCH(chain_pair_backward_mfma_kernel<1024>, dim3((uint32_t)tl.n[r], (uint32_t)B), dim3(1024),
                                       smem_m, stream, src, cB, cT, cR, cC, (int)tl.n[r - 1], (int)K, G, root, (int64_t)(K * K),
                                       (int64_t)K, (int64_t)1, (const T *)out_vec, (const T *)grad_vec, (const T *)grad_chain, dsrc);
                else
                    ALAN_LAUNCH(chain_pair_backward_mfma_kernel<256>, dim3((uint32_t)tl.n[r], (uint32_t)B), dim3(256),
                                       smem_m, stream, src, cB, cT, cR, cC, (int)tl.n[r - 1], (int)K, G, root, (int64_t)(K * K),
                                       (int64_t)K, (int64_t)1, (const T *)out_vec, (const T *)grad_vec, (const T *)grad_chain, dsrc);
                if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
                continue;
            }
        }
        ALAN_LAUNCH(kern, dim3((uint32_t)tl.n[r], (uint32_t)B), dim3(CHAIN_THREADS), smem, stream, src, cB, cT,
                           cR, cC, (int)tl.n[r - 1], (int)K, G, root, (int64_t)(K * K), (int64_t)K, (int64_t)1,
                           (const T *)out_vec, (const T *)grad_vec, (const T *)grad_chain, dsrc);
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    }
    return ALAN_OK;
}

}  // namespace alan

using namespace alan;

static size_t elt_of(int32_t dtype) { return dtype == ALAN_F64 ? 8 : 4; }

extern "C" size_t alan_chain_batched_workspace_bytes(int64_t B, int64_t T, int64_t K, int32_t dtype) {
    if (B < 1 || T < 1 || K < 1) return 0;
    return tree_layout(B, T, K, elt_of(dtype)).bytes;
}

extern "C" size_t alan_chain_backward_batched_workspace_bytes(int64_t B, int64_t T, int64_t K, int32_t dtype) {
    return alan_chain_batched_workspace_bytes(B, T, K, dtype);     // one gradient per tree node
}

extern "C" int alan_chain_logmmexp_batched(const void *ms, int32_t dtype, int64_t B, int64_t T, int64_t K, int64_t sB,
                                           int64_t sT, int64_t sRow, int64_t sCol, void *out_chain, void *out_vec,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    if (!ms || B < 1 || T < 1 || K < 1 || (!out_chain && !out_vec)) return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31) || B > 65535) return ALAN_ERR_UNSUPPORTED;      // B rides on gridDim.y
    if (dtype == ALAN_F32)
        return chain_run<float>(ms, B, T, K, sB, sT, sRow, sCol, out_chain, out_vec, workspace, workspace_bytes,
                                (hipStream_t)stream);
    if (dtype == ALAN_F64)
        return chain_run<double>(ms, B, T, K, sB, sT, sRow, sCol, out_chain, out_vec, workspace, workspace_bytes,
                                 (hipStream_t)stream);
    return ALAN_ERR_BAD_DESC;
}

extern "C" int alan_chain_logmmexp_terms_final(const void *const *terms, const int64_t *strides, int32_t n_terms,
                                               const alan_chain_normal_t *normal, const alan_chain_final_t *fin,
                                               int32_t dtype, int64_t B, int64_t T, int64_t K, void *out_chain, void *out_vec,
                                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!terms || !strides || n_terms < 1 || n_terms > 3 || !terms[0] || B < 1 || T < 1 || K < 1 ||
        (!out_chain && !out_vec))
        return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31) || B > 65535) return ALAN_ERR_UNSUPPORTED;
    if (fin) {
        if (fin->n_extra < 0 || fin->n_extra > 3 || !out_vec || (!fin->out && !fin->ring_n)) return ALAN_ERR_BAD_DESC;
        if (fin->ring_n && (fin->ring_n < 0 || !fin->ring_slots || !fin->ring_counter)) return ALAN_ERR_BAD_DESC;
        for (int f = 0; f < fin->n_extra; ++f) {
            if (!fin->extra[f]) return ALAN_ERR_BAD_DESC;
            if (fin->stride[f] < 0 || fin->stride[f] * K >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
        }
        // one chain, the one-wave-per-product kernel's sizes (chain_run declines the rest)
        if (B != 1 || dtype != ALAN_F32 || K <= 12 || K > 32) return ALAN_ERR_UNSUPPORTED;
    }
    if (dtype == ALAN_F32)
        return chain_run<float>(terms[0], B, T, K, strides[0], strides[1], strides[2], strides[3], out_chain, out_vec,
                                workspace, workspace_bytes, (hipStream_t)stream, terms + 1, strides + 4, n_terms - 1, normal, fin);
    if (dtype == ALAN_F64)
        return chain_run<double>(terms[0], B, T, K, strides[0], strides[1], strides[2], strides[3], out_chain, out_vec,
                                 workspace, workspace_bytes, (hipStream_t)stream, terms + 1, strides + 4, n_terms - 1, normal);
    return ALAN_ERR_BAD_DESC;
}

extern "C" int alan_chain_logmmexp_backward_batched(const void *ms, int32_t dtype, int64_t B, int64_t T, int64_t K,
                                                    int64_t sB, int64_t sT, int64_t sRow, int64_t sCol,
                                                    const void *tree, const void *out_vec, const void *grad_vec,
                                                    const void *grad_chain, void *grad_ms, void *workspace,
                                                    size_t workspace_bytes, void *stream) {
    if (!ms || !tree || !grad_ms || B < 1 || T < 1 || K < 1) return ALAN_ERR_BAD_DESC;
    if ((!grad_vec && !grad_chain) || (grad_vec && !out_vec)) return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31) || B > 65535) return ALAN_ERR_UNSUPPORTED;
    if (dtype == ALAN_F32)
        return chain_backward_run<float>(ms, B, T, K, sB, sT, sRow, sCol, tree, out_vec, grad_vec, grad_chain, grad_ms,
                                         workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == ALAN_F64)
        return chain_backward_run<double>(ms, B, T, K, sB, sT, sRow, sCol, tree, out_vec, grad_vec, grad_chain,
                                          grad_ms, workspace, workspace_bytes, (hipStream_t)stream);
    return ALAN_ERR_BAD_DESC;
}

