// Posterior sampling of a timeseries variable's K index at every timestep (role of sample_Ks_timeseries,
// reduce_Ks.py:85-232, which evaluates chain_logmmexp on every prefix: O(T^2) chains) as backward messages + forward
// sampling, O(T K^2), in TWO launches whatever T is:
//
//   alan_chain_messages    beta[c, T, :] = 0,  beta[c, t, a] = LSE_b( ms[c, t, a, b] + beta[c, t+1, b] )   t = T-1 .. 1
//   alan_chain_sample      k_t ~ softmax_b( ms[c, t, k_{t-1}, b] + beta[c, t+1, b] ),  t = 0 .. T-1,  k_{-1} = init
//
// and, for parity checks against what the reference actually draws from, the forward (filtering) recursion
//
//   alan_chain_filter      alpha[c, 0, n, :] = ms[c, 0, init[n], :],  alpha[c, t, n, b] = LSE_a( alpha[c, t-1, n, a] + ms[c, t, a, b] )
//
// ms[c, t, a, b] = log weight of going from particle a of step t-1 to particle b of step t (the [T, K_init, K] factor
// of logpq.py:133, one per chain c of a batch).  Plain stable log-sum-exp (no eps): these are this library's own
// recursions, not restatements of reference arithmetic.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace alan {

constexpr int PST_THREADS = 256;

// One workgroup per chain; the matrix of step t is staged in LDS with coalesced loads while step t+1 is reduced.
// NQ: matrix elements per thread and step (compile-time: the prefetch registers are indexed statically).
template <int NQ>
__global__ __launch_bounds__(PST_THREADS) void chain_messages_kernel(const float *ms, int64_t sC, int64_t sT, int64_t sR,
                                                                     int64_t sCol, int T, int K, float *beta) {
    extern __shared__ __align__(16) float lds[];
    const int KS = K | 1;                                  // odd row stride: a thread walks its own row
    float *M = lds;                                        // [K][KS]
    float *bv = lds + K * KS;                              // [K] beta of step t+1
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    ms += c * sC;
    beta += c * (int64_t)(T + 1) * K;
    for (int i = tid; i < K; i += PST_THREADS) {
        bv[i] = 0.f;
        beta[(int64_t)T * K + i] = 0.f;
    }
    const int KK = K * K;
    float reg[NQ];
    auto fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = min(tid + q * PST_THREADS, KK - 1);          // (clamped: no branches among the loads)
            reg[q] = ms[(int64_t)t * sT + (e / K) * sR + (e % K) * sCol];
        }
    };
    if (T > 1) fetch(T - 1);
    for (int t = T - 1; t >= 1; --t) {
        __syncthreads();                                   // bv written, M free
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * PST_THREADS;
            if (e < KK) M[(e / K) * KS + (e % K)] = reg[q];
        }
        if (t > 1) fetch(t - 1);
        __syncthreads();
        float out = 0.f;
        if (tid < K) {
            float mx = -__builtin_huge_valf();
            for (int b = 0; b < K; ++b) mx = fmaxf(mx, M[tid * KS + b] + bv[b]);
            const float mref = (mx == -__builtin_huge_valf() || mx == __builtin_huge_valf()) ? 0.f : mx;
            float s = 0.f;
            for (int b = 0; b < K; ++b) s += __expf(M[tid * KS + b] + bv[b] - mref);
            out = logf(s) + mref;
            beta[(int64_t)t * K + tid] = out;
        }
        __syncthreads();                                   // everyone has read bv
        if (tid < K) bv[tid] = out;
    }
}

// One thread per (sample n, plate element b): chain c = n * cN + b * cB.  u holds one uniform per draw.
__global__ __launch_bounds__(PST_THREADS) void chain_sample_kernel(const float *ms, int64_t sC, int64_t sT, int64_t sR,
                                                                   int64_t sCol, const float *beta, int T, int K,
                                                                   const int64_t *init, int64_t iN, int64_t iB,
                                                                   const float *u, int64_t N, int64_t B, int64_t cN,
                                                                   int64_t cB, int64_t *out) {
    const int64_t idx = (int64_t)blockIdx.x * PST_THREADS + threadIdx.x;
    if (idx >= N * B) return;
    const int64_t n = idx / B, b = idx - n * B;
    const int64_t c = n * cN + b * cB;
    const float *m = ms + c * sC;
    const float *be = beta + c * (int64_t)(T + 1) * K;
    int prev = (int)init[n * iN + b * iB];
    for (int t = 0; t < T; ++t) {
        const float *row = m + (int64_t)t * sT + (int64_t)prev * sR;
        const float *bt = be + (int64_t)(t + 1) * K;
        float mx = -__builtin_huge_valf();
        for (int j = 0; j < K; ++j) mx = fmaxf(mx, row[j * sCol] + bt[j]);
        float tot = 0.f;
        for (int j = 0; j < K; ++j) tot += __expf(row[j * sCol] + bt[j] - mx);
        const float target = u[(n * B + b) * (int64_t)T + t] * tot;
        float acc = 0.f;
        int pick = K - 1;
        for (int j = 0; j < K; ++j) {
            acc += __expf(row[j * sCol] + bt[j] - mx);
            if (acc > target) {
                pick = j;
                break;
            }
        }
        out[(n * B + b) * (int64_t)T + t] = pick;
        prev = pick;
    }
}

// Filtering recursion, one workgroup per (chain, sample): alpha[c][t][n][:]
__global__ __launch_bounds__(PST_THREADS) void chain_filter_kernel(const float *ms, int64_t sC, int64_t sT, int64_t sR,
                                                                   int64_t sCol, int T, int K, const int64_t *init,
                                                                   int N, float *alpha) {
    extern __shared__ __align__(16) float lds[];
    float *prev = lds, *cur = lds + K;
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x, n = blockIdx.y;
    ms += c * sC;
    const int i0 = (int)init[n];
    for (int b = tid; b < K; b += PST_THREADS) {
        const float v = ms[(int64_t)i0 * sR + b * sCol];
        prev[b] = v;
        alpha[(((int64_t)c * T + 0) * N + n) * K + b] = v;
    }
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        for (int b = tid; b < K; b += PST_THREADS) {
            float mx = -__builtin_huge_valf();
            for (int a = 0; a < K; ++a) mx = fmaxf(mx, prev[a] + ms[(int64_t)t * sT + a * sR + b * sCol]);
            const float mref = (mx == -__builtin_huge_valf() || mx == __builtin_huge_valf()) ? 0.f : mx;
            float s = 0.f;
            for (int a = 0; a < K; ++a) s += __expf(prev[a] + ms[(int64_t)t * sT + a * sR + b * sCol] - mref);
            const float v = logf(s) + mref;
            cur[b] = v;
            alpha[(((int64_t)c * T + t) * N + n) * K + b] = v;
        }
        __syncthreads();
        for (int b = tid; b < K; b += PST_THREADS) prev[b] = cur[b];
        __syncthreads();
    }
}

}  // namespace alan

using namespace alan;

extern "C" int alan_chain_messages(const void *ms, int64_t C, int64_t T, int64_t K, int64_t sC, int64_t sT,
                                   int64_t sRow, int64_t sCol, void *beta, void *stream) {
    if (!ms || !beta || C < 1 || T < 1 || K < 1) return ALAN_ERR_BAD_DESC;
    if (K > 128 || T >= (1ll << 31) || C >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    const size_t smem = (size_t)(K * (K | 1) + K) * sizeof(float);
    auto go = [&](auto kern) {
        if (smem > 64 * 1024)
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return (int)ALAN_ERR_LAUNCH;
        ALAN_LAUNCH(kern, dim3((uint32_t)C), dim3(PST_THREADS), smem, (hipStream_t)stream, (const float *)ms, sC, sT,
                           sRow, sCol, (int)T, (int)K, (float *)beta);
        return hipGetLastError() == hipSuccess ? (int)ALAN_OK : (int)ALAN_ERR_LAUNCH;
    };
    return K <= 32 ? go(chain_messages_kernel<4>) : K <= 64 ? go(chain_messages_kernel<16>) : go(chain_messages_kernel<64>);
}

extern "C" int alan_chain_sample(const void *ms, int64_t T, int64_t K, int64_t sC, int64_t sT, int64_t sRow,
                                 int64_t sCol, const void *beta, const void *init, int64_t iN, int64_t iB,
                                 const void *uniforms, int64_t N, int64_t B, int64_t cN, int64_t cB, void *out,
                                 void *stream) {
    if (!ms || !beta || !init || !uniforms || !out || T < 1 || K < 1 || N < 1 || B < 1) return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31) || N * B >= (1ll << 40)) return ALAN_ERR_UNSUPPORTED;
    const int64_t blocks = (N * B + PST_THREADS - 1) / PST_THREADS;
    if (blocks >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    ALAN_LAUNCH(chain_sample_kernel, dim3((uint32_t)blocks), dim3(PST_THREADS), 0, (hipStream_t)stream,
                       (const float *)ms, sC, sT, sRow, sCol, (const float *)beta, (int)T, (int)K, (const int64_t *)init,
                       iN, iB, (const float *)uniforms, N, B, cN, cB, (int64_t *)out);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

extern "C" int alan_chain_filter(const void *ms, int64_t C, int64_t T, int64_t K, int64_t sC, int64_t sT, int64_t sRow,
                                 int64_t sCol, const void *init, int64_t N, void *alpha, void *stream) {
    if (!ms || !init || !alpha || C < 1 || T < 1 || K < 1 || N < 1) return ALAN_ERR_BAD_DESC;
    if (K > 4096 || C >= (1ll << 31) || N > 65535) return ALAN_ERR_UNSUPPORTED;
    ALAN_LAUNCH(chain_filter_kernel, dim3((uint32_t)C, (uint32_t)N), dim3(PST_THREADS), 2 * K * sizeof(float),
                       (hipStream_t)stream, (const float *)ms, sC, sT, sRow, sCol, (int)T, (int)K, (const int64_t *)init,
                       (int)N, (float *)alpha);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}
