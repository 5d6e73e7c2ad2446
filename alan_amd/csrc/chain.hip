// alan_chain_logmmexp: the timeseries plate.  Replaces utils.py:478-510 (chain_reduce / logmmexp /
// chain_logmmexp) and the trailing t.logsumexp(lp, -1) of logpq.py:139.
//
// The reference reduces [T,K,K] by a pairwise tree (ceil(log2 T) rounds, ~9 torch ops per round).
// Log-matrix multiplication is associative, so here each workgroup multiplies a contiguous SEGMENT
// of matrices left to right entirely in LDS (one logmmexp per step, same normalisation and
// eps-in-log as utils.py:503-507), and a few levels of segments reduce T -> 1.  Re-bracketing only
// changes rounding.  At T=1000, K=30 the whole input is 3.6 MB: this path is latency-bound, so the
// design goal is few launches (3 levels), not bandwidth.
#include "common.h"

namespace alan {

constexpr int CHAIN_THREADS = 256;

// P, C: log-space [K,K] in LDS (row stride K).  On exit P = logmmexp(P, C).  pe/ce alias P/C.
template <typename T>
__device__ void logmm_step(T *P, T *C, T *pm, T *cm, int K) {
    const int tid = threadIdx.x;
    // row max of P, column max of C   (utils.py:503-504)
    for (int i = tid; i < K; i += CHAIN_THREADS) {
        T a = Num<T>::ninf(), b = Num<T>::ninf();
        for (int k = 0; k < K; ++k) {
            a = fmax(a, P[i * K + k]);
            b = fmax(b, C[k * K + i]);
        }
        pm[i] = a;
        cm[i] = b;
    }
    __syncthreads();
    for (int e = tid; e < K * K; e += CHAIN_THREADS) {
        const int i = e / K, j = e - i * K;
        P[e] = Num<T>::exp(P[e] - pm[i]);
        C[e] = Num<T>::exp(C[e] - cm[j]);
    }
    __syncthreads();
    // R = Pe @ Ce, kept in registers until everyone has finished reading Pe
    constexpr int MAXE = 40;  // K <= 100 -> ceil(K*K/256) <= 40
    T acc[MAXE];
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = tid + q * CHAIN_THREADS;
        T a = T(0);
        if (e < K * K) {
            const int i = e / K, j = e - i * K;
            for (int k = 0; k < K; ++k) a += P[i * K + k] * C[k * K + j];
        }
        acc[q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = tid + q * CHAIN_THREADS;
        if (e < K * K) {
            const int i = e / K, j = e - i * K;
            P[e] = Num<T>::log(acc[q] + Num<T>::eps) + pm[i] + cm[j];  // utils.py:506-507
        }
    }
    __syncthreads();
}

// One workgroup per segment: out[seg] = ms[t0] (x) ms[t0+1] (x) ... (x) ms[t1-1]   (log-space)
// If vec_out != nullptr (final level, one segment) also writes logsumexp(out, -1).
template <typename T>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_segment_kernel(
    const T *ms, int64_t sT, int64_t sRow, int64_t sCol, int T_total, int seg_len, int K,
    T *out, T *vec_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);
    T *C = P + K * K;
    T *pm = C + K * K;
    T *cm = pm + K;
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * seg_len;
    const int t1 = min(T_total, t0 + seg_len);

    for (int e = tid; e < K * K; e += CHAIN_THREADS) {
        const int i = e / K, j = e - i * K;
        P[e] = ms[(int64_t)t0 * sT + i * sRow + j * sCol];
    }
    for (int t = t0 + 1; t < t1; ++t) {
        for (int e = tid; e < K * K; e += CHAIN_THREADS) {
            const int i = e / K, j = e - i * K;
            C[e] = ms[(int64_t)t * sT + i * sRow + j * sCol];
        }
        __syncthreads();
        logmm_step<T>(P, C, pm, cm, K);
    }
    __syncthreads();
    if (out)
        for (int e = tid; e < K * K; e += CHAIN_THREADS) out[(int64_t)blockIdx.x * K * K + e] = P[e];
    if (vec_out) {
        // torch.logsumexp(lp, -1)  (logpq.py:139): no eps; -inf rows stay -inf
        for (int i = tid; i < K; i += CHAIN_THREADS) {
            T mx = Num<T>::ninf();
            for (int j = 0; j < K; ++j) mx = fmax(mx, P[i * K + j]);
            T s = T(0);
            const T mref = (mx == Num<T>::ninf() || mx == -Num<T>::ninf()) ? T(0) : mx;
            for (int j = 0; j < K; ++j) s += Num<T>::exp(P[i * K + j] - mref);
            vec_out[i] = Num<T>::log(s) + mref;
        }
    }
}

static int pick_segment(int64_t T) {
    // 3 levels reach 1 for T <= seg^3: 10 covers T = 1000
    int seg = 2;
    while ((int64_t)seg * seg * seg < T && seg < 32) ++seg;
    return seg;
}

template <typename T>
static int chain_run(const void *ms_, int64_t Tn, int64_t K, int64_t sT, int64_t sRow, int64_t sCol,
                     void *out_chain, void *out_vec, void *ws, size_t ws_bytes, hipStream_t stream) {
    const size_t mat = (size_t)K * K * sizeof(T);
    const size_t smem = 2 * mat + 2 * K * sizeof(T);
    if (K > 100 || smem > 160 * 1024) return ALAN_ERR_UNSUPPORTED;
    auto kern = chain_segment_kernel<T>;
    if (smem > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    const int seg = pick_segment(Tn);
    const T *src = (const T *)ms_;
    int64_t n = Tn;
    T *bufs[2] = {(T *)ws, nullptr};
    // ping-pong level buffers inside the workspace
    const int64_t n1 = (Tn + seg - 1) / seg;
    bufs[1] = (T *)((char *)ws + ((n1 * mat + 255) & ~(size_t)255));
    int lvl = 0;
    int64_t cT = sT, cR = sRow, cC = sCol;
    while (true) {
        const int64_t nseg = (n + seg - 1) / seg;
        const bool last = nseg == 1;
        T *dst = last ? (T *)out_chain : bufs[lvl & 1];
        if (!last) {
            const size_t need = (size_t)((char *)dst - (char *)ws) + (size_t)nseg * mat;
            if (!ws || need > ws_bytes) return ALAN_ERR_WORKSPACE;
        }
        hipLaunchKernelGGL(kern, dim3((uint32_t)nseg), dim3(CHAIN_THREADS), smem, stream, src, cT, cR, cC,
                           (int)n, seg, (int)K, dst, last ? (T *)out_vec : (T *)nullptr);
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
        if (last) break;
        src = dst;
        cT = K * K;
        cR = K;
        cC = 1;
        n = nseg;
        ++lvl;
    }
    return ALAN_OK;
}

}  // namespace alan

using namespace alan;

extern "C" size_t alan_chain_workspace_bytes(int64_t T, int64_t K, int32_t dtype) {
    if (T < 1 || K < 1) return 0;
    const size_t mat = (size_t)K * K * (dtype == ALAN_F64 ? 8 : 4);
    int seg = 2;
    while ((int64_t)seg * seg * seg < T && seg < 32) ++seg;
    const int64_t n1 = (T + seg - 1) / seg;
    const int64_t n2 = (n1 + seg - 1) / seg;
    return ((n1 * mat + 255) & ~(size_t)255) + ((n2 * mat + 255) & ~(size_t)255);
}

extern "C" int alan_chain_logmmexp(const void *ms, int32_t dtype, int64_t T, int64_t K, int64_t sT,
                                   int64_t sRow, int64_t sCol, void *out_chain, void *out_vec,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!ms || T < 1 || K < 1 || (!out_chain && !out_vec)) return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    if (dtype == ALAN_F32)
        return chain_run<float>(ms, T, K, sT, sRow, sCol, out_chain, out_vec, workspace, workspace_bytes,
                                (hipStream_t)stream);
    if (dtype == ALAN_F64)
        return chain_run<double>(ms, T, K, sT, sRow, sCol, out_chain, out_vec, workspace, workspace_bytes,
                                 (hipStream_t)stream);
    return ALAN_ERR_BAD_DESC;
}
