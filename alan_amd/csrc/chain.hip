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

// ------------------------------------------------------------------------------------------------
// Backward of  out[i] = logsumexp_j (M_1 (x) ... (x) M_T)[i, j]  with respect to every M_t:
//   d out_i / d M_t[a,b] = exp(alpha_{t-1}[i,a] + M_t[a,b] + beta_t[b] - out_i)       (a pairwise marginal)
//   grad M_t[a,b] = sum_i g_i * (that)  = exp(la_{t-1}[a] + M_t[a,b] + beta_t[b]),
//   la_0[a] = log|g_a| - out_a,  la_t[b] = LSE_a(la_{t-1}[a] + M_t[a,b]),  beta_T = 0,
//   beta_{t-1}[a] = LSE_b(M_t[a,b] + beta_t[b]).
// Two O(T K^2) scans (forward for la, backward for beta + the gradient), one workgroup; positive and
// negative parts of g are scanned separately (log domain) and subtracted.  Latency-bound by design:
// this is the gradient of a 3.6 MB / 54 MFLOP problem.
template <typename T>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_backward_kernel(
    const T *ms, int64_t sT, int64_t sRow, int64_t sCol, int Tn, int K, const T *out_vec,
    const T *grad_out, T *grad_ms, T *la_ws /* [Tn][K] */, double *off_ws /* [Tn] */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *M = reinterpret_cast<T *>(smem_raw);   // [K][K] current matrix
    T *vec = M + K * K;                        // [K]  la_{t-1} or beta_t, max-normalised
    T *nxt = vec + K;                          // [K]
    const int tid = threadIdx.x;
    const T NINF = Num<T>::ninf();

    // Both scans keep their vectors max-normalised and carry the (large, mutually cancelling) offsets
    // as double scalars: la ~ +|log evidence| and beta ~ -|log evidence| would otherwise lose
    // ~1e-4 of relative precision in fp32 at T = 1000.
    auto renorm = [&](double &off) {   // vec <- nxt - max(nxt); off += max   (every thread, uniformly)
        T mx = NINF;
        for (int a = 0; a < K; ++a) mx = fmax(mx, nxt[a]);
        const T sub = (mx == NINF) ? T(0) : mx;
        off += (double)sub;
        __syncthreads();
        for (int a = tid; a < K; a += CHAIN_THREADS) vec[a] = nxt[a] - sub;
        __syncthreads();
    };

    for (int sign = 0; sign < 2; ++sign) {
        // ---- forward scan: la_t
        for (int a = tid; a < K; a += CHAIN_THREADS) {
            const T g = sign == 0 ? grad_out[a] : -grad_out[a];
            nxt[a] = g > T(0) ? Num<T>::log(g) - out_vec[a] : NINF;
        }
        __syncthreads();
        bool any = false;
        for (int a = 0; a < K; ++a) any = any || (nxt[a] != NINF);
        if (!any) {   // no weight of this sign (uniform decision)
            if (sign == 0)
                for (int64_t e = tid; e < (int64_t)Tn * K * K; e += CHAIN_THREADS) grad_ms[e] = T(0);
            __syncthreads();
            continue;
        }
        double off = 0.0;
        renorm(off);
        for (int a = tid; a < K; a += CHAIN_THREADS) la_ws[a] = vec[a];
        if (tid == 0) off_ws[0] = off;
        for (int t = 0; t + 1 < Tn; ++t) {
            for (int e = tid; e < K * K; e += CHAIN_THREADS) {
                const int a = e / K, b = e - a * K;
                M[e] = ms[(int64_t)t * sT + a * sRow + b * sCol];
            }
            __syncthreads();
            for (int b = tid; b < K; b += CHAIN_THREADS) {
                T m = NINF, s = T(0);
                for (int a = 0; a < K; ++a) lse_push(m, s, vec[a] + M[a * K + b]);
                nxt[b] = (m == NINF) ? NINF : Num<T>::log(s) + m;
            }
            __syncthreads();
            renorm(off);
            for (int b = tid; b < K; b += CHAIN_THREADS) la_ws[(int64_t)(t + 1) * K + b] = vec[b];
            if (tid == 0) off_ws[t + 1] = off;
        }
        __syncthreads();
        // ---- backward scan: beta_t and the gradient
        double boff = 0.0;
        for (int b = tid; b < K; b += CHAIN_THREADS) vec[b] = T(0);
        __syncthreads();
        for (int t = Tn - 1; t >= 0; --t) {
            for (int e = tid; e < K * K; e += CHAIN_THREADS) {
                const int a = e / K, b = e - a * K;
                M[e] = ms[(int64_t)t * sT + a * sRow + b * sCol];
            }
            __syncthreads();
            const T shift = (T)(off_ws[t] + boff);
            for (int e = tid; e < K * K; e += CHAIN_THREADS) {
                const int a = e / K, b = e - a * K;
                const T lw = la_ws[(int64_t)t * K + a];
                const T gv = (lw == NINF) ? T(0) : Num<T>::exp((lw + M[e] + vec[b]) + shift);
                T *dst = grad_ms + (int64_t)t * K * K + e;
                if (sign == 0)
                    *dst = gv;
                else
                    *dst -= gv;
            }
            for (int a = tid; a < K; a += CHAIN_THREADS) {
                T m = NINF, s = T(0);
                for (int b = 0; b < K; ++b) lse_push(m, s, M[a * K + b] + vec[b]);
                nxt[a] = (m == NINF) ? NINF : Num<T>::log(s) + m;
            }
            __syncthreads();
            renorm(boff);
        }
    }
}

template <typename T>
static int chain_backward_run(const void *ms, int64_t Tn, int64_t K, int64_t sT, int64_t sRow, int64_t sCol,
                              const void *out_vec, const void *grad_out, void *grad_ms, void *ws,
                              size_t ws_bytes, hipStream_t stream) {
    const size_t smem = ((size_t)K * K + 2 * K) * sizeof(T);
    if (K > 128 || smem > 160 * 1024) return ALAN_ERR_UNSUPPORTED;
    const size_t la_bytes = ((size_t)Tn * K * sizeof(T) + 255) & ~(size_t)255;
    if (!ws || ws_bytes < la_bytes + (size_t)Tn * sizeof(double)) return ALAN_ERR_WORKSPACE;
    auto kern = chain_backward_kernel<T>;
    if (smem > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(1), dim3(CHAIN_THREADS), smem, stream, (const T *)ms, sT, sRow, sCol, (int)Tn,
                       (int)K, (const T *)out_vec, (const T *)grad_out, (T *)grad_ms, (T *)ws,
                       (double *)((char *)ws + la_bytes));
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

}  // namespace alan

using namespace alan;

extern "C" size_t alan_chain_backward_workspace_bytes(int64_t T, int64_t K, int32_t dtype) {
    if (T < 1 || K < 1) return 0;
    return (((size_t)T * K * (dtype == ALAN_F64 ? 8 : 4) + 255) & ~(size_t)255) + (((size_t)T * 8 + 255) & ~(size_t)255);
}

extern "C" int alan_chain_logmmexp_backward(const void *ms, int32_t dtype, int64_t T, int64_t K, int64_t sT,
                                            int64_t sRow, int64_t sCol, const void *out_vec,
                                            const void *grad_out, void *grad_ms, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    if (!ms || !out_vec || !grad_out || !grad_ms || T < 1 || K < 1) return ALAN_ERR_BAD_DESC;
    if (T >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    if (dtype == ALAN_F32)
        return chain_backward_run<float>(ms, T, K, sT, sRow, sCol, out_vec, grad_out, grad_ms, workspace,
                                         workspace_bytes, (hipStream_t)stream);
    if (dtype == ALAN_F64)
        return chain_backward_run<double>(ms, T, K, sT, sRow, sCol, out_vec, grad_out, grad_ms, workspace,
                                          workspace_bytes, (hipStream_t)stream);
    return ALAN_ERR_BAD_DESC;
}

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
