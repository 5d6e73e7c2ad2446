// alan_chain_logmmexp: the timeseries plate.  Replaces utils.py:478-510 (chain_reduce / logmmexp /
// chain_logmmexp) and the trailing t.logsumexp(lp, -1) of logpq.py:139.
//
// The reference reduces [T,K,K] by a pairwise tree (ceil(log2 T) rounds, ~9 torch ops per round).
// Log-matrix multiplication is associative, so here each workgroup multiplies a contiguous SEGMENT
// of matrices left to right entirely in LDS (one logmmexp per step, same normalisation and
// eps-in-log as utils.py:503-507), and a few levels of segments reduce T -> 1.  Re-bracketing only
// changes rounding.  At T=1000, K=30 the whole input is 3.6 MB: this path is latency-bound, so the
// design goal is few launches (3 levels), not bandwidth.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace alan {

constexpr int CHAIN_THREADS = 256;

template <typename T>
struct alignas(4 * sizeof(T)) Vec4 {
    T x, y, z, w;
};

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
    const T *ms, int64_t sT, int64_t sRow, int64_t sCol, int T_total, int seg_len, int K,
    T *out, T *vec_out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
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
        const T v = ms[(int64_t)t0 * sT + i * sRow + j * sCol];
        P[i * PS + j] = v;
        lds_max(&pm[i], v);
        if (t0 + 1 < t1) {
            const T c = ms[(int64_t)(t0 + 1) * sT + i * sRow + j * sCol];
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
                    creg[q] = ms[(int64_t)tt * sT + i * sRow + j * sCol];
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

// Segment length: levels cost a launch each (a ~ 5.7 us measured), every level runs its longest segment's
// (len - 1) dependent steps (b ~ 2.4 us at K <= 32, growing with the K^2 work of a step).  T = 1000, K = 30:
// 4 (five levels of 3 steps) rather than 10 (three levels of 9).
static int pick_segment(int64_t T, int64_t K) {
    if (const char *e = getenv("ALAN_CHAIN_SEG")) return std::max(2, std::min(64, atoi(e)));   // tuning knob
    const double kk = std::max(1.0, (double)K / 32.0);
    const double a = 5.7, b = 2.44 * kk * kk;
    int best = 2;
    double best_cost = 1e300;
    for (int seg = 2; seg <= 32; ++seg) {
        double cost = 0;
        for (int64_t n = T; n > 1; n = (n + seg - 1) / seg) cost += a + (double)(std::min<int64_t>(seg, n) - 1) * b;
        if (cost < best_cost) best_cost = cost, best = seg;
    }
    return best;
}

template <typename T>
static int chain_run(const void *ms_, int64_t Tn, int64_t K, int64_t sT, int64_t sRow, int64_t sCol,
                     void *out_chain, void *out_vec, void *ws, size_t ws_bytes, hipStream_t stream) {
    const size_t mat = (size_t)K * K * sizeof(T);
    const size_t KP = (size_t)((K + 3) & ~3);
    const size_t smem = (KP * (KP + 4) + KP * KP + 4 * KP) * sizeof(T);
    if (K > 100 || smem > 160 * 1024) return ALAN_ERR_UNSUPPORTED;
    auto kern = K <= 32 ? chain_segment_kernel<T, 4, 1> : K <= 64 ? chain_segment_kernel<T, 16, 4>
                                                                  : chain_segment_kernel<T, 40, 10>;
    if (smem > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return ALAN_ERR_LAUNCH;
    const int seg = pick_segment(Tn, K);
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
    const int seg = alan::pick_segment(T, K);
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
