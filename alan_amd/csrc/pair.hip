// Pair contraction: a reduce_Ks step whose OUTPUT is bigger than every factor --
//
//   out[b, i, j] = sum_p  LSE_a( sum_{row-side f} f[b, p, a, i]  +  sum_{col-side f} f[b, p, a, j] )  + add_const
//
// (reduce_Ks.py:249-251 logsumexp_sum, then the plate sum of logpq.py:149): bus_breakdown's Borough plate, where the
// likelihood summed over the ID plate carries (K_alpha, K_global) and log P(alpha | beta, sigma_alpha) carries
// (K_alpha, K_year) -- eliminating K_alpha leaves [K_global, K_year] per (Year, Borough): 6 log-space products of
// [100 x 100] . [100 x 100] at K = 100, 6 M (output, a) pairs from two 240 KB factors.  The generic kernel gives every
// output its own lane group that walks `a` through global memory (12 M strided loads: 28.6 us); here a workgroup owns a
// 8 x 8 tile of (i, j), stages the two tiles' rows over `a` in LDS once -- each factor element is read once per tile
// it touches, with `a` (their contiguous dim) along the lanes -- and a thread takes its output's exact max and its sum
// of exps from LDS (two passes, as utils.py:218-220 does), adding the plate elements in order (deterministic).
// NOT the matrix cores: a log-space product has no linear-space GEMM without giving up the exact max of each output.
#include <algorithm>
#include <cstring>

#include "plan.h"

namespace alan {

// BERN: the same tiles with another function of the summed term -- the Bernoulli likelihood of observations y[b, a] whose
// logits are row + col (ALAN_MODE_BERNOULLI_LINEAR with plain terms only: bus_breakdown's `alpha + phi @ x1 + psi @ x2`
// once the dot products, which lack the K_alpha dim, have been evaluated: 9 M elements at K = 100),
//   out[b, i, j] = out_scale * sum_a [ logsigmoid(l) - (1 - y[b, a]) l ] + add_const,    l = row[a, i] + col[a, j].
template <bool BERN>
__global__ __launch_bounds__(256) void pair_lse_kernel(const PairDesc d) {
    extern __shared__ __align__(16) float lds[];
    constexpr int HS = MAXF / 2;
    const int tid = threadIdx.x, R = d.R, R1 = R | 1;
    float *row = lds, *col = lds + PAIR_T * R1;          // [16][R1] each: a thread walks its own row of each
    float *yl = col + PAIR_T * R1;                       // BERN: [R]
    // four lanes per output, each a quarter of the reduce dim (interleaved): 64 outputs per workgroup -- an output per lane
    // would be fewer waves than the chip has SIMDs at bus_breakdown's size (60,000 outputs), each alone with its LDS latency
    const int q4 = tid & 3, tj = (tid >> 2) & (PAIR_T - 1), ti = tid >> 5;
    const int tile_i = blockIdx.x / d.ntj, tile_j = blockIdx.x - tile_i * d.ntj;
    const int i0 = tile_i * PAIR_T, j0 = tile_j * PAIR_T;
    // the kept batch index of this workgroup
    int32_t bb[2][HS], ob = 0;
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
        for (int h = 0; h < HS; ++h) bb[sd][h] = 0;
    {
        uint32_t o = blockIdx.y;
#pragma unroll
        for (int k = PAIR_NB - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.bdiv[k]);
            const int32_t idx = (int32_t)(o - q * d.bdiv[k].d);
            o = q;
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                for (int h = 0; h < HS; ++h) bb[sd][h] += idx * d.sb[sd][h][k];
            ob += idx * d.osb[k];
        }
    }
    int32_t yb = 0;
    if (BERN) {
        uint32_t o = blockIdx.y;
#pragma unroll
        for (int k = PAIR_NB - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.bdiv[k]);
            yb += (int32_t)(o - q * d.bdiv[k].d) * d.y_sb[k];
            o = q;
        }
    }
    const float ninf = -__builtin_huge_valf();
    float total = 0.f;
    // (split: one plate element per workgroup, gridDim.z of them; the partial results are added by pair_sum_kernel)
    const int p_lo = d.split ? (int)blockIdx.z : 0, p_hi = d.split ? (int)blockIdx.z + 1 : d.n_plate;
    for (int p = p_lo; p < p_hi; ++p) {
        int32_t pb[2][HS];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int h = 0; h < HS; ++h) pb[sd][h] = bb[sd][h];
        {
            uint32_t o = (uint32_t)p;
#pragma unroll
            for (int k = PAIR_NP - 1; k >= 0; --k) {
                const uint32_t q = fd_div(o, d.pdiv[k]);
                const int32_t idx = (int32_t)(o - q * d.pdiv[k].d);
                o = q;
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int h = 0; h < HS; ++h) pb[sd][h] += idx * d.sp[sd][h][k];
            }
        }
        __syncthreads();                                  // (the previous plate element's tiles have been read)
        // ---- stage: element (a, x) of both tiles.  Every load of a thread's share is issued before the first is used,
        // and there is NO branch around a load (a conditional load is followed by its own wait: 21 of those in a row
        // were 22 of this kernel's first 27 us): indices are clamped, factor slots a side does not use repeat its first
        // factor with weight 0, and slots beyond the tile are not stored.
        constexpr int NE = PAIR_T * PAIR_RMAX / 256;
        // element slot e of a tile = (x, a): `a` along the lanes where the side's biggest factor is contiguous in `a`,
        // else x along the lanes (16 consecutive tile rows: 64-byte runs of a factor stored [.., a, x])
        auto split = [&](int e, int a_fast, int &x, int &a) {
            if (a_fast)
                x = (int)fd_div((uint32_t)e, d.rdiv), a = e - x * R;
            else
                a = e / PAIR_T, x = e & (PAIR_T - 1);
        };
        float lr[NE][HS], lc[NE][HS];
        const float yv = BERN ? d.y[yb + min(tid, R - 1) * d.y_sa] : 0.f;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
#pragma unroll
            for (int h = 0; h < HS; ++h) lr[u][h] = lc[u][h] = 0.f;
            if (256 * u >= PAIR_T * R) continue;       // (uniform: a whole round beyond the tile -- no load, no wait)
            const int e = min(tid + 256 * u, PAIR_T * R - 1);
            int xr, ar, xc, ac;
            split(e, d.a_fast[0], xr, ar);
            split(e, d.a_fast[1], xc, ac);
            const int ir = min(i0 + xr, d.NI - 1), jc = min(j0 + xc, d.NJ - 1);
#pragma unroll
            for (int h = 0; h < HS; ++h) {
                lr[u][h] = d.f[0][h][pb[0][h] + ar * d.sa[0][h] + ir * d.sx[0][h]];
                lc[u][h] = d.f[1][h][pb[1][h] + ac * d.sa[1][h] + jc * d.sx[1][h]];
            }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u;
            float rv = 0.f, cv = 0.f;
#pragma unroll
            for (int h = 0; h < HS; ++h) {             // (a select, not a product with 0: 0 x -inf would be NaN)
                rv += h < d.ns[0] ? d.w[0][h] * lr[u][h] : 0.f;
                cv += h < d.ns[1] ? d.w[1][h] * lc[u][h] : 0.f;
            }
            if (e < PAIR_T * R) {
                int xr, ar, xc, ac;
                split(e, d.a_fast[0], xr, ar);
                split(e, d.a_fast[1], xc, ac);
                row[xr * R1 + ar] = rv;
                col[xc * R1 + ac] = cv;
            }
        }
        if (BERN && tid < R) yl[tid] = yv;
        __syncthreads();
        const float *rp = row + ti * R1, *cp = col + tj * R1;
        if (BERN) {
            // logsigmoid(x) - (1 - y) x on the fast transcendental instructions, as the lane-group kernel evaluates it
            float s0 = 0.f, s1 = 0.f;
            int a = q4;
            for (; a + 4 < R; a += 8) {
                const float x0 = rp[a] + cp[a], x1 = rp[a + 4] + cp[a + 4];
                const float e0 = __builtin_amdgcn_exp2f(-fabsf(x0) * 1.44269504088896340736f);
                const float e1 = __builtin_amdgcn_exp2f(-fabsf(x1) * 1.44269504088896340736f);
                s0 += fminf(x0, 0.f) - __builtin_amdgcn_logf(1.f + e0) * 0.69314718055994530942f - (1.f - yl[a]) * x0;
                s1 += fminf(x1, 0.f) - __builtin_amdgcn_logf(1.f + e1) * 0.69314718055994530942f - (1.f - yl[a + 4]) * x1;
            }
            for (; a < R; a += 4) {
                const float x0 = rp[a] + cp[a];
                const float e0 = __builtin_amdgcn_exp2f(-fabsf(x0) * 1.44269504088896340736f);
                s0 += fminf(x0, 0.f) - __builtin_amdgcn_logf(1.f + e0) * 0.69314718055994530942f - (1.f - yl[a]) * x0;
            }
            s0 += s1;
            s0 += __shfl_xor(s0, 1);
            s0 += __shfl_xor(s0, 2);
            s1 = 0.f;
            total += (s0 + s1) * d.out_scale;
            continue;
        }
        // ---- this thread's output: exact max, then the sum of exps (utils.py:218-220)
        float m0 = ninf, m1 = ninf;
        int a = q4;
        for (; a + 4 < R; a += 8) {
            m0 = fmaxf(m0, rp[a] + cp[a]);
            m1 = fmaxf(m1, rp[a + 4] + cp[a + 4]);
        }
        for (; a < R; a += 4) m0 = fmaxf(m0, rp[a] + cp[a]);
        // (fmaxf drops a NaN operand: a NaN term must poison the result as torch's amax does -- it does through the sum)
        float m = fmaxf(m0, m1);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float s0 = 0.f, s1 = 0.f;
        const float mf = (m == ninf || m == -ninf) ? 0.f : m;
        for (a = q4; a + 4 < R; a += 8) {
            s0 += __expf(rp[a] + cp[a] - mf);
            s1 += __expf(rp[a + 4] + cp[a + 4] - mf);
        }
        for (; a < R; a += 4) s0 += __expf(rp[a] + cp[a] - mf);
        s0 += s1;
        s0 += __shfl_xor(s0, 1);
        s0 += __shfl_xor(s0, 2);
        total += lse_finish(m, s0);                       // (a NaN term: exp(NaN) = NaN reaches the sum)
    }
    if (q4 == 0 && i0 + ti < d.NI && j0 + tj < d.NJ) {
        if (d.split)
            d.ws[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * d.NI + i0 + ti) * d.NJ + j0 + tj] = total;
        else
            d.out[ob + (i0 + ti) * d.osi + (j0 + tj) * d.osj] = total + d.add_const;
    }
}

// out[b, i, j] = sum_p ws[p, b, i, j] + add_const, the plate elements in order
__global__ __launch_bounds__(256) void pair_sum_kernel(const PairDesc d, const int n_batch) {
    const uint32_t n = (uint32_t)n_batch * d.NI * d.NJ, o = blockIdx.x * 256u + threadIdx.x;
    if (o >= n) return;
    float tot = 0.f;
    for (int p = 0; p < d.n_plate; ++p) tot += d.ws[(size_t)p * n + o];
    const uint32_t bi = o / (uint32_t)d.NJ, j = o - bi * d.NJ, b = bi / (uint32_t)d.NI, i = bi - b * d.NI;
    int32_t ob = 0;
    uint32_t r = b;
#pragma unroll
    for (int k = PAIR_NB - 1; k >= 0; --k) {
        const uint32_t q = fd_div(r, d.bdiv[k]);
        ob += (int32_t)(r - q * d.bdiv[k].d) * d.osb[k];
        r = q;
    }
    d.out[ob + (int32_t)i * d.osi + (int32_t)j * d.osj] = tot + d.add_const;
}

// Is this alan_reduce call a pair contraction worth the tile kernel?  Fills its argument.
bool pair_prepare(const alan_reduce_desc_t &d, uint32_t keep, uint32_t red, uint32_t plate, PairDesc &pd, dim3 &grid,
                  size_t &lds_bytes) {
    static const int knob = env_knob("ALAN_PAIR");                                   // ablation knob: 0 = off
    if (knob == 0) return false;
    const bool bern = d.mode == ALAN_MODE_BERNOULLI_LINEAR;
    if ((d.mode != ALAN_MODE_LSE && !bern) || d.n_factors < 2 || d.n_factors > MAXF) return false;
    if (d.weight.data || d.lse_out.data || d.ring_n || d.out.dtype != ALAN_F32 || !d.out.data) return false;
    for (int f = 0; f < d.n_factors; ++f)
        if (d.factor[f].dtype != ALAN_F32 || !d.factor[f].data) return false;
    // BERN: factor 0 is the observations; every other factor a plain term of its own (term ids 1, 2, ...: no dot
    // products, no DOT dims) -- and no plate sum, which that mode does not have
    const int f0 = bern ? 1 : 0;
    if (bern) {
        if (plate || d.n_factors < 3) return false;
        for (int f = 1; f < d.n_factors; ++f)
            if ((int)d.factor[f].scale != f) return false;
        for (int i = 0; i < d.ndim; ++i)
            if (d.role[i] == ALAN_DOT && d.size[i] > 1) return false;
    }
    // the one reduce dim; kept dims; plate dims (sizes > 1 only)
    int ia = -1, kd[MAXD], nk = 0, pl[MAXD], np = 0;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.size[i] <= 1) continue;
        if ((red >> i) & 1) {
            if (ia >= 0) return false;
            ia = i;
        } else if ((plate >> i) & 1) {
            pl[np++] = i;
        } else if ((keep >> i) & 1) {
            kd[nk++] = i;
        }
    }
    if (ia < 0 || d.size[ia] > PAIR_RMAX || d.size[ia] < 8 || np > PAIR_NP || nk < 2 || nk > PAIR_NB + 2) return false;
    // the tile dims: two kept dims no factor carries both of, the pair with the most outputs
    int bi = -1, bj = -1;
    int64_t best = 0;
    for (int x = 0; x < nk; ++x)
        for (int y = x + 1; y < nk; ++y) {
            bool ok = true;
            for (int f = f0; f < d.n_factors; ++f)
                if (d.factor[f].stride[kd[x]] != 0 && d.factor[f].stride[kd[y]] != 0) ok = false;
            if (bern && (d.factor[0].stride[kd[x]] != 0 || d.factor[0].stride[kd[y]] != 0)) ok = false;
            const int64_t n = d.size[kd[x]] * d.size[kd[y]];
            if (ok && n > best) best = n, bi = kd[x], bj = kd[y];
        }
    if (bi < 0 || d.size[bi] < 8 || d.size[bj] < 8) return false;
    int64_t n_batch = 1, n_plate = 1;
    for (int x = 0; x < nk; ++x)
        if (kd[x] != bi && kd[x] != bj) n_batch *= d.size[kd[x]];
    for (int x = 0; x < np; ++x) n_plate *= d.size[pl[x]];
    // worth it: a million (output, a) pairs or more (below that the small kernels are launch-bound anyway), and every
    // factor smaller than the output (else the rows kernel streams the big factor)
    const int64_t pairs = best * n_batch * n_plate * d.size[ia];
    if (pairs < (1ll << 20) || n_batch > 65535 || n_plate > (1 << 20)) return false;
    std::memset(&pd, 0, sizeof(pd));
    const int64_t lim = (1ll << 31) - 1;
    for (int k = 0; k < PAIR_NB; ++k) pd.bdiv[k] = make_fastdiv(1);
    for (int k = 0; k < PAIR_NP; ++k) pd.pdiv[k] = make_fastdiv(1);
    int bdims[PAIR_NB], nb = 0;
    for (int x = 0; x < nk; ++x)
        if (kd[x] != bi && kd[x] != bj) bdims[nb++] = kd[x];
    if (nb > PAIR_NB) return false;
    for (int k = 0; k < nb; ++k) pd.bdiv[PAIR_NB - nb + k] = make_fastdiv((uint32_t)d.size[bdims[k]]);
    for (int k = 0; k < np; ++k) pd.pdiv[PAIR_NP - np + k] = make_fastdiv((uint32_t)d.size[pl[k]]);
    auto reach_ok = [&](const alan_tensor_t &x) {
        int64_t reach = 0;
        for (int i = 0; i < d.ndim; ++i) {
            const int64_t st = x.stride[i] < 0 ? -x.stride[i] : x.stride[i];
            if (st > lim) return false;
            reach += (d.size[i] - 1) * st;
        }
        return reach <= lim;
    };
    int n_side[2] = {0, 0};
    int64_t big[2] = {-1, -1};
    pd.a_fast[0] = pd.a_fast[1] = 1;
    if (bern) {
        const alan_tensor_t &y = d.factor[0];
        if (!reach_ok(y)) return false;
        for (int i = 0; i < d.ndim; ++i)
            if (y.stride[i] < 0) return false;
        pd.y = (const float *)y.data;
        pd.y_sa = (int32_t)y.stride[ia];
        for (int k = 0; k < nb; ++k) pd.y_sb[PAIR_NB - nb + k] = (int32_t)y.stride[bdims[k]];
        pd.out_scale = d.out.scale;
    }
    for (int f = f0; f < d.n_factors; ++f) {
        const alan_tensor_t &x = d.factor[f];
        if (!reach_ok(x)) return false;
        for (int i = 0; i < d.ndim; ++i)
            if (x.stride[i] < 0) return false;
        const int sd = x.stride[bj] != 0 ? 1 : 0, h = n_side[sd]++;
        if (h == MAXF / 2) return false;
        pd.f[sd][h] = (const float *)x.data;
        pd.w[sd][h] = bern ? 1.f : x.scale;
        pd.sa[sd][h] = (int32_t)x.stride[ia];
        pd.sx[sd][h] = (int32_t)x.stride[sd ? bj : bi];
        for (int k = 0; k < nb; ++k) pd.sb[sd][h][PAIR_NB - nb + k] = (int32_t)x.stride[bdims[k]];
        for (int k = 0; k < np; ++k) pd.sp[sd][h][PAIR_NP - np + k] = (int32_t)x.stride[pl[k]];
        // lanes along `a` or along the tile dim: whichever the side's biggest factor stores closer together
        int64_t ext = 1;
        for (int i = 0; i < d.ndim; ++i)
            if (x.stride[i] != 0) ext *= d.size[i];
        if (ext > big[sd]) {
            big[sd] = ext;
            pd.a_fast[sd] = pd.sx[sd][h] == 0 || (pd.sa[sd][h] != 0 && pd.sa[sd][h] <= pd.sx[sd][h]);
        }
    }
    if (n_side[0] == 0 || n_side[1] == 0) return false;
    pd.ns[0] = n_side[0], pd.ns[1] = n_side[1];
    for (int sd = 0; sd < 2; ++sd)
        for (int h = n_side[sd]; h < MAXF / 2; ++h) {
            pd.f[sd][h] = pd.f[sd][0], pd.w[sd][h] = 0.f, pd.sa[sd][h] = pd.sa[sd][0], pd.sx[sd][h] = pd.sx[sd][0];
            for (int k = 0; k < PAIR_NB; ++k) pd.sb[sd][h][k] = pd.sb[sd][0][k];
            for (int k = 0; k < PAIR_NP; ++k) pd.sp[sd][h][k] = pd.sp[sd][0][k];
        }
    if (!reach_ok(d.out)) return false;
    for (int i = 0; i < d.ndim; ++i)
        if (d.out.stride[i] < 0) return false;
    for (int x = 0; x < np; ++x)
        if (d.out.stride[pl[x]] != 0) return false;
    if (d.out.stride[ia] != 0) return false;
    pd.out = (float *)d.out.data;
    pd.osi = (int32_t)d.out.stride[bi], pd.osj = (int32_t)d.out.stride[bj];
    for (int k = 0; k < nb; ++k) pd.osb[PAIR_NB - nb + k] = (int32_t)d.out.stride[bdims[k]];
    pd.R = (int32_t)d.size[ia], pd.NI = (int32_t)d.size[bi], pd.NJ = (int32_t)d.size[bj];
    pd.n_plate = (int32_t)n_plate;
    pd.rdiv = make_fastdiv((uint32_t)pd.R);
    pd.add_const = (float)d.add_const;
    const int nti = (pd.NI + PAIR_T - 1) / PAIR_T;
    pd.ntj = (pd.NJ + PAIR_T - 1) / PAIR_T;
    grid = dim3((uint32_t)(nti * pd.ntj), (uint32_t)n_batch);
    // few tiles and several plate elements: a workgroup per (tile, plate element) -- bus_breakdown at K = 100 is 98 tiles of
    // 3 plate elements each, a third of the chip's CUs with one wave per SIMD -- and a second launch that adds the plate
    // elements in order
    pd.bern = bern ? 1 : 0;
    if (n_plate >= 2 && n_plate <= 65535 && (int64_t)grid.x * grid.y < 512 &&
        n_batch * pd.NI * pd.NJ < (1ll << 31)) {
        pd.split = 1;
        grid.z = (uint32_t)n_plate;
    }
    lds_bytes = ((size_t)2 * PAIR_T * (pd.R | 1) + (bern ? pd.R : 0)) * sizeof(float);
    return true;
}

size_t pair_workspace_bytes(const PairDesc &pd, dim3 grid) {
    return pd.split ? (size_t)pd.n_plate * grid.y * pd.NI * pd.NJ * sizeof(float) : 0;
}

int launch_pair(const PairDesc &pd_, dim3 grid, size_t lds_bytes, void *workspace, size_t workspace_bytes,
                hipStream_t stream, const EvPair &ev) {
    PairDesc pd = pd_;
    if (pd.split) {
        if (!workspace || workspace_bytes < pair_workspace_bytes(pd, grid)) return ALAN_ERR_WORKSPACE;
        pd.ws = (float *)workspace;
    }
    if (pd.bern)
        ALAN_LAUNCH_EXT(pair_lse_kernel<true>, grid, dim3(256), lds_bytes, stream, ev.start, ev.stop, 0, pd);
    else
        ALAN_LAUNCH_EXT(pair_lse_kernel<false>, grid, dim3(256), lds_bytes, stream, ev.start, ev.stop, 0, pd);
    if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    if (pd.split) {
        const uint32_t n = grid.y * (uint32_t)pd.NI * (uint32_t)pd.NJ;
        ALAN_LAUNCH(pair_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pd, (int)grid.y);
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    }
    return ALAN_OK;
}

}  // namespace alan
