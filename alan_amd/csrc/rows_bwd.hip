// Backward of the rows kernel's problem (rows.hip): out[row] = sum_p LSE_j( F[p,row,j] + sum_f sh_f[p,j] ).
// With the saved per-(p,row) log-sum-exp values and the upstream gradient G[row]
//
//     w[p,row,j]   = exp(F[p,row,j] + sum_f sh_f[p,j] - lse[p,row]) * G[row]
//     grad F       = w                                (written where F's gradient is wanted)
//     grad sh_f    = sum over rows (and over every dim sh_f lacks) of w
//
// which is what autograd derives from utils.py:218-220 + logpq.py:149 (up to the eps term).  One pass over F
// produces every gradient: the slab is staged in LDS exactly as in the forward, each lane turns its row into w
// in place, the slab is streamed back out as grad F, and its column sums (a per-window partial of grad sh_f) are
// taken from LDS.  The generic route is one ALAN_MODE_WEXPSUM launch per factor, each re-reading F with the
// access pattern of whichever dims that factor keeps (3 x 63 us at the movielens K=30 size against ~20 us here).
#include <algorithm>
#include <climits>
#include <cstring>

#include "plan.h"

namespace alan {

struct RowsBwdDesc {
    const float *F;
    float *gF;              // nullable
    int64_t total;
    int32_t L, RB;
    uint32_t NO, P, p_chunk;
    int32_t nshared, nki, gs_off, cs_off;
    const float *lse;
    int64_t l_ps;
    const float *G;
    float *part;            // nullable: [n_windows][P][L] column sums of w
    const float *sh_p[MAXF];
    int64_t sh_ps[MAXF];
    int32_t sh_rs[MAXF];
    FastDiv kdiv[MAXD];
    int64_t lks[MAXD], gks[MAXD];
};

template <int LOGG>
__global__ __launch_bounds__(256) void rows_backward_kernel(const RowsBwdDesc d) {
    extern __shared__ __align__(16) float lds[];
    constexpr int G = 1 << LOGG;
    const int t = threadIdx.x;
    const int L = d.L;
    const uint32_t o0 = blockIdx.x * (uint32_t)d.RB;
    const uint32_t nrows = min((uint32_t)d.RB, d.NO - o0);
    const uint32_t p0 = blockIdx.y * d.p_chunk;
    const uint32_t p1 = min(d.P, p0 + d.p_chunk);
    const int gl = t & (G - 1);
    const bool has_row = (uint32_t)(t >> LOGG) < nrows;
    const int r = has_row ? (t >> LOGG) : (int)nrows - 1;
    float *gs = lds + d.gs_off;
    float *cs = lds + d.cs_off;

    int64_t lbase = 0, gbase = 0;
    {
        uint32_t o = o0 + (uint32_t)r;
        for (int k = d.nki - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.kdiv[k]);
            const int64_t idx = (int64_t)(o - q * d.kdiv[k].d);
            o = q;
            lbase += idx * d.lks[k];
            gbase += idx * d.gks[k];
        }
    }
    const float g_row = d.G[gbase];
    const int nparts = max(1, 256 / L);

    for (uint32_t p = p0; p < p1; ++p) {
        // ---- stage the slab: 16-byte loads from the aligned-down start (as the forward does)
        const int64_t e0 = ((int64_t)p * d.NO + o0) * L;
        const int64_t a0 = e0 & ~(int64_t)3;
        const int shift = (int)(e0 - a0);
        const int n4 = (shift + (int)nrows * L + 3) >> 2;
        if (a0 + 4 * (int64_t)n4 <= d.total) {
            const float4 *src = reinterpret_cast<const float4 *>(d.F + a0);
            float4 *dst = reinterpret_cast<float4 *>(lds);
#pragma unroll 4
            for (int i = t; i < n4; i += 256) dst[i] = src[i];
        } else {
            const int n = (int)(d.total - a0);
            for (int i = t; i < n; i += 256) lds[i] = d.F[a0 + i];
        }
        if (t < L) {
            float g = 0.f;
            for (int f = 0; f < d.nshared; ++f) g += d.sh_p[f][(int64_t)p * d.sh_ps[f] + (int64_t)t * d.sh_rs[f]];
            gs[t] = g;
        }
        __syncthreads();
        // ---- w in place of this lane's elements of its row
        if (has_row) {
            const float lse_pr = d.lse[lbase + (int64_t)p * d.l_ps];
            float *row = lds + shift + r * L;
#pragma unroll 4
            for (int j = gl; j < L; j += G) row[j] = __expf(row[j] + gs[j] - lse_pr) * g_row;
        }
        __syncthreads();
        // ---- grad F: the slab streamed back, consecutive lanes -> consecutive addresses
        const int nel = (int)nrows * L;
        if (d.gF) {
#pragma unroll 4
            for (int i = t; i < nel; i += 256) d.gF[e0 + i] = lds[shift + i];
        }
        // ---- column sums over the window's rows
        if (d.part) {
            if (t < nparts * L) {
                const int pt = t / L, j = t - pt * L;
                float s = 0.f;
                for (int rr = pt; rr < (int)nrows; rr += nparts) s += lds[shift + rr * L + j];
                cs[t] = s;
            }
            __syncthreads();
            if (t < L) {
                float s = 0.f;
                for (int pt = 0; pt < nparts; ++pt) s += cs[pt * L + t];
                d.part[((int64_t)blockIdx.x * d.P + p) * L + t] = s;
            }
        }
        __syncthreads();
    }
}

}  // namespace alan

using namespace alan;

namespace {

struct BwdPlan {
    Canon c;
    RowsPlan rp;
    int k0 = 0;           // 1 when the canonical problem has a plate dim
    uint32_t n_chunks = 1, p_chunk = 1;
    size_t part_bytes = 0;
    bool want_part = false;
};

// ALAN_OK when the backward fits the rows path; ALAN_ERR_UNSUPPORTED otherwise (the caller then runs one
// ALAN_MODE_WEXPSUM launch per factor).
int plan_backward(const alan_backward_desc_t &b, BwdPlan &bp) {
    const alan_reduce_desc_t &d = b.fwd;
    if (d.mode != ALAN_MODE_LSE || !d.weight.data || !d.lse_out.data) return ALAN_ERR_BAD_DESC;
    if (d.ndim < 0 || d.ndim > MAXD || d.n_factors < 1 || d.n_factors > MAXF) return ALAN_ERR_BAD_DESC;
    uint32_t keep = 0, red = 0, plate = 0;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.role[i] == ALAN_KEEP) keep |= 1u << i;
        else if (d.role[i] == ALAN_REDUCE) red |= 1u << i;
        else if (d.role[i] == ALAN_PLATE) plate |= 1u << i;
        else return ALAN_ERR_BAD_DESC;
    }
    if (!red) return ALAN_ERR_UNSUPPORTED;
    if (d.weight.dtype != ALAN_F32 || d.lse_out.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    for (int f = 0; f < d.n_factors; ++f) {
        if (d.factor[f].scale != 1.f || d.factor[f].dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
        if (!b.grad[f].data) continue;
        if (b.grad[f].dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
        for (int i = 0; i < d.ndim; ++i)   // gradients are laid out like their factors
            if (d.size[i] > 1 && b.grad[f].stride[i] != d.factor[f].stride[i]) return ALAN_ERR_UNSUPPORTED;
    }
    alan_reduce_desc_t tmp = d;
    tmp.mode = ALAN_MODE_WEXPSUM;           // canonicalise() then carries the upstream gradient's strides as c.w
    alan_tensor_t dummy = d.factor[0];      // an output is required syntactically; its strides are not used
    if (canonicalise(tmp, keep | plate, red, dummy, bp.c, plate, &d.lse_out) != ALAN_OK) return ALAN_ERR_UNSUPPORTED;
    bp.c.o.dtype = ALAN_F32;
    bp.rp = plan_rows(bp.c, ALAN_MODE_LSE, ALAN_F32);
    if (!bp.rp.ok) return ALAN_ERR_UNSUPPORTED;
    const Canon &c = bp.c;
    bp.k0 = (c.nk > 0 && c.kplate[0]) ? 1 : 0;
    for (int f = 0; f < c.nf; ++f) {
        if (f == c.dominant) continue;
        for (int j = bp.k0; j < c.nk; ++j)
            if (c.f[f].ks[j] != 0) return ALAN_ERR_UNSUPPORTED;   // a factor varying over the window's rows
        if (c.f[f].rs[0] > INT32_MAX || c.f[f].rs[0] < INT32_MIN) return ALAN_ERR_UNSUPPORTED;
        if (b.grad[f].data) bp.want_part = true;
    }
    const uint32_t P = bp.rp.P;
    const uint32_t nch = std::max(1u, std::min(P, 2048u / std::max(1u, bp.rp.n_windows)));
    bp.p_chunk = (P + nch - 1) / nch;
    bp.n_chunks = (P + bp.p_chunk - 1) / bp.p_chunk;
    bp.part_bytes = bp.want_part ? (size_t)bp.rp.n_windows * P * bp.rp.L * sizeof(float) : 0;
    return ALAN_OK;
}

}  // namespace

extern "C" size_t alan_reduce_backward_workspace_bytes(const alan_backward_desc_t *b) {
    if (!b) return 0;
    BwdPlan bp;
    if (plan_backward(*b, bp) != ALAN_OK) return 0;
    return (bp.part_bytes + 255) & ~(size_t)255;
}

extern "C" int alan_reduce_backward(const alan_backward_desc_t *b, void *workspace, size_t workspace_bytes,
                                    void *stream_) {
    if (!b) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    BwdPlan bp;
    int rc = plan_backward(*b, bp);
    if (rc != ALAN_OK) return rc;
    if (bp.want_part && (!workspace || workspace_bytes < bp.part_bytes)) return ALAN_ERR_WORKSPACE;
    const Canon &c = bp.c;
    const RowsPlan &rp = bp.rp;
    const int k0 = bp.k0;

    RowsBwdDesc d;
    std::memset(&d, 0, sizeof(d));
    d.F = (const float *)c.f[c.dominant].p;
    d.gF = (float *)const_cast<void *>(b->grad[c.dominant].data);
    d.total = c.n_out * rp.L;
    d.L = rp.L;
    d.RB = rp.RB;
    d.NO = rp.NO;
    d.P = rp.P;
    d.p_chunk = bp.p_chunk;
    d.nki = c.nk - k0;
    for (int j = 0; j < d.nki; ++j) {
        d.kdiv[j] = make_fastdiv((uint32_t)c.ksize[k0 + j]);
        d.lks[j] = c.l.ks[k0 + j];
        d.gks[j] = c.w.ks[k0 + j];
    }
    d.lse = (const float *)c.l.p;
    d.l_ps = k0 ? c.l.ks[0] : 0;
    d.G = (const float *)c.w.p;
    d.part = bp.want_part ? (float *)workspace : nullptr;
    for (int f = 0; f < c.nf; ++f) {
        if (f == c.dominant) continue;
        const int i = d.nshared++;
        d.sh_p[i] = (const float *)c.f[f].p;
        d.sh_ps[i] = k0 ? c.f[f].ks[0] : 0;
        d.sh_rs[i] = (int32_t)c.f[f].rs[0];
    }
    const int nparts = std::max(1, 256 / rp.L);
    d.gs_off = (int)((((int64_t)rp.RB * rp.L + 3 + 3) / 4) * 4);
    d.cs_off = d.gs_off + ((rp.L + 3) / 4) * 4;
    const size_t lds_bytes = (size_t)(d.cs_off + nparts * rp.L + 4) * sizeof(float);
    if (lds_bytes > 64 * 1024) return ALAN_ERR_UNSUPPORTED;

    if (d.gF || d.part) {
        const dim3 grid(rp.n_windows, bp.n_chunks), block(256);
        switch (rp.logG) {
            case 0: ALAN_LAUNCH(rows_backward_kernel<0>, grid, block, lds_bytes, stream, d); break;
            case 1: ALAN_LAUNCH(rows_backward_kernel<1>, grid, block, lds_bytes, stream, d); break;
            case 2: ALAN_LAUNCH(rows_backward_kernel<2>, grid, block, lds_bytes, stream, d); break;
            default: ALAN_LAUNCH(rows_backward_kernel<3>, grid, block, lds_bytes, stream, d); break;
        }
        if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    }

    // ---- grad of every small factor: part[window][p][j] summed over the windows and over the dims it lacks
    for (int f = 0; f < c.nf; ++f) {
        if (f == c.dominant || !b->grad[f].data) continue;
        const bool has_p = k0 && c.f[f].ks[0] != 0, has_j = c.f[f].rs[0] != 0;
        Canon s2;
        s2.nf = 1;
        s2.dominant = 0;
        s2.f[0].p = workspace;
        s2.f[0].dtype = ALAN_F32;
        s2.f[0].scale = 1.f;
        s2.w.p = nullptr;
        s2.l.p = nullptr;
        s2.o.p = b->grad[f].data;
        s2.o.dtype = ALAN_F32;
        s2.o.scale = 1.f;
        s2.nk = s2.nr = 0;
        s2.n_out = s2.n_red = 1;
        auto add_dim = [&](bool keep_dim, int64_t size, int64_t src_stride, int64_t out_stride) {
            if (size <= 1) return;
            if (keep_dim) {
                s2.ksize[s2.nk] = size;
                s2.f[0].ks[s2.nk] = src_stride;
                s2.o.ks[s2.nk] = out_stride;
                s2.kplate[s2.nk] = false;
                ++s2.nk;
                s2.n_out *= size;
            } else {
                s2.rsize[s2.nr] = size;
                s2.f[0].rs[s2.nr] = src_stride;
                ++s2.nr;
                s2.n_red *= size;
            }
        };
        for (int j = 0; j < MAXD; ++j) s2.f[0].ks[j] = s2.f[0].rs[j] = s2.o.ks[j] = 0;
        // outermost -> innermost: window, p, j
        add_dim(false, rp.n_windows, (int64_t)rp.P * rp.L, 0);
        add_dim(has_p, rp.P, rp.L, has_p ? c.f[f].ks[0] : 0);
        add_dim(has_j, rp.L, 1, has_j ? c.f[f].rs[0] : 0);
        s2.red_contig = !has_j && rp.L > 1;
        s2.keep_contig = has_j;
        GroupDesc gd;
        GroupLaunch gl;
        rc = plan_group(s2, ALAN_F32, 0.0, gd, gl);
        if (rc != ALAN_OK) return rc;
        rc = try_launch_small(s2, gd, gl, ALAN_MODE_SUM, ALAN_F32, stream, EvPair());
        if (rc == ALAN_ERR_UNSUPPORTED) rc = launch_group(gd, gl, ALAN_MODE_SUM, ALAN_F32, stream);
        if (rc != ALAN_OK) return rc;
    }
    return ALAN_OK;
}
