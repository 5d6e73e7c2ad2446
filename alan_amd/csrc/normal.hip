// normal_outer kernel: register-blocked fast path of alan_reduce(mode = ALAN_MODE_NORMAL), the fused
// PRODUCER of the big log-prob factor (replaces TorchDimDist.py:127-162 + utils.py:147-152 for
// td.Normal when value, loc and scale carry DISJOINT first-class dims -- e.g. movielens
//   z[plate_1, K_z, d] ~ Normal(mu_z[K_mu, d], exp(psi_z)[K_psi, d])  ->  F[plate_1, K_mu, K_psi, K_z]).
//
//   out[v, l, s] = -sum_e (value[v,e] - loc[l,e])^2 * (0.5 / scale[s,e]^2) - sum_e log scale[s,e] - E log sqrt(2 pi)
//
// (same arithmetic as torch.distributions.Normal.log_prob: no expanded square, so no cancellation.)
// A thread owns one value row in registers and walks the (loc row, scale row) cross product; loc/scale
// rows are workgroup-uniform, so they come from LDS as broadcast reads: ~1.4 issue slots per (output, e).
// The kernel is bound by the store of F (HBM write), not by arithmetic.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "plan.h"

namespace alan {

struct NormalDesc {
    const float *val, *loc, *scl;
    float *out;
    int32_t E, Ep;                 // event length, padded to a multiple of 4
    uint32_t NV, NL, NS, l_chunk;
    int32_t nv;
    FastDiv vdiv[MAXD];
    int64_t v_vs[MAXD], v_os[MAXD];   // value / out strides over the value's keep dims
    int64_t l_rs, s_rs;               // row strides of loc / scale (elements)
    int64_t l_os, s_os;               // out strides along the loc / scale dims
    int32_t dbg;                      // tuning-only: 1 = no stores, 2 = no arithmetic
};

template <int EMAX>
__global__ __launch_bounds__(256) void normal_outer_kernel(const NormalDesc d) {
    extern __shared__ __align__(16) float lds[];
    const int E = d.E, Ep = d.Ep;
    float *w = lds;                       // [NS][Ep], zero padded
    float *lg = w + (size_t)d.NS * Ep;    // [NS]
    float *mu = lg + ((d.NS + 3) & ~3u);  // [l_chunk][Ep]
    const int tid = threadIdx.x;
    const uint32_t l0 = blockIdx.y * d.l_chunk;
    const uint32_t l1 = min(d.NL, l0 + d.l_chunk);

    // ---- workgroup-uniform tables (one pass over the scale rows: w = 0.5/sigma^2 and, by LDS float
    //      atomics, lg[s] = sum_e log sigma + E log sqrt(2 pi))
    for (uint32_t is = tid; is < d.NS; is += 256) lg[is] = (float)E * 0.91893853320467274178f;
    for (uint32_t i = tid; i < (l1 - l0) * (uint32_t)Ep; i += 256) {
        const uint32_t il = i / Ep, e = i - il * Ep;
        mu[i] = (int)e < E ? d.loc[(int64_t)(l0 + il) * d.l_rs + e] : 0.f;
    }
    __syncthreads();
    for (uint32_t i = tid; i < d.NS * (uint32_t)Ep; i += 256) {
        const uint32_t is = i / Ep, e = i - is * Ep;
        float v = 0.f;
        if ((int)e < E) {
            const float sc = d.scl[(int64_t)is * d.s_rs + e];
            v = 0.5f / (sc * sc);
            atomicAdd(&lg[is], __logf(sc));
        }
        w[i] = v;
    }

    // ---- this thread's value row
    const uint32_t r = blockIdx.x * 256u + tid;
    const bool active = r < d.NV;
    int64_t voff = 0, ooff = 0;
    {
        uint32_t o = active ? r : d.NV - 1u;
        for (int k = d.nv - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.vdiv[k]);
            const int64_t idx = (int64_t)(o - q * d.vdiv[k].d);
            o = q;
            voff += idx * d.v_vs[k];
            ooff += idx * d.v_os[k];
        }
    }
    float v[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) v[e] = (e < E) ? d.val[voff + e] : 0.f;
    __syncthreads();

    for (uint32_t il = l0; il < l1; ++il) {
        if (d.dbg == 3) {   // tuning-only: prologue cost
            if (active && v[0] == 12345.678f) d.out[ooff] = v[1];
            break;
        }
        const float4 *m4 = reinterpret_cast<const float4 *>(mu + (size_t)(il - l0) * Ep);
        float dd[EMAX];
#pragma unroll
        for (int q = 0; q < EMAX / 4; ++q) {
            if (q * 4 < Ep) {
                const float4 m = m4[q];
                const float a = v[4 * q] - m.x, b = v[4 * q + 1] - m.y, c = v[4 * q + 2] - m.z,
                            e = v[4 * q + 3] - m.w;
                dd[4 * q] = a * a;
                dd[4 * q + 1] = b * b;
                dd[4 * q + 2] = c * c;
                dd[4 * q + 3] = e * e;
            }
        }
        float *orow = d.out + ooff + (int64_t)il * d.l_os;
        for (uint32_t is = 0; is < d.NS; ++is) {
            const float4 *w4 = reinterpret_cast<const float4 *>(w + (size_t)is * Ep);
            float acc = 0.f;
            if (d.dbg != 2) {
#pragma unroll
                for (int q = 0; q < EMAX / 4; ++q) {
                    if (q * 4 < Ep) {
                        const float4 ww = w4[q];
                        acc = fmaf(dd[4 * q], ww.x, acc);
                        acc = fmaf(dd[4 * q + 1], ww.y, acc);
                        acc = fmaf(dd[4 * q + 2], ww.z, acc);
                        acc = fmaf(dd[4 * q + 3], ww.w, acc);
                    }
                }
            }
            if (d.dbg == 1) {
                if (active && acc == 12345.678f) orow[(int64_t)is * d.s_os] = acc;
            } else if (active) {
                orow[(int64_t)is * d.s_os] = -acc - lg[is];
            }
        }
    }
}

// Coalesced-store variant.  The plain kernel's per-thread scalar stores reach HBM as Vi-float (120-byte)
// pieces and the kernel is store-ISSUE bound (25 us at K=30 for a 32 MB output).  When the scale dim and
// the value's innermost dim are adjacent in the output (movielens: out[m, k_mu, (k_psi, k_z)] = 900
// contiguous floats per (m, k_mu)), a workgroup instead takes RB = G*Vi value rows (G whole groups of the
// innermost value dim), parks its results for one loc row in an LDS tile [NS][RB] and writes each group's
// NS*Vi-float region with full-width 16-byte stores.
struct NormalTWDesc {
    NormalDesc b;
    int32_t Vi, G, RB, nvo;        // innermost value dim, groups / rows per workgroup, outer value dims
    FastDiv odiv[MAXD], vidiv;
    int64_t o_vs[MAXD], o_os[MAXD];
    int64_t vi_vs;                 // value stride of the innermost value dim
    uint32_t NG;                   // number of groups = NV / Vi
};

template <int EMAX>
__global__ __launch_bounds__(256) void normal_outer_tw_kernel(const NormalTWDesc D) {
    const NormalDesc &d = D.b;
    extern __shared__ __align__(16) float lds[];
    const int E = d.E, Ep = d.Ep;
    float *w = lds;                         // [NS][Ep]
    float *lg = w + (size_t)d.NS * Ep;      // [NS]
    float *mu = lg + ((d.NS + 3) & ~3u);    // [Ep]
    float *tile = mu + Ep;                  // [NS][RB]
    const int tid = threadIdx.x;
    const int RB = D.RB, Vi = D.Vi;

    for (uint32_t is = tid; is < d.NS; is += 256) lg[is] = (float)E * 0.91893853320467274178f;
    __syncthreads();
    for (uint32_t i = tid; i < d.NS * (uint32_t)Ep; i += 256) {
        const uint32_t is = i / Ep, e = i - is * Ep;
        float v = 0.f;
        if ((int)e < E) {
            const float sc = d.scl[(int64_t)is * d.s_rs + e];
            v = 0.5f / (sc * sc);
            atomicAdd(&lg[is], __logf(sc));
        }
        w[i] = v;
    }

    const uint32_t g0 = blockIdx.x * (uint32_t)D.G;
    const uint32_t gl = fd_div((uint32_t)tid, D.vidiv);
    const int vi = tid - (int)gl * Vi;
    const bool active = tid < RB && (g0 + gl) < D.NG;
    int64_t voff = 0;
    {
        uint32_t o = active ? g0 + gl : 0u;
        for (int k = D.nvo - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, D.odiv[k]);
            voff += (int64_t)(o - q * D.odiv[k].d) * D.o_vs[k];
            o = q;
        }
        voff += (int64_t)vi * D.vi_vs;
    }
    float v[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) v[e] = (active && e < E) ? d.val[voff + e] : 0.f;

    const uint32_t l0 = blockIdx.y * d.l_chunk;
    const uint32_t l1 = min(d.NL, l0 + d.l_chunk);
    const int region4 = ((int)d.NS * Vi) >> 2;      // 16-byte pieces per (group, loc row)
    for (uint32_t il = l0; il < l1; ++il) {
        __syncthreads();                            // tables ready / previous tile drained
        for (int e = tid; e < Ep; e += 256) mu[e] = e < E ? d.loc[(int64_t)il * d.l_rs + e] : 0.f;
        __syncthreads();
        float dd[EMAX];
        const float4 *m4 = reinterpret_cast<const float4 *>(mu);
#pragma unroll
        for (int q = 0; q < EMAX / 4; ++q) {
            if (q * 4 < Ep) {
                const float4 m = m4[q];
                const float a = v[4 * q] - m.x, b = v[4 * q + 1] - m.y, c = v[4 * q + 2] - m.z,
                            e = v[4 * q + 3] - m.w;
                dd[4 * q] = a * a;
                dd[4 * q + 1] = b * b;
                dd[4 * q + 2] = c * c;
                dd[4 * q + 3] = e * e;
            }
        }
        if (tid < RB) {
            for (uint32_t is = 0; is < d.NS; ++is) {
                const float4 *w4 = reinterpret_cast<const float4 *>(w + (size_t)is * Ep);
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < EMAX / 4; ++q) {
                    if (q * 4 < Ep) {
                        const float4 ww = w4[q];
                        acc = fmaf(dd[4 * q], ww.x, acc);
                        acc = fmaf(dd[4 * q + 1], ww.y, acc);
                        acc = fmaf(dd[4 * q + 2], ww.z, acc);
                        acc = fmaf(dd[4 * q + 3], ww.w, acc);
                    }
                }
                tile[is * RB + tid] = -acc - lg[is];
            }
        }
        __syncthreads();
        // ---- write-out: group g -> out[group base + il*l_os + (is*Vi + x)], contiguous, 16 bytes per lane
        for (int g = 0; g < D.G; ++g) {
            if (g0 + g >= D.NG) break;
            int64_t obase = (int64_t)il * d.l_os;
            uint32_t o = g0 + g;
            for (int k = D.nvo - 1; k >= 0; --k) {
                const uint32_t q = fd_div(o, D.odiv[k]);
                obase += (int64_t)(o - q * D.odiv[k].d) * D.o_os[k];
                o = q;
            }
            float4 *dst = reinterpret_cast<float4 *>(d.out + obase);
            const float *src = tile + g * Vi;
            for (int f = tid; f < region4; f += 256) {
                uint32_t is = fd_div((uint32_t)(4 * f), D.vidiv);
                int x = 4 * f - (int)is * Vi;
                float r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r[u] = src[is * RB + x];
                    if (++x == Vi) {
                        x = 0;
                        ++is;
                    }
                }
                dst[f] = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    }
}

// Returns ALAN_ERR_UNSUPPORTED when the canonical problem is not an outer-product Normal.
int try_launch_normal_outer(const Canon &c, hipStream_t stream, const EvPair &ev) {
    if (c.nf != 3 || c.nr != 1) return ALAN_ERR_UNSUPPORTED;
    for (int f = 0; f < 3; ++f)
        if (c.f[f].dtype != ALAN_F32 || c.f[f].rs[0] != 1) return ALAN_ERR_UNSUPPORTED;
    if (c.o.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    const int64_t E = c.rsize[0];
    if (E < 1 || E > 64) return ALAN_ERR_UNSUPPORTED;

    NormalDesc d;
    std::memset(&d, 0, sizeof(d));
    int nL = 0, nS = 0;
    int64_t NV = 1;
    d.NL = d.NS = 1;
    for (int j = 0; j < c.nk; ++j) {
        const bool inV = c.f[0].ks[j] != 0, inL = c.f[1].ks[j] != 0, inS = c.f[2].ks[j] != 0;
        if ((int)inV + (int)inL + (int)inS != 1) return ALAN_ERR_UNSUPPORTED;  // shared or pure-broadcast dim
        if (inV) {
            d.vdiv[d.nv] = make_fastdiv((uint32_t)c.ksize[j]);
            d.v_vs[d.nv] = c.f[0].ks[j];
            d.v_os[d.nv] = c.o.ks[j];
            ++d.nv;
            NV *= c.ksize[j];
        } else if (inL) {
            if (nL++) return ALAN_ERR_UNSUPPORTED;
            d.NL = (uint32_t)c.ksize[j];
            d.l_rs = c.f[1].ks[j];
            d.l_os = c.o.ks[j];
        } else {
            if (nS++) return ALAN_ERR_UNSUPPORTED;
            d.NS = (uint32_t)c.ksize[j];
            d.s_rs = c.f[2].ks[j];
            d.s_os = c.o.ks[j];
        }
    }
    if (NV * d.NL * d.NS < 65536) return ALAN_ERR_UNSUPPORTED;  // tiny: the generic kernel is fine
    d.val = (const float *)c.f[0].p;
    d.loc = (const float *)c.f[1].p;
    d.scl = (const float *)c.f[2].p;
    d.out = (float *)const_cast<void *>(c.o.p);
    d.E = (int)E;
    d.Ep = (int)((E + 3) & ~3);
    d.NV = (uint32_t)NV;
    if (const char *e = getenv("ALAN_NORMAL_DBG")) d.dbg = atoi(e);

    // ---- coalesced-store variant: scale dim and innermost value dim adjacent in the output, every
    //      (group, loc row) region 16-byte aligned
    {
        int vin = -1;
        for (int j = 0; j < d.nv; ++j)
            if (d.v_os[j] == 1) vin = j;
        const int64_t Vi = vin >= 0 ? (int64_t)d.vdiv[vin].d : 0;
        bool ok = vin >= 0 && nS == 1 && d.s_os == Vi && Vi <= 128 && (NV % Vi) == 0 && ((d.NS * Vi) % 4) == 0 &&
                  (d.l_os % 4) == 0 && (reinterpret_cast<uintptr_t>(d.out) & 15) == 0;
        for (int j = 0; j < d.nv && ok; ++j)
            if (j != vin) ok = (d.v_os[j] % 4) == 0;
        if (const char *e = getenv("ALAN_NORMAL_TW")) ok = ok && atoi(e) != 0;   // tuning knob
        if (ok) {
            NormalTWDesc D;
            std::memset(&D, 0, sizeof(D));
            D.b = d;
            D.Vi = (int)Vi;
            D.vidiv = make_fastdiv((uint32_t)Vi);
            D.G = (int)std::max<int64_t>(1, 256 / Vi);
            D.RB = D.G * D.Vi;
            D.NG = (uint32_t)(NV / Vi);
            D.vi_vs = d.v_vs[vin];
            for (int j = 0; j < d.nv; ++j) {
                if (j == vin) continue;
                D.odiv[D.nvo] = d.vdiv[j];
                D.o_vs[D.nvo] = d.v_vs[j];
                D.o_os[D.nvo] = d.v_os[j];
                ++D.nvo;
            }
            const uint32_t gx2 = (D.NG + D.G - 1) / D.G;
            uint32_t gy2 = std::min<uint32_t>(d.NL, std::max<uint32_t>(1, 16384 / std::max(1u, gx2)));
            if (const char *e = getenv("ALAN_NORMAL_GY")) gy2 = std::min<uint32_t>(d.NL, std::max(1, atoi(e)));
            D.b.l_chunk = (d.NL + gy2 - 1) / gy2;
            gy2 = (d.NL + D.b.l_chunk - 1) / D.b.l_chunk;
            const size_t lds2 = ((size_t)d.NS * d.Ep + ((d.NS + 3) & ~3u) + d.Ep + (size_t)d.NS * D.RB) * sizeof(float);
            if (lds2 <= 40 * 1024) {   // keep >= 4 workgroups per CU
                const dim3 grid2(gx2, gy2), block2(256);
                if (d.Ep <= 8)
                    hipExtLaunchKernelGGL(normal_outer_tw_kernel<8>, grid2, block2, (uint32_t)lds2, stream, ev.start, ev.stop, 0, D);
                else if (d.Ep <= 16)
                    hipExtLaunchKernelGGL(normal_outer_tw_kernel<16>, grid2, block2, (uint32_t)lds2, stream, ev.start, ev.stop, 0, D);
                else if (d.Ep <= 32)
                    hipExtLaunchKernelGGL(normal_outer_tw_kernel<32>, grid2, block2, (uint32_t)lds2, stream, ev.start, ev.stop, 0, D);
                else
                    hipExtLaunchKernelGGL(normal_outer_tw_kernel<64>, grid2, block2, (uint32_t)lds2, stream, ev.start, ev.stop, 0, D);
                return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
            }
        }
    }

    const uint32_t gx = (uint32_t)((NV + 255) / 256);
    // enough workgroups to fill the chip: split the loc rows over grid.y
    // one loc row per workgroup is fastest (more waves to hide the read -> FMA chain -> store latency per scale
    // row; measured 25 us at gy = NL vs 310 us at gy = 1 for K = 30): split the loc rows as far as the grid allows
    uint32_t gy = std::min<uint32_t>(d.NL, std::max<uint32_t>(1, 16384 / std::max(1u, gx)));
    if (const char *e = getenv("ALAN_NORMAL_GY")) gy = std::min<uint32_t>(d.NL, std::max(1, atoi(e)));   // tuning knob
    d.l_chunk = (d.NL + gy - 1) / gy;
    gy = (d.NL + d.l_chunk - 1) / d.l_chunk;
    const size_t lds = ((size_t)d.NS * d.Ep + ((d.NS + 3) & ~3u) + (size_t)d.l_chunk * d.Ep) * sizeof(float);
    if (lds > 64 * 1024) return ALAN_ERR_UNSUPPORTED;

    const dim3 grid(gx, gy), block(256);
    if (d.Ep <= 8)
        hipExtLaunchKernelGGL(normal_outer_kernel<8>, grid, block, (uint32_t)lds, stream, ev.start, ev.stop, 0, d);
    else if (d.Ep <= 16)
        hipExtLaunchKernelGGL(normal_outer_kernel<16>, grid, block, (uint32_t)lds, stream, ev.start, ev.stop, 0, d);
    else if (d.Ep <= 32)
        hipExtLaunchKernelGGL(normal_outer_kernel<32>, grid, block, (uint32_t)lds, stream, ev.start, ev.stop, 0, d);
    else
        hipExtLaunchKernelGGL(normal_outer_kernel<64>, grid, block, (uint32_t)lds, stream, ev.start, ev.stop, 0, d);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

}  // namespace alan
