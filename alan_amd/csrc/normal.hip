// The outer-product Normal producer: fast paths of alan_reduce(mode = ALAN_MODE_NORMAL / NORMAL_LOGSCALE), the fused
// PRODUCER of the big log-prob factor.  Two kernels: normal_mfma_kernel (matrix cores, the default while the factor
// fits the Infinity Cache; further down) and normal_outer_kernel (vector unit, register-blocked; below).
// Replaces TorchDimDist.py:127-162 + utils.py:147-152 for td.Normal when value, loc and scale carry DISJOINT first-class dims -- e.g. movielens
//   z[plate_1, K_z, d] ~ Normal(mu_z[K_mu, d], exp(psi_z)[K_psi, d])  ->  F[plate_1, K_mu, K_psi, K_z]).
//
//   out[v, l, s] = -sum_e (value[v,e] - loc[l,e])^2 * (0.5 / scale[s,e]^2) - sum_e log scale[s,e] - E log sqrt(2 pi)
//
// (same arithmetic as torch.distributions.Normal.log_prob: no expanded square, so no cancellation.)
// normal_outer_kernel: a thread owns one value row in registers and walks the (loc row, scale row) cross product; loc/scale
// rows are workgroup-uniform, so they come from LDS as broadcast reads: ~1.4 issue slots per (output, e).
// The event length is a template parameter (tables zero-padded to it): the LDS reads of a scale row are then
// straight-line and issue back to back -- with run-time bounds every 16-byte read sat in its own basic block
// behind an s_waitcnt, and the kernel ran at LDS latency (26 us at K=30 instead of 17).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "plan.h"

namespace alan {

// Kernel argument (by value); the scalars every wave needs first sit in the leading two 64-byte lines.
struct NormalDesc {
    const float *val, *loc, *scl;
    float *out;
    int32_t E, Ep;                 // event length; Ep = the kernel's EMAX (table row stride)
    uint32_t NV, NL, NS, l_chunk;
    int32_t nv;
    int32_t log_scale;                // scl holds log(scale)
    int32_t rows_contig;              // value rows of a workgroup are one contiguous, 16-byte aligned block
    int32_t vstage_off;               // LDS offset (floats) of the value staging area
    float out_scale, add_const;       // out = out_scale * log_prob + add_const
    uint32_t out_bytes;               // bytes spanned by out (MFMA kernel: buffer descriptor range)
    uint32_t ts_nk;                   // MFMA kernel, transposed stores: size of the value's innermost dim (a wave owns
                                      // one index of the outer value dims); 0 = lanes store their own rows
    int64_t l_rs, s_rs;               // row strides of loc / scale (elements)
    int64_t l_os, s_os;               // out strides along the loc / scale dims
    FastDiv vdiv[MAXD];
    int64_t v_vs[MAXD], v_os[MAXD];   // value / out strides over the value's keep dims
#ifdef ALAN_ABLATE
    int32_t dbg;                      // ablation knob (ALAN_NORMAL_ABLATE): 1 prologue only, 2 no stores, 3 one scale row
#endif
};

// R = value rows per thread (rows t, t+256, ... of the workgroup's block of 256*R): every 16-byte read of a
// scale row from LDS then feeds 4*R FMAs.  At R = 1 the LDS return path (5 broadcast ds_read_b128 per 20
// FMAs) is what bounds the main loop, not the FMAs.
template <int EMAX, int R>
__global__ __launch_bounds__(256) void normal_outer_kernel(const NormalDesc d) {
    extern __shared__ __align__(16) float lds[];
    const int E = d.E;
    constexpr int Ep = EMAX;              // table row stride: the event length padded to the template's
    float *w = lds;                       // [NS][Ep], zero padded
    float *lg = w + (size_t)d.NS * Ep;    // [NS]
    float *mu = lg + ((d.NS + 3) & ~3u);  // [l_chunk][Ep]
    float *lgs = mu + (size_t)d.l_chunk * Ep;   // [NS][Ep] log(scale), zero padded (prologue only)
    const int tid = threadIdx.x;
    const uint32_t l0 = blockIdx.y * d.l_chunk;
    const uint32_t l1 = min(d.NL, l0 + d.l_chunk);

    // ---- workgroup-uniform tables: every scale element is loaded once, by its own thread (all loads in
    // flight together); the per-row log-normaliser is then summed from LDS
#pragma unroll 4
    for (uint32_t i = tid; i < d.NS * (uint32_t)Ep; i += 256) {
        const uint32_t is = i / Ep, e = i - is * Ep;
        float v = 0.f, lgv = 0.f;
        if ((int)e < E) {
            const float sc = d.scl[(int64_t)is * d.s_rs + e];
            v = d.log_scale ? 0.5f * expf(-2.f * sc) : 0.5f / (sc * sc);
            lgv = d.log_scale ? sc : logf(sc);
        }
        w[i] = v;
        lgs[i] = lgv;
    }
    for (uint32_t i = tid; i < (l1 - l0) * (uint32_t)Ep; i += 256) {
        const uint32_t il = i / Ep, e = i - il * Ep;
        mu[i] = (int)e < E ? d.loc[(int64_t)(l0 + il) * d.l_rs + e] : 0.f;
    }

    // ---- this thread's value rows
    bool active[R];
    int64_t ooff[R];
    float v[R][EMAX];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t r0 = (blockIdx.x * R + j) * 256u;          // first row of sub-block j
        const uint32_t r = r0 + tid;
        active[j] = r < d.NV;
        int64_t voff = 0;
        ooff[j] = 0;
        uint32_t o = active[j] ? r : d.NV - 1u;
        for (int k = d.nv - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.vdiv[k]);
            const int64_t idx = (int64_t)(o - q * d.vdiv[k].d);
            o = q;
            voff += idx * d.v_vs[k];
            ooff[j] += idx * d.v_os[k];
        }
        if (d.rows_contig) {
            // the sub-block's 256 rows are one contiguous run: stream it into LDS with 16-byte loads (each
            // thread reading its own row from global memory costs one cache line per lane per load: the
            // texture addresser, not HBM, then bounds the prologue)
            float *vs = lds + d.vstage_off;
            if (j > 0) __syncthreads();                           // the previous sub-block has been read
            if (r0 < d.NV) {
                const uint32_t nrow = min(256u, d.NV - r0);
                const int n4 = (int)((nrow * (uint32_t)E + 3u) >> 2);
                const float4 *src = reinterpret_cast<const float4 *>(d.val + (int64_t)r0 * E);
                float4 *dst = reinterpret_cast<float4 *>(vs);
                const int64_t lim4 = ((int64_t)d.NV * E + 3) >> 2;    // 16-byte units in the whole value tensor
                const int64_t base4 = ((int64_t)r0 * E) >> 2;
                for (int i = tid; i < n4; i += 256)
                    if (base4 + i < lim4) dst[i] = src[i];         // (torch pads allocations beyond 16 bytes)
            }
            __syncthreads();
            const uint32_t last = r0 < d.NV ? min(256u, d.NV - r0) - 1u : 0u;
            const float *row = vs + (size_t)min((uint32_t)tid, last) * E;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const float x = row[min(e, E - 1)];
                v[j][e] = e < E ? x : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const float x = d.val[voff + min(e, E - 1)];   // branch-free: every load is issued, pad slots zeroed
                v[j][e] = e < E ? x : 0.f;
            }
        }
    }
    __syncthreads();
    for (uint32_t is = tid; is < d.NS; is += 256) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) s += lgs[(size_t)is * Ep + e];   // (pad slots hold 0)
        lg[is] = s + (float)E * 0.91893853320467274178f;
    }
    __syncthreads();

#ifdef ALAN_ABLATE
    if (d.dbg == 1) {   // ablation: prologue only
        float q = lg[tid % d.NS];
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int e = 0; e < EMAX; ++e) q += v[j][e];
        if (active[0]) d.out[ooff[0] + (int64_t)l0 * d.l_os] = q;
        return;
    }
    const uint32_t ns_run = d.dbg == 3 ? 1u : d.NS;
    const bool store_on = d.dbg != 2;
#else
    const uint32_t ns_run = d.NS;
    constexpr bool store_on = true;
#endif
    bool do_store[R];
#pragma unroll
    for (int j = 0; j < R; ++j) do_store[j] = active[j] && store_on;
    for (uint32_t il = l0; il < l1; ++il) {
        const float4 *m4 = reinterpret_cast<const float4 *>(mu + (size_t)(il - l0) * Ep);
        float dd[R][EMAX];
#pragma unroll
        for (int q = 0; q < EMAX / 4; ++q) {
            const float4 m = m4[q];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float a = v[j][4 * q] - m.x, b = v[j][4 * q + 1] - m.y, c = v[j][4 * q + 2] - m.z,
                            e = v[j][4 * q + 3] - m.w;
                dd[j][4 * q] = a * a;
                dd[j][4 * q + 1] = b * b;
                dd[j][4 * q + 2] = c * c;
                dd[j][4 * q + 3] = e * e;
            }
        }
        for (uint32_t is = 0; is < ns_run; ++is) {
            const float4 *w4 = reinterpret_cast<const float4 *>(w + (size_t)is * Ep);
            float4 ww[EMAX / 4];
#pragma unroll
            for (int q = 0; q < EMAX / 4; ++q) ww[q] = w4[q];      // straight-line: all reads issue together
            const float lgi = lg[is];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < EMAX / 4; ++q) {
                    acc = fmaf(dd[j][4 * q], ww[q].x, acc);
                    acc = fmaf(dd[j][4 * q + 1], ww[q].y, acc);
                    acc = fmaf(dd[j][4 * q + 2], ww[q].z, acc);
                    acc = fmaf(dd[j][4 * q + 3], ww[q].w, acc);
                }
                const float res = (-acc - lgi) * d.out_scale + d.add_const;
                if (do_store[j] || (!store_on && res == 12345.678f))
                    d.out[ooff[j] + (int64_t)il * d.l_os + (int64_t)is * d.s_os] = res;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The same producer on the matrix cores.  For one loc row l the block out[., l, .] IS a GEMM over the event dim,
//   out[v, l, s] = -( sum_e w[s,e] * d2[v,e] + lg[s] ),   d2[v,e] = (value[v,e] - loc[l,e])^2,  w = 0.5 / scale^2,
// and v_mfma_f32_32x32x2_f32 computes it as an exact k-ordered fp32 fmaf chain (same numerics as the loop above) at
// the packed-FMA rate WITHOUT the operand traffic: the vector kernel's main loop is bound by its broadcast LDS reads
// of the scale rows, here A (32 scale rows x 2 events per instruction) and B (2 events x 32 value rows) are one VGPR
// each, held for the whole kernel (A) or rebuilt with two VALU ops per loc row (B).  A wave owns 32 value rows and all
// scale rows (NST tiles of 32); the log-normaliser lg[s] enters as the C operand of the first MFMA.  No LDS, no
// barriers: every wave is independent.  D[i = scale row][j = value row]: lanes run along the value rows, which is
// the output's contiguous dim in the plate step (K_z), so each store instruction writes two runs of up to 128 bytes.
typedef float f32x16 __attribute__((ext_vector_type(16)));

//
// TS (transposed stores; NST = 1): when the value's innermost dim kz (<= 32 long) is also the output's, with the scale
// dim next (the plate step's F[plate_1, K_mu, K_psi, K_z]), the [NS, kz] block a wave produces per loc row is ONE
// contiguous run of the output.  The wave then owns exactly one index of the outer value dims (lanes = kz), passes
// the tile through a private 4 KB of LDS in the output's own order and writes it with 16 bytes per lane: 4 store
// instructions of up to 1 KB each instead of 16 of 2 x 120 bytes whose rows straddle cache lines.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int EH, int NST, bool TS>      // EH = ceil(E / 2) MFMA steps; NST = tiles of 32 scale rows
__global__ __launch_bounds__(256) void normal_mfma_kernel(const NormalDesc d) {
    constexpr int EP = 2 * EH, NSP = 32 * NST;
    static_assert(!TS || NST == 1, "transposed stores: one scale tile");
    __shared__ float wt[NSP * EP];            // 0.5 / scale^2, zero padded to [NSP][EP]
    __shared__ float lgt[NSP * EP];           // log scale, zero padded; lgt[s * EP] becomes the row's log-normaliser
    __shared__ __align__(16) float tbuf[TS ? 4 * 1024 : 4];   // TS: a [NS][kz] tile per wave
    extern __shared__ __align__(16) float mut[];   // [l_chunk][EP] this workgroup's loc rows, zero padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int E = d.E;
    const uint32_t l0 = blockIdx.y * d.l_chunk, l1 = min(d.NL, l0 + d.l_chunk);
    // ---- workgroup-uniform tables: every scale element once per workgroup (one thread each), summed per row after
#pragma unroll 2
    for (int i = tid; i < NSP * EP; i += 256) {
        const uint32_t srow = i / EP;
        const int e = i - srow * EP;
        float wv = 0.f, lgv = 0.f;
        if (srow < d.NS && e < E) {
            const float sc = d.scl[(int64_t)srow * d.s_rs + e];
            wv = d.log_scale ? 0.5f * expf(-2.f * sc) : 0.5f / (sc * sc);
            lgv = d.log_scale ? sc : logf(sc);
        }
        wt[i] = wv;
        lgt[i] = lgv;
    }
    for (int i = tid; i < (int)(l1 - l0) * EP; i += 256) {
        const int il = i / EP, e = i - il * EP;
        mut[i] = e < E ? d.loc[(int64_t)(l0 + il) * d.l_rs + e] : 0.f;
    }
    __syncthreads();
    float rowsum = 0.f;
    if (tid < NSP) {
#pragma unroll
        for (int e = 0; e < EP; ++e) rowsum += lgt[tid * EP + e];
    }
    __syncthreads();
    if (tid < NSP) lgt[tid * EP] = rowsum + (float)E * 0.91893853320467274178f;
    __syncthreads();
    uint32_t vr;
    bool active;
    if (TS) {
        const uint32_t w = blockIdx.x * 4u + wave;             // index over the outer value dims
        if (w * d.ts_nk >= d.NV) return;
        vr = w * d.ts_nk + (uint32_t)j;
        active = (uint32_t)j < d.ts_nk;
    } else {
        const uint32_t v0 = (blockIdx.x * 4u + wave) * 32u;
        if (v0 >= d.NV) return;                   // (no barriers below)
        vr = v0 + j;
        active = vr < d.NV;
    }
    int64_t voff = 0, ooff = 0;
    {
        uint32_t o = active ? vr : d.NV - 1u;
        for (int k = d.nv - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.vdiv[k]);
            const int64_t idx = (int64_t)(o - q * d.vdiv[k].d);
            o = q;
            voff += idx * d.v_vs[k];
            ooff += idx * d.v_os[k];
        }
    }
    // ---- A operand: lane (j, h) holds w[32 st + j][2 step + h]; C operand of the first MFMA: the log-normaliser of
    // the 16 rows this lane's accumulator registers stand for
    float wreg[NST][EH];
    f32x16 cinit[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int step = 0; step < EH; ++step) wreg[st][step] = wt[(32 * st + j) * EP + 2 * step + h];
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[st][r] = lgt[(32 * st + (r & 3) + 8 * (r >> 2) + 4 * h) * EP];
    }
    // ---- this lane's half of its value row (pad slots 0, like the loc table's)
    float z[EH];
#pragma unroll
    for (int step = 0; step < EH; ++step) {
        const int e = 2 * step + h;
        const float x = d.val[voff + min(e, E - 1)];
        z[step] = e < E ? x : 0.f;
    }
#ifdef ALAN_ABLATE
    const bool store_on = d.dbg != 2, mfma_on = d.dbg != 4;
    if (d.dbg == 1) {
        if (z[0] + wreg[0][0] + cinit[0][0] == 12345.678f) d.out[0] = 1.f;
        return;
    }
#else
    constexpr bool store_on = true, mfma_on = true;
#endif
    // Buffer stores: address = descriptor base + (wave-uniform byte offset of the (loc row, scale tile), an SGPR) +
    // (this lane's byte offset, ONE VGPR stepped from scale row to scale row).  With flat stores the compiler kept 16
    // 64-bit row addresses per tile in vector registers (312 VGPRs at NST = 4: one wave per SIMD).  Lanes without a
    // value row, and scale rows beyond NS, get an offset beyond the descriptor's range: the hardware drops the store.
    const uint32_t row_bstride = (uint32_t)d.s_os * 4u;
    const uint32_t OOB = 0x80000000u;          // >= the descriptor's range with or without the SGPR offset added (out spans < 2 GiB)
    const uint32_t lane_boff = active ? (uint32_t)(ooff + 4 * h * d.s_os) * 4u : OOB;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)d.out_bytes, 0x00020000);
    // TS: the wave's block of the output starts at lane 0's element (kz = 0); n4 16-byte pieces, lane + 64 q each
    float *tb = tbuf + (TS ? wave * 1024 : 0);
    const uint32_t wave_boff = TS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)ooff * 4u)) : 0u;
    const uint32_t n4 = TS ? (d.NS * d.ts_nk) >> 2 : 0u;
    for (uint32_t il = l0; il < l1; ++il) {
        const float *lp = mut + (size_t)(il - l0) * EP + h;
        float dd[EH];
#pragma unroll
        for (int step = 0; step < EH; ++step) {
            const float t = z[step] - lp[2 * step];
            dd[step] = t * t;
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x16 acc = cinit[st];
            if (mfma_on) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[st][0], dd[0], cinit[st], 0, 0, 0);
#pragma unroll
                for (int step = 1; step < EH; ++step)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[st][step], dd[step], acc, 0, 0, 0);
            } else {
                acc[0] += dd[0] + dd[EH - 1];
            }
            const uint32_t tile_boff = (uint32_t)((int64_t)il * d.l_os + (int64_t)(32 * st) * d.s_os) * 4u;
            if (!store_on) {
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q += acc[r];
                if (q == 12345.678f) d.out[0] = q;
            } else if (TS) {
                const int rows_ok = (int)d.NS - 4 * h;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();               // the previous tile's reads are issued (LDS is in order)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    if (active && row < rows_ok) tb[(row + 4 * h) * (int)d.ts_nk + j] = -acc[r] * d.out_scale + d.add_const;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const uint32_t blk = wave_boff + (uint32_t)((int64_t)il * d.l_os) * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i4 = (uint32_t)lane + 64u * q;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(tb + 4 * (i4 < n4 ? i4 : 0u));
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i4 < n4 ? i4 * 16u : OOB, blk, 0);
                }
            } else {
                const int rows_ok = (int)d.NS - 32 * st - 4 * h;      // this lane's rows (r&3) + 8 (r>>2) below it exist
                uint32_t vo = lane_boff;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(uint32_t, -acc[r] * d.out_scale + d.add_const), rsrc,
                        row < rows_ok ? vo : OOB, tile_boff, 0);
                    if (lane_boff != OOB) vo += (r & 3) == 3 ? 5u * row_bstride : row_bstride;
                }
            }
        }
    }
}

template <int EH>
static void launch_normal_mfma(int nst, dim3 grid, hipStream_t stream, const EvPair &ev, const NormalDesc &d) {
    const uint32_t lds = d.l_chunk * 2u * EH * sizeof(float);        // the loc rows of a workgroup
    if (nst == 1 && d.ts_nk)
        ALAN_LAUNCH_EXT((normal_mfma_kernel<EH, 1, true>), grid, dim3(256), lds, stream, ev.start, ev.stop, 0, d);
    else if (nst == 1)
        ALAN_LAUNCH_EXT((normal_mfma_kernel<EH, 1, false>), grid, dim3(256), lds, stream, ev.start, ev.stop, 0, d);
    else if (nst == 2)
        ALAN_LAUNCH_EXT((normal_mfma_kernel<EH, 2, false>), grid, dim3(256), lds, stream, ev.start, ev.stop, 0, d);
    else
        ALAN_LAUNCH_EXT((normal_mfma_kernel<EH, 4, false>), grid, dim3(256), lds, stream, ev.start, ev.stop, 0, d);
}

// d is filled except for l_chunk.  Declines (false) outside E <= 32, NS <= 128.
static bool try_normal_mfma(NormalDesc d, int64_t NV, hipStream_t stream, const EvPair &ev) {
    static const int mfma_knob = env_knob("ALAN_NORMAL_MFMA");                                   // ablation knob
    const bool off = mfma_knob == 0;
    if (off || d.E > 32 || d.NS > 128) return false;
    {   // lanes address the output by a 32-bit element offset from a wave-uniform base
        int64_t span = (int64_t)(d.NL - 1) * d.l_os + (int64_t)(d.NS - 1) * d.s_os;
        for (int k = 0; k < d.nv; ++k) span += (int64_t)(d.vdiv[k].d - 1) * d.v_os[k];
        if (d.l_os < 0 || d.s_os < 0 || span >= (1ll << 29) - 8) return false;
        d.out_bytes = (uint32_t)((span + 1) * 4);
        // beyond the 256 MiB Infinity Cache the factor streams to HBM, where this kernel's 2 x 128-byte runs per store
        // reach 2.3 TB/s against the vector kernel's 3.0 (K=100 unsplit, 1.2 GB); inside it, 4.2 against 2.9
        const bool force = mfma_knob == 2;
        if (d.out_bytes > (224u << 20) && !force) return false;
        for (int k = 0; k < d.nv; ++k)
            if (d.v_os[k] < 0) return false;
    }
    const int nst = d.NS <= 32 ? 1 : d.NS <= 64 ? 2 : 4;
    {   // transposed stores (see the kernel): the [NS, kz] block of one (outer value index, loc row) is one 16-byte
        // aligned contiguous run of the output
        const int in = d.nv - 1;
        // (worth it when the rows straddle cache lines and most lanes have one: measured 14.1 -> 12.8 us at kz = 30,
        // but 13.3 -> 14.2 at kz = 32, whose 128-byte rows are already whole lines, and 4.5 -> 5.7 at kz = 10)
        bool ts = nst == 1 && d.nv >= 1 && d.v_os[in] == 1 && d.vdiv[in].d >= 24 && d.vdiv[in].d < 32 &&
                  d.s_os == (int64_t)d.vdiv[in].d && (d.NS * d.vdiv[in].d) % 4 == 0 && d.l_os % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(d.out) % 16 == 0;
        for (int k = 0; ts && k < in; ++k) ts = d.v_os[k] % 4 == 0;
        d.ts_nk = ts ? d.vdiv[in].d : 0u;
    }
    const uint32_t gx = d.ts_nk ? (uint32_t)((NV / d.ts_nk + 3) / 4) : (uint32_t)((NV + 127) / 128);
    // ONE residency round: the workgroups that fit the chip at once (256 CUs x the waves per SIMD the kernel's
    // registers allow: 204 at NST = 4, 84 at NST = 1), each walking as many loc rows as that takes.  (600 workgroups on
    // 512 slots ran two rounds: 25 us of MFMA phase for 11 us of MFMAs.)
    const uint32_t slots = 256u * (nst == 4 ? 2u : nst == 2 ? 3u : 4u);
    uint32_t gy = std::min<uint32_t>(d.NL, std::max<uint32_t>(1, slots / std::max(1u, gx)));
    d.l_chunk = (d.NL + gy - 1) / gy;
    gy = (d.NL + d.l_chunk - 1) / d.l_chunk;
    const dim3 grid(gx, gy);
    const int eh = (d.E + 1) / 2;
    switch (eh) {
        case 1: launch_normal_mfma<1>(nst, grid, stream, ev, d); break;
        case 2: launch_normal_mfma<2>(nst, grid, stream, ev, d); break;
        case 3: launch_normal_mfma<3>(nst, grid, stream, ev, d); break;
        case 4: launch_normal_mfma<4>(nst, grid, stream, ev, d); break;
        case 5: launch_normal_mfma<5>(nst, grid, stream, ev, d); break;
        case 6: launch_normal_mfma<6>(nst, grid, stream, ev, d); break;
        case 7: case 8: launch_normal_mfma<8>(nst, grid, stream, ev, d); break;
        case 9: launch_normal_mfma<9>(nst, grid, stream, ev, d); break;
        case 10: launch_normal_mfma<10>(nst, grid, stream, ev, d); break;
        case 11: case 12: launch_normal_mfma<12>(nst, grid, stream, ev, d); break;
        default: launch_normal_mfma<16>(nst, grid, stream, ev, d);
    }
    return true;
}

// Returns ALAN_ERR_UNSUPPORTED when the canonical problem is not an outer-product Normal.
int try_launch_normal_outer(const Canon &c, bool log_scale, float out_scale, double add_const, hipStream_t stream,
                            const EvPair &ev, bool dry) {
    if (c.nf != 3 || c.nr != 1) return ALAN_ERR_UNSUPPORTED;
    if (c.f[0].scale != 1.f || c.f[1].scale != 1.f || c.f[2].scale == 2.f) return ALAN_ERR_UNSUPPORTED;   // weighted / scaled / flagged terms: generic kernel
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
    if (dry) return ALAN_OK;
    d.val = (const float *)c.f[0].p;
    d.loc = (const float *)c.f[1].p;
    d.scl = (const float *)c.f[2].p;
    d.out = (float *)const_cast<void *>(c.o.p);
    d.E = (int)E;
    static const int kEmax[] = {4, 8, 12, 16, 20, 24, 28, 32, 48, 64};
    d.Ep = 64;
    for (int em : kEmax)
        if (E <= em) {
            d.Ep = em;      // = the kernel's template parameter: table row stride
            break;
        }
    d.NV = (uint32_t)NV;
    d.log_scale = log_scale ? 1 : 0;
    d.out_scale = out_scale;
    d.add_const = (float)add_const;
#ifdef ALAN_ABLATE
    static const int ablate_knob = env_knob("ALAN_NORMAL_ABLATE");
    if (ablate_knob != ENV_UNSET) d.dbg = ablate_knob;
#endif
    if (try_normal_mfma(d, NV, stream, ev)) return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    {   // value rows contiguous in row-index order?  (voff(r) = r * E)
        bool contig = reinterpret_cast<uintptr_t>(d.val) % 16 == 0;   // (a workgroup's block starts 1 KiB-aligned)
        int64_t run = E;
        for (int k = d.nv - 1; k >= 0; --k) {
            contig = contig && d.v_vs[k] == run;
            run *= d.vdiv[k].d;
        }
        d.rows_contig = contig ? 1 : 0;
    }

    // two value rows per thread once that still leaves >= 512 workgroups
    int R = (NV * (int64_t)d.NL >= 512ll * 512) ? 2 : 1;
    const uint32_t gx = (uint32_t)((NV + 256 * R - 1) / (256 * R));
    // enough workgroups to fill the chip: split the loc rows over grid.y, one loc row per workgroup when the
    // grid allows (K=30: 25 us with gy = NL, 310 us with gy = 1).  Measured budget at K=30, R=2 (HIP events,
    // 4.7 us of which is the event floor): tables + value staging 3.6 us, FMAs/LDS 4-6 us, stores of F 6 us.
    uint32_t gy = std::min<uint32_t>(d.NL, std::max<uint32_t>(1, 16384 / std::max(1u, gx)));
    d.l_chunk = (d.NL + gy - 1) / gy;
    gy = (d.NL + d.l_chunk - 1) / d.l_chunk;
    size_t lds_f = 2 * (size_t)d.NS * d.Ep + ((d.NS + 3) & ~3u) + (size_t)d.l_chunk * d.Ep;
    if (d.rows_contig) {
        d.vstage_off = (int)((lds_f + 3) & ~(size_t)3);
        lds_f = d.vstage_off + 256 * (size_t)d.E + 4;
    }
    const size_t lds = lds_f * sizeof(float);
    if (lds > 64 * 1024) return ALAN_ERR_UNSUPPORTED;

    const dim3 grid(gx, gy), block(256);
#define ALAN_NORMAL_CASE(EM)                                                                                     \
    case EM:                                                                                                     \
        if (R == 2)                                                                                              \
            ALAN_LAUNCH_EXT((normal_outer_kernel<EM, 2>), grid, block, (uint32_t)lds, stream, ev.start,     \
                                  ev.stop, 0, d);                                                                \
        else                                                                                                     \
            ALAN_LAUNCH_EXT((normal_outer_kernel<EM, 1>), grid, block, (uint32_t)lds, stream, ev.start,     \
                                  ev.stop, 0, d);                                                                \
        break
    switch (d.Ep) {
        ALAN_NORMAL_CASE(4);
        ALAN_NORMAL_CASE(8);
        ALAN_NORMAL_CASE(12);
        ALAN_NORMAL_CASE(16);
        ALAN_NORMAL_CASE(20);
        ALAN_NORMAL_CASE(24);
        ALAN_NORMAL_CASE(28);
        ALAN_NORMAL_CASE(32);
        ALAN_NORMAL_CASE(48);
        default: ALAN_LAUNCH_EXT((normal_outer_kernel<64, 1>), grid, block, (uint32_t)lds, stream, ev.start, ev.stop, 0, d);
    }
#undef ALAN_NORMAL_CASE
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

}  // namespace alan
