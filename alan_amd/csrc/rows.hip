// rows kernel: the LDS-staged fast path of alan_reduce for the dominant shape of the hot path --
// the reduced K dim is the contiguous (innermost) dim of the largest factor, e.g. movielens
// F[plate_1, K_mu, K_psi, K_z] (+ g[plate_1, K_z]) reduced over K_z, then summed over plate_1
// (logpq.py:128,149; reduce_Ks.py:249-251; utils.py:218-220).
//
// Data movement (HBM-bound; ~4 B and ~10 VALU ops per element):
//   * a workgroup owns a window of RB consecutive rows and walks a chunk of the plate; per plate
//     index it copies the window's contiguous slab (RB*L floats) HBM -> LDS with 16-byte loads,
//     fully coalesced whatever L is (rows of 30 floats are only 8-byte aligned);
//   * thread t (or G = 2/4/8 lanes for long rows) then owns row t: it reads the row from LDS ONCE into <= 32
//     registers per lane -- straight-line reads with compile-time offsets, slots past the row's end masked --
//     adds the window-constant small factors (staged once per slab in LDS), takes the exact row max and sums
//     exp(x - max) from the registers: the reference's two-pass arithmetic, so -inf / NaN corner cases fall
//     out identically; row stride L words with 8-byte reads is conflict-free for L = 2 (mod 4) (K = 10, 30)
//     and 2-way for K = 100; rows whose stride is 0 (mod 16 words) are read as rotated float4 quads;
//   * the plate sum is accumulated in a register across the chunk; each workgroup writes one partial
//     per row, and a tiny second stage adds the chunks (deterministic: no float atomics).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "plan.h"

namespace alan {

// Kernel argument (by value).  Field ORDER matters: a wave's first global load depends on a chain of scalar
// loads from this struct, one cache-line miss each -- everything the common path (one or two window-constant
// small factors, partials or a direct store) needs sits in the first two 64-byte lines.
struct RowsDesc {
    // ---- line 0
    const float *F;
    int64_t total;          // elements of F reachable from F (tail guard for 16-byte loads)
    int32_t L, RB;
    uint32_t NO, P, p_chunk;
    int32_t nshared, ngen, nki, gs_off;
    float add_const;
    float *partial;         // per-chunk partials [n_chunks, NO] (n_chunks > 1) ...
    // ---- line 1
    void *out;              // ... or the final output (fp32, or fp64 when a small factor is fp64: see plan_rows)
    void *lse;              // optional per-row log-sum-exp values (same dtype as out)
    int64_t l_ps;
    const void *sh_p[2];    // first two "shared" secondary factors: constant over the window (no inner keep dims)
    int64_t sh_ps[2];       //   stride along the plate dim
    int32_t sh_rs[2];       //   stride along the row
    int32_t f64_mask;       // bit i: shared factor i is fp64; bit 30: out / lse are fp64
    // ---- the rest
    const void *shx_p[MAXF];    // further shared factors (rare)
    int64_t shx_ps[MAXF];
    int32_t shx_rs[MAXF];
    FastDiv kdiv[MAXD];     // inner keep dims
    int64_t oks[MAXD];
    int64_t lks[MAXD];
    KTensor gen[MAXF];      // general secondary factors: ks over INNER keep dims, rs[0] along the row
    int64_t gen_ps[MAXF];
#ifdef ALAN_ABLATE
    int32_t dbg;            // ablation knob (ALAN_ROWS_ABLATE): 1 = no reduction, 2 = no global loads
#endif
};

#ifndef ROWS_NT
#define ROWS_NT 1
#endif
constexpr int ROWS_UNR = 8;
typedef float f32x4 __attribute__((ext_vector_type(4)));  // 16-byte loads in flight per thread (one slab <= 256*8*16 B = 32 KiB)

// PF = prefetch the next slab into registers while the current one is reduced (workgroups that walk a
// chunk of the plate).  With one slab per workgroup (the literal movielens size: 1200 workgroups) the
// staging registers are dead during the reduction, the kernel fits 5 workgroups per CU and the whole
// grid is resident in a single round.
template <int MODE, int LOGG, bool VEC2, bool ROT, bool GEN, bool PF, bool SHORT>
__global__ __launch_bounds__(256, 4) void rows_kernel(const RowsDesc d) {
    extern __shared__ __align__(16) float lds[];
    constexpr int G = 1 << LOGG;
    constexpr int NS = SHORT ? 8 : 16;   // float2 slots per lane: a lane holds <= 32 (SHORT: 16) row elements
    const int t = threadIdx.x;
    const int L = d.L;
#ifdef ALAN_ABLATE
    if (d.dbg == 4) return;
#endif
    const uint32_t o0 = blockIdx.x * (uint32_t)d.RB;
    const uint32_t nrows = min((uint32_t)d.RB, d.NO - o0);
    const uint32_t p0 = blockIdx.y * d.p_chunk;
    const uint32_t p1 = min(d.P, p0 + d.p_chunk);
    const int gl = t & (G - 1);
    const bool has_row = (uint32_t)(t >> LOGG) < nrows;
    const int r = has_row ? (t >> LOGG) : (int)nrows - 1;
    float *gs = lds + d.gs_off;

    // ---- this thread's row: offsets of the general secondary factors (fixed across the plate).  The output /
    // lse offsets are recomputed where they are used (a few multiply-highs) rather than held in registers
    // through the reduction: the one-slab variant runs at a 96-VGPR cap and used to spill ~20 dwords per
    // thread -- 24 MB of scratch writes per launch at the literal movielens size.
    int64_t sb[GEN ? MAXF : 1];
#pragma unroll
    for (int f = 0; f < (GEN ? MAXF : 1); ++f) sb[f] = 0;
    if (GEN) {
        uint32_t o = o0 + (uint32_t)r;
        for (int k = d.nki - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.kdiv[k]);
            const int64_t idx = (int64_t)(o - q * d.kdiv[k].d);
            o = q;
#pragma unroll
            for (int f = 0; f < MAXF; ++f)
                if (f < d.ngen) sb[f] += idx * d.gen[f].ks[k];
        }
    }
    auto row_offset = [&](const int64_t *strides) {
        int64_t off = 0;
        uint32_t o = o0 + (uint32_t)r;
        for (int k = d.nki - 1; k >= 0; --k) {
            const uint32_t q = fd_div(o, d.kdiv[k]);
            off += (int64_t)(o - q * d.kdiv[k].d) * strides[k];
            o = q;
        }
        return off;
    };

    // slab of plate index p: 16-byte loads from a 16-byte aligned-down start into registers
    f32x4 v[ROWS_UNR];
    auto slab_fetch = [&](uint32_t p) {
        const int64_t e0 = ((int64_t)p * d.NO + o0) * L;
        const int64_t a0 = e0 & ~(int64_t)3;
        const int n4 = ((int)(e0 - a0) + (int)nrows * L + 3) >> 2;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(d.F + a0);
#ifdef ALAN_ABLATE
        if (d.dbg >= 2) return;
#endif
        if (a0 + 4 * (int64_t)n4 <= d.total) {  // whole slab inside the tensor (all but the very last one)
#pragma unroll
            for (int u = 0; u < ROWS_UNR; ++u) {
                // clamped, not predicated: straight-line loads with all 8 in flight (threads past the
                // slab's end re-read its last 16 bytes)
                const int i = min(u * 256 + t, n4 - 1);
                v[u] = ROWS_NT ? __builtin_nontemporal_load(src + i) : src[i];   // read-once stream
            }
        }
        // else: the 16-byte loads of the last slab would run past the tensor; slab_store copies it
        // element-wise straight into LDS instead (no staging registers)
    };
    auto slab_store = [&](uint32_t p) {
        const int64_t e0 = ((int64_t)p * d.NO + o0) * L;
        const int64_t a0 = e0 & ~(int64_t)3;
        const int shift = (int)(e0 - a0);
        const int n4 = (shift + (int)nrows * L + 3) >> 2;
        if (a0 + 4 * (int64_t)n4 <= d.total) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
            for (int u = 0; u < ROWS_UNR; ++u) {
                const int i = min(u * 256 + t, n4 - 1);   // duplicates rewrite the same 16 bytes
                dst[i] = v[u];
            }
        } else {
            const int n = (int)(d.total - a0);
            for (int i = t; i < n; i += 256) lds[i] = d.F[a0 + i];
        }
    };

    // the window-constant small factors of plate index p: thread j < L holds sum_f sh_f[p, j]
    // The first two stay PENDING loads until the next slab is staged (adding them here would put an
    // s_waitcnt vmcnt(0) right behind the slab prefetch and serialise it with the reduction); further ones
    // (rare) are added at once, in a rolled loop -- unrolled over MAXF the compiler hoists six 64-bit base
    // addresses into the prologue and spills them.
    float gn0 = 0.f, gn1 = 0.f, gn_rest = 0.f;
    auto shared_fetch = [&](uint32_t p) {
        gn0 = gn1 = gn_rest = 0.f;
        if (d.nshared > 0 && t < L) {
            gn0 = load_as<float>(d.sh_p[0], d.f64_mask & 1, (int64_t)p * d.sh_ps[0] + (int64_t)t * d.sh_rs[0]);
            if (d.nshared > 1)
                gn1 = load_as<float>(d.sh_p[1], (d.f64_mask >> 1) & 1, (int64_t)p * d.sh_ps[1] + (int64_t)t * d.sh_rs[1]);
#pragma unroll 1
            for (int f = 2; f < d.nshared; ++f)
                gn_rest += load_as<float>(d.shx_p[f], (d.f64_mask >> f) & 1, (int64_t)p * d.shx_ps[f] + (int64_t)t * d.shx_rs[f]);
        }
    };

    float acc = 0.f;
    if (PF) {
        slab_fetch(p0);
        shared_fetch(p0);
    }
    for (uint32_t p = p0; p < p1; ++p) {
        if (!PF) {
            slab_fetch(p);
            shared_fetch(p);
        }
        const int64_t e0 = ((int64_t)p * d.NO + o0) * L;
        const int shift = (int)(e0 & 3);
        slab_store(p);
        if (d.nshared > 0 && t < L) gs[t] = (gn0 + gn1) + gn_rest;
        __syncthreads();
        if (PF && p + 1 < p1) {   // in flight while this slab is reduced
            slab_fetch(p + 1);
            shared_fetch(p + 1);
        }

        const float *row = lds + shift + r * L;
        // ---- the row (+ small factors) into registers, ONE pass over LDS: lane gl of the row's G lanes takes
        // elements gl, gl+G, ... (<= 32 of them); slots past the row's end are masked, not skipped, so the
        // reads are straight-line with compile-time offsets and all in flight together
        float x[2 * NS];
        constexpr float NEG = -__builtin_huge_valf();
        constexpr float PAD = MODE == ALAN_MODE_LSE ? NEG : 0.f;
        if (VEC2) {
            const float2 *r2 = reinterpret_cast<const float2 *>(row) + gl;
            const float2 *g2 = reinterpret_cast<const float2 *>(gs) + gl;
            const int nq = (L >> 1) - gl;
            // straight-line batches of 8 slots (row values, then the small factors): every read of a batch is
            // in flight before the first use -- with the add inside one loop each pair of reads sat behind
            // its own s_waitcnt and the pass ran at LDS latency
            constexpr int HB = PF ? 8 : 4;   // (the one-slab variant has fewer registers to spare)
#pragma unroll
            for (int h = 0; h < NS / HB; ++h) {
                float2 a[HB];
#pragma unroll
                for (int u = 0; u < HB; ++u) a[u] = r2[(h * HB + u) * G];   // (reads past the row stay inside the allocation)
                if (d.nshared > 0) {
                    float2 g[HB];
#pragma unroll
                    for (int u = 0; u < HB; ++u) g[u] = g2[(h * HB + u) * G];
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
                        a[u].x += g[u].x;
                        a[u].y += g[u].y;
                    }
                }
#pragma unroll
                for (int u = 0; u < HB; ++u) {
                    const bool ok = (h * HB + u) * G < nq;
                    x[2 * (h * HB + u)] = ok ? a[u].x : PAD;
                    x[2 * (h * HB + u) + 1] = ok ? a[u].y : PAD;
                }
            }
        } else {
            if (GEN) {
                // general small factors (they vary over the window's rows): folded into this lane's own
                // elements of the LDS row first, with a rolled loop (global loads, L1/L2-resident)
                float *wrow = lds + shift + r * L;
#pragma unroll 2
                for (int j = gl; j < L; j += G) {
                    float a = wrow[j];
#pragma unroll
                    for (int f = 0; f < MAXF; ++f)
                        if (f < d.ngen)
                            a += ((const float *)d.gen[f].p)[sb[f] + (int64_t)p * d.gen_ps[f] +
                                                             (int64_t)j * d.gen[f].rs[0]];
                    if (has_row) wrow[j] = a;   // (threads past the window's last row mirror it: no write)
                }
            }
            if (ROT) {
                // row stride = 0 (mod 16 words): every lane would start on the same bank.  Rows are then
                // 16-byte aligned, so read float4 quads, each row starting at a different (rotated) quad
                const int NQ = L >> 2;
                const float4 *r4 = reinterpret_cast<const float4 *>(row);
                const float4 *g4 = reinterpret_cast<const float4 *>(gs);
                int qq = (gl + r) % NQ;
#pragma unroll
                for (int u = 0; u < NS / 2; ++u) {
                    float4 a = r4[qq];
                    if (d.nshared > 0) {
                        const float4 g = g4[qq];
                        a.x += g.x, a.y += g.y, a.z += g.z, a.w += g.w;
                    }
                    const bool ok = gl + u * G < NQ;
                    x[4 * u] = ok ? a.x : PAD;
                    x[4 * u + 1] = ok ? a.y : PAD;
                    x[4 * u + 2] = ok ? a.z : PAD;
                    x[4 * u + 3] = ok ? a.w : PAD;
                    qq += G;
                    if (qq >= NQ) qq -= NQ;
                }
            } else {
                const float *r1 = row + gl, *g1 = gs + gl;
#pragma unroll
                for (int u = 0; u < 2 * NS; ++u) {
                    float a = r1[u * G];
                    if (d.nshared > 0) a += g1[u * G];
                    x[u] = gl + u * G < L ? a : PAD;
                }
            }
        }
        float val;
#ifdef ALAN_ABLATE
        if (d.dbg == 1 || d.dbg == 3) {
            val = x[0];
        } else
#endif
        if (MODE == ALAN_MODE_LSE) {
            float m = NEG;   // exact row max, then sum exp(x - max): the reference's two-pass arithmetic
#pragma unroll
            for (int u = 0; u < 2 * NS; ++u) m = fmaxf(m, x[u]);
#pragma unroll
            for (int ofs = G >> 1; ofs > 0; ofs >>= 1) m = fmaxf(m, __shfl_xor(m, ofs));
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                s0 += __expf(x[2 * u] - m);     // utils.py:219   (masked slots: exp(-inf) = 0)
                s1 += __expf(x[2 * u + 1] - m);
            }
            float s = s0 + s1;
#pragma unroll
            for (int ofs = G >> 1; ofs > 0; ofs >>= 1) s += __shfl_xor(s, ofs);
            val = logf(s + Num<float>::eps) + m;  // utils.py:220
        } else {  // ALAN_MODE_SUM
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                s0 += x[2 * u];
                s1 += x[2 * u + 1];
            }
            float s = s0 + s1;
#pragma unroll
            for (int ofs = G >> 1; ofs > 0; ofs >>= 1) s += __shfl_xor(s, ofs);
            val = s;
        }
        if (has_row && gl == 0) {
            if (d.lse) store_as<float>(d.lse, (d.f64_mask >> 30) & 1, row_offset(d.lks) + (int64_t)p * d.l_ps, val);
            acc += val;
        }
        __syncthreads();
    }
    if (has_row && gl == 0) {
        if (d.partial)
            d.partial[(int64_t)blockIdx.y * d.NO + o0 + r] = acc;
        else
            store_as<float>(d.out, (d.f64_mask >> 30) & 1, row_offset(d.oks), acc + d.add_const);
    }
}

static int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }

RowsPlan plan_rows(const Canon &c, int mode, int compute_dtype) {
    RowsPlan rp;
    if (mode != ALAN_MODE_LSE && mode != ALAN_MODE_SUM) return rp;
    (void)compute_dtype;   // the kernel computes in fp32 = the dominant factor's precision (see below)
    if (c.nr != 1 || !c.red_contig) return rp;
    const int64_t L = c.rsize[0];
    if (L < 8 || L > 256) return rp;
    // dtypes: the dominant factor must be fp32.  Window-constant small factors may be fp64 (real data sets carry
    // fp64 observations, so the likelihood factor of an fp32 model arrives in fp64): they are converted on load,
    // the reduction runs in fp32 -- the precision of the factor that holds all but a per-mille of the values --
    // and the result is stored in the promoted dtype the caller expects.  (The reference would run such a step in
    // fp64 throughout: a difference of fp32 rounding, inside the 1e-4 ELBO tolerance.)
    const KTensor &dom = c.f[c.dominant];
    if (dom.dtype != ALAN_F32) return rp;
    const int k0p = (c.nk > 0 && c.kplate[0]) ? 1 : 0;
    for (int f = 0; f < c.nf; ++f) {
        if (f == c.dominant || c.f[f].dtype == ALAN_F32) continue;
        for (int j = k0p; j < c.nk; ++j)
            if (c.f[f].ks[j] != 0) return rp;     // an fp64 factor that varies over the window's rows: not here
    }
    if (c.l.p && c.l.dtype != c.o.dtype) return rp;
    if ((reinterpret_cast<uintptr_t>(dom.p) & 15) != 0) return rp;
    if (dom.scale != 1.f) return rp;
    int64_t run = L;  // the dominant factor must be dense with rows of L
    for (int j = c.nk - 1; j >= 0; --j) {
        if (dom.ks[j] != run) return rp;
        run *= c.ksize[j];
    }
    for (int f = 0; f < c.nf; ++f)
        if (f != c.dominant && c.f[f].scale != 1.f) return rp;
    // at most one (merged) plate dim, and it must be the outermost dim of the dominant factor
    int nplate = 0;
    for (int j = 0; j < c.nk; ++j) nplate += c.kplate[j] ? 1 : 0;
    if (nplate > 1 || (nplate == 1 && !c.kplate[0])) return rp;
    if (c.n_out * L < 16384) return rp;  // latency regime: the group kernel is as good

    rp.L = (int)L;
    rp.P = nplate ? (uint32_t)c.ksize[0] : 1u;
    rp.NO = (uint32_t)(c.n_out / rp.P);
    rp.logG = L <= 32 ? 0 : L <= 64 ? 1 : L <= 128 ? 2 : 3;
    rp.rot = gcd_i((int)L, 64) >= 16;
    if (rp.rot) while ((1 << rp.logG) > L / 4) --rp.logG;   // (the quad rotation wraps once per step)
    int rbmax = (int)std::min<int64_t>(256 >> rp.logG, (256 * ROWS_UNR * 4 - 4) / L);
    rp.n_windows = (rp.NO + rbmax - 1) / rbmax;
    rp.RB = (int)((rp.NO + rp.n_windows - 1) / rp.n_windows);
    rp.threads = 256;
    uint32_t target_blocks = 4096;
    static const int blocks_knob = env_knob("ALAN_ROWS_BLOCKS");                                       // tuning knob
    if (blocks_knob != ENV_UNSET) target_blocks = (uint32_t)std::max(1, blocks_knob);
    uint32_t nch = std::max(1u, std::min(rp.P, target_blocks / std::max(1u, rp.n_windows)));
    rp.p_chunk = (rp.P + nch - 1) / nch;
    rp.n_chunks = (rp.P + rp.p_chunk - 1) / rp.p_chunk;
    rp.gs_off = (int)((((int64_t)rp.RB * L + 3 + 3) / 4) * 4);
    // + slack: the unrolled row reads run up to 32*G floats past a row's start (masked, never used)
    rp.lds_bytes = (size_t)(rp.gs_off + ((L + 3) / 4) * 4 + 32 * (1 << rp.logG) + 8) * 4;
    rp.partial_bytes = (nplate && rp.n_chunks > 1) ? (size_t)rp.n_chunks * rp.NO * 4 : 0;
    rp.vec2 = !rp.rot && (L % 2 == 0);
    rp.ok = true;
    return rp;
}

int launch_rows(const Canon &c, const RowsPlan &rp, int mode, double add_const, void *workspace,
                size_t workspace_bytes, hipStream_t stream, const EvPair &ev) {
    if (!rp.ok) return ALAN_ERR_UNSUPPORTED;
    const bool fused = rp.P > 1 || (c.nk > 0 && c.kplate[0]);
    const bool two_stage = fused && rp.n_chunks > 1;
    if (two_stage && (!workspace || workspace_bytes < rp.partial_bytes)) return ALAN_ERR_WORKSPACE;

    RowsDesc d;
    std::memset(&d, 0, sizeof(d));
    const KTensor &dom = c.f[c.dominant];
    d.F = (const float *)dom.p;
    d.total = c.n_out * rp.L;
    d.L = rp.L;
    d.RB = rp.RB;
    d.NO = rp.NO;
    d.P = rp.P;
    d.p_chunk = rp.p_chunk;
    d.gs_off = rp.gs_off;
    const int k0 = (c.nk > 0 && c.kplate[0]) ? 1 : 0;  // first inner keep dim
    d.nki = c.nk - k0;
    for (int j = 0; j < d.nki; ++j) d.kdiv[j] = make_fastdiv((uint32_t)c.ksize[k0 + j]);
    for (int f = 0; f < c.nf; ++f) {
        if (f == c.dominant) continue;
        bool shared = true;  // constant over the window: no inner keep dim
        for (int j = 0; j < d.nki; ++j) shared = shared && c.f[f].ks[k0 + j] == 0;
        const int64_t ps = k0 ? c.f[f].ks[0] : 0;
        if (shared) {
            if (c.f[f].rs[0] > INT32_MAX || c.f[f].rs[0] < INT32_MIN) return ALAN_ERR_UNSUPPORTED;
            const int i = d.nshared++;
            if (i < 2) {
                d.sh_p[i] = c.f[f].p;
                if (c.f[f].dtype == ALAN_F64) d.f64_mask |= 1 << i;
                d.sh_ps[i] = ps;
                d.sh_rs[i] = (int32_t)c.f[f].rs[0];
            } else {
                d.shx_p[i] = c.f[f].p;
                if (c.f[f].dtype == ALAN_F64) d.f64_mask |= 1 << i;
                d.shx_ps[i] = ps;
                d.shx_rs[i] = (int32_t)c.f[f].rs[0];
            }
        } else {
            KTensor &dst = d.gen[d.ngen];
            dst.p = c.f[f].p;
            dst.dtype = c.f[f].dtype;
            dst.scale = 1.f;
            for (int j = 0; j < d.nki; ++j) dst.ks[j] = c.f[f].ks[k0 + j];
            dst.rs[0] = c.f[f].rs[0];
            d.gen_ps[d.ngen++] = ps;
        }
    }
    d.out = const_cast<void *>(c.o.p);
    if (c.o.dtype == ALAN_F64) d.f64_mask |= 1 << 30;
    for (int j = 0; j < d.nki; ++j) d.oks[j] = c.o.ks[k0 + j];
    d.partial = two_stage ? (float *)workspace : nullptr;
    d.lse = const_cast<void *>(c.l.p);
    for (int j = 0; j < d.nki; ++j) d.lks[j] = c.l.ks[k0 + j];
    d.l_ps = k0 ? c.l.ks[0] : 0;
    d.add_const = two_stage ? 0.f : (float)add_const;
#ifdef ALAN_ABLATE
    static const int ablate_knob = env_knob("ALAN_ROWS_ABLATE");
    if (ablate_knob != ENV_UNSET) d.dbg = ablate_knob;
#endif

    const dim3 grid(rp.n_windows, rp.n_chunks);
    const dim3 block(256);
#define ALAN_ROWS5(MODE, G, V, R, GN, PFV, SH)                                                                  \
    ALAN_LAUNCH_EXT((rows_kernel<MODE, G, V, R, GN, PFV, SH>), grid, block, rp.lds_bytes, stream, ev.start, \
                          ev.stop, 0, d)
#define ALAN_ROWS4(MODE, G, V, R, GN)                    \
    if (rp.p_chunk > 1) {                                \
        if (G == 0 && rp.L <= 16) ALAN_ROWS5(MODE, G, V, R, GN, true, (G == 0));  \
        else ALAN_ROWS5(MODE, G, V, R, GN, true, false);  \
    } else {                                             \
        if (G == 0 && rp.L <= 16) ALAN_ROWS5(MODE, G, V, R, GN, false, (G == 0)); \
        else ALAN_ROWS5(MODE, G, V, R, GN, false, false); \
    }
#define ALAN_ROWS3(MODE, G, V, R)                         \
    if (d.ngen > 0) { ALAN_ROWS4(MODE, G, false, R, true) } \
    else { ALAN_ROWS4(MODE, G, V, R, false) }
#define ALAN_ROWS2(MODE, V, R)                        \
    switch (rp.logG) {                                \
        case 0: ALAN_ROWS3(MODE, 0, V, R); break;     \
        case 1: ALAN_ROWS3(MODE, 1, V, R); break;     \
        case 2: ALAN_ROWS3(MODE, 2, V, R); break;     \
        default: ALAN_ROWS3(MODE, 3, V, R); break;    \
    }
    if (mode == ALAN_MODE_LSE) {
        if (rp.rot) { ALAN_ROWS2(ALAN_MODE_LSE, false, true) }
        else if (rp.vec2) { ALAN_ROWS2(ALAN_MODE_LSE, true, false) }
        else { ALAN_ROWS2(ALAN_MODE_LSE, false, false) }
    } else {
        if (rp.rot) { ALAN_ROWS2(ALAN_MODE_SUM, false, true) }
        else if (rp.vec2) { ALAN_ROWS2(ALAN_MODE_SUM, true, false) }
        else { ALAN_ROWS2(ALAN_MODE_SUM, false, false) }
    }
#undef ALAN_ROWS5
#undef ALAN_ROWS4
#undef ALAN_ROWS2
#undef ALAN_ROWS3
    if (hipGetLastError() != hipSuccess) return ALAN_ERR_LAUNCH;
    if (!two_stage) return ALAN_OK;

    // ---- second stage: out[o] = sum_chunk partial[chunk, o]  (+ add_const), via the group kernel
    Canon s2;
    s2.nk = d.nki;
    s2.nr = 1;
    s2.nf = 1;
    s2.dominant = 0;
    s2.rsize[0] = rp.n_chunks;
    s2.f[0].p = workspace;
    s2.f[0].dtype = ALAN_F32;
    s2.f[0].scale = 1.f;
    s2.f[0].rs[0] = rp.NO;
    int64_t st = 1;
    for (int j = d.nki - 1; j >= 0; --j) {
        s2.ksize[j] = c.ksize[k0 + j];
        s2.f[0].ks[j] = st;
        st *= s2.ksize[j];
        s2.o.ks[j] = c.o.ks[k0 + j];
        s2.kplate[j] = false;
    }
    s2.w.p = nullptr;
    s2.l.p = nullptr;
    s2.o.p = c.o.p;
    s2.o.dtype = c.o.dtype;
    s2.n_out = rp.NO;
    s2.n_red = rp.n_chunks;
    s2.red_contig = false;
    s2.keep_contig = true;
    GroupDesc gd;
    GroupLaunch gl;
    int rc = plan_group(s2, c.o.dtype, add_const, gd, gl);
    if (rc != ALAN_OK) return rc;
    rc = try_launch_small(s2, gd, gl, ALAN_MODE_SUM, ALAN_F32, stream, EvPair());
    if (rc != ALAN_ERR_UNSUPPORTED) return rc;
    return launch_group(gd, gl, ALAN_MODE_SUM, ALAN_F32, stream);
}

}  // namespace alan
