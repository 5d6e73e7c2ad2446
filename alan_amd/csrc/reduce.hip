// alan_reduce: fused multi-factor broadcast-add + reduction (log-sum-exp / sum / weighted exp-sum)
// over arbitrary strided factors.  Replaces reduce_Ks.py:249-251 (logsumexp_sum),
// utils.py:207-225 (logsumexp_dims / logmeanexp_dims), logpq.py:149-153 (plate sum) and the
// autograd backward of those.  gfx950 (CDNA4, wave64) only.
//
// Kernel families
//   group kernel   G = 2^g lanes (1..64) or a whole 256-thread block cooperate on ONE output
//                  element, striding over the flattened reduce index; any strides, any dtypes.
//                  Lanes of a group read consecutive reduce elements, consecutive groups read
//                  consecutive outputs, so whichever of the two is the dominant factor's contiguous
//                  dim gives coalesced loads.
//   rows kernel    (rows.hip) LDS-staged fast path for the dominant shape: reduce dim contiguous in
//                  the largest factor (movielens F[M,Ka,Kb,Kz] over Kz), optional fused plate sum.
#include <cstring>
#include <type_traits>

#include "common.h"
#include <cstddef>

#include "plan.h"
#include "small_device.h"

namespace alan {


// ------------------------------------------------------------------------------------------
template <typename T, int MODE, bool BLOCK>
__global__ __launch_bounds__(256) void reduce_group_kernel(const GroupDesc d, const int logG) {
    const uint32_t G = BLOCK ? 256u : (1u << logG);
    uint32_t grp, gl;
    if (BLOCK) {
        grp = blockIdx.x;
        gl = threadIdx.x;
    } else {
        const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
        grp = gid >> logG;
        gl = gid & (G - 1u);
    }
    const bool active = grp < d.n_out;
    uint32_t o = active ? grp : d.n_out - 1u;

    // ---- decompose the output index over the keep dims (innermost last), once per thread
    int64_t base[MAXF];
    int64_t wbase = 0, obase = 0;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) base[f] = 0;
    for (int k = d.nk - 1; k >= 0; --k) {
        const uint32_t q = fd_div(o, d.kdiv[k]);
        const int64_t idx = (int64_t)(o - q * d.kdiv[k].d);
        o = q;
#pragma unroll
        for (int f = 0; f < MAXF; ++f)
            if (f < d.nf) base[f] += idx * d.f[f].ks[k];
        if (MODE == ALAN_MODE_WEXPSUM) wbase += idx * d.w.ks[k];
        obase += idx * d.oks[k];
    }

    // ---- stream over the reduce index, UNR elements per step: their loads are all issued before the
    // first is consumed (small problems are pure latency chains; big ones want the loads in flight)
    T m = Num<T>::ninf(), s = T(0);
    auto stream = [&](auto unr_c) {
    constexpr int UNR = decltype(unr_c)::value;
    for (uint32_t r0 = gl; r0 < d.n_red; r0 += UNR * G) {
        T val[UNR][MAXF];
        T wv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t ru = r0 + (uint32_t)u * G;
            const uint32_t r = ru < d.n_red ? ru : r0;   // clamped: the slot is masked below
            int64_t off[MAXF];
            int64_t woff = wbase;
            if (d.nr == 1) {
#pragma unroll
                for (int f = 0; f < MAXF; ++f) off[f] = base[f] + (int64_t)r * d.f[f < d.nf ? f : 0].rs[0];
                if (MODE == ALAN_MODE_WEXPSUM) woff += (int64_t)r * d.w.rs[0];
            } else {
#pragma unroll
                for (int f = 0; f < MAXF; ++f) off[f] = base[f];
                uint32_t rr = r;
                for (int k = d.nr - 1; k >= 0; --k) {
                    const uint32_t q = fd_div(rr, d.rdiv[k]);
                    const int64_t idx = (int64_t)(rr - q * d.rdiv[k].d);
                    rr = q;
#pragma unroll
                    for (int f = 0; f < MAXF; ++f)
                        if (f < d.nf) off[f] += idx * d.f[f].rs[k];
                    if (MODE == ALAN_MODE_WEXPSUM) woff += idx * d.w.rs[k];
                }
            }
#pragma unroll
            for (int f = 0; f < MAXF; ++f)
                if (f < (MODE == ALAN_MODE_BERNOULLI ? 2 : d.nf))
                    val[u][f] = load_as<T>(d.f[f].p, d.f[f].dtype, off[f]);
            if (MODE == ALAN_MODE_WEXPSUM) wv[u] = load_as<T>(d.w.p, d.w.dtype, woff);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float sc[MAXF];
#pragma unroll
            for (int f = 0; f < MAXF; ++f) sc[f] = d.f[f].scale;
            accumulate<T, MODE>(m, s, val[u], wv[u], sc, d.nf, r0 + (uint32_t)u * G < d.n_red);
        }
    }
    };
    // one element per lane (elementwise-like launches, e.g. the backward wrt the big factor): no unrolling,
    // otherwise every lane would issue 3 masked duplicate loads
    if (d.n_red <= G)
        stream(std::integral_constant<int, 1>{});
    else
        stream(std::integral_constant<int, 4>{});

    combine_lanes<T, MODE, BLOCK>(m, s, G);
    if (active && gl == 0) {
        T v = (MODE == ALAN_MODE_LSE) ? lse_finish(m, s) : s;
        if (MODE == ALAN_MODE_NORMAL || MODE == ALAN_MODE_NORMAL_LOGSCALE || MODE == ALAN_MODE_BERNOULLI ||
            MODE == ALAN_MODE_PRODUCER_GRAD)
            v *= (T)d.out_scale;
        v += (T)d.add_const;
        store_as<T>(d.out, d.out_dtype, obase, v);
    }
}


template <int MODE, bool BLOCK>
__global__ __launch_bounds__(256) void reduce_small_kernel(const SmallDesc d, const int logG) {
    small_body<MODE, BLOCK>(d, logG, blockIdx.x);
}

// ONE output from a long reduction -- the last launch of an evaluation: LSE over the parents' K x K grid of the plate
// step's partial sums (34 slices of 900 values at K = 30, 5 of 10^4 at K = 100) plus the parents' own factors
// (reduce_Ks.py:249-251 at the top level, Sample.py:69-86).  In the 256-thread kernel that is a chain of load rounds (a
// thread has 4 elements x 8 slices in flight, a fifth of what it needs: 5.4 us at K = 30); here 1024 threads hold UNR x PF
// = 40 loads each, all of a thread's share in two rounds, and the 16 waves' (max, sum) pairs meet through LDS.
template <int MODE, int UNR, int PF>
__global__ __launch_bounds__(1024) void reduce_wide_kernel(const SmallDesc d) {
#ifdef ALAN_TIMELINE
    const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0)
        for (int i = 0; i < SM_TL_SLOTS; ++i) sm_tl[i] = 0;
    SM_STAMP(0);
#endif
    small_body<MODE, true, 16, UNR, PF>(d, 10, 0);
#ifdef ALAN_TIMELINE
    SM_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SM_STAMP(5);
    if (threadIdx.x == 0) {                               // (the last row of the buffer: tools/small_timeline.py)
        sm_tl[6] = real0, sm_tl[7] = __builtin_amdgcn_s_memrealtime();
        sm_tl[9] = (100ull << 32), sm_tl[10] = 1;
        for (int i = 0; i < SM_TL_SLOTS; ++i) sm_timeline[(SM_TL_WGS - 1) * SM_TL_SLOTS + i] = sm_tl[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Small log-sum-exp + plate sum in one launch (SmallPlateDesc): a lane group per output element; for each plate element
// in turn the group reduces the REDUCE dims (lanes along them, shuffles), lane 0 adds the value to the plate sum.
__global__ __launch_bounds__(256) void reduce_small_plate_kernel(const SmallPlateDesc d, const int logG) {
    const uint32_t G = 1u << logG;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t grp = gid >> logG, gl = gid & (G - 1u);
    const bool active = grp < d.n_out;
    uint32_t o = active ? grp : d.n_out - 1u;
    int32_t base[MAXF], obase = 0, lbase = 0;
#pragma unroll
    for (int f = 0; f < MAXF; ++f) base[f] = 0;
#pragma unroll
    for (int k = SP_NK - 1; k >= 0; --k) {
        const uint32_t q = fd_div(o, d.kdiv[k]);
        const int32_t idx = (int32_t)(o - q * d.kdiv[k].d);
        o = q;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) base[f] += idx * d.fks[f][k];
        obase += idx * d.oks[k];
        lbase += idx * d.lks[k];
    }
    float sc[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) sc[f] = d.fscale[f];
    float total = 0.f;
    constexpr int UNR = 4;
    // the offsets of plate element p: every factor's, and the lse output's
    auto plate_offsets = [&](uint32_t p, int32_t (&pb)[MAXF], int32_t &lp) {
        lp = lbase;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) pb[f] = base[f];
        uint32_t pp = p;
#pragma unroll
        for (int k = SP_NP - 1; k >= 0; --k) {
            const uint32_t q = fd_div(pp, d.pdiv[k]);
            const int32_t idx = (int32_t)(pp - q * d.pdiv[k].d);
            pp = q;
#pragma unroll
            for (int f = 0; f < MAXF; ++f) pb[f] += idx * d.fps[f][k];
            lp += idx * d.lps[k];
        }
    };
    // one round of a lane's loads: UNR elements of the reduced dims from r0 on
    auto load_round = [&](const int32_t (&pb)[MAXF], uint32_t r0, float (&val)[UNR][MAXF]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t ru = r0 + (uint32_t)u * G;
            uint32_t rr = ru < d.n_red ? ru : r0;         // clamped: the slot is masked in accumulate()
            int32_t off[MAXF];
#pragma unroll
            for (int f = 0; f < MAXF; ++f) off[f] = pb[f];
#pragma unroll
            for (int k = SP_NR - 1; k >= 0; --k) {
                const uint32_t q = fd_div(rr, d.rdiv[k]);
                const int32_t idx = (int32_t)(rr - q * d.rdiv[k].d);
                rr = q;
#pragma unroll
                for (int f = 0; f < MAXF; ++f) off[f] += idx * d.frs[f][k];
            }
#pragma unroll
            for (int f = 0; f < MAXF; ++f) val[u][f] = d.f[f][off[f]];        // (unused slots alias factor 0, stride 0)
        }
    };
    if (d.n_red <= UNR * G) {
        // (the usual case: a plate element is ONE round of loads per lane -- the next element's are requested before this
        // one's lanes are combined: the plate walk was a chain of load latencies, one per element)
        float cur[UNR][MAXF], nxt[UNR][MAXF];
        int32_t pb[MAXF], lp, lp_n = 0;
        plate_offsets(0, pb, lp);
        load_round(pb, gl < d.n_red ? gl : 0u, cur);
        for (uint32_t p = 0; p < d.n_plate; ++p) {
            if (p + 1 < d.n_plate) {
                plate_offsets(p + 1, pb, lp_n);
                load_round(pb, gl < d.n_red ? gl : 0u, nxt);
            }
            float m = Num<float>::ninf(), s = 0.f;
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                accumulate<float, ALAN_MODE_LSE>(m, s, cur[u], 0.f, sc, d.nf, gl + (uint32_t)u * G < d.n_red);
            combine_lanes<float, ALAN_MODE_LSE, false>(m, s, G);
            const float v = lse_finish(m, s);
            if (active && gl == 0 && d.lse) d.lse[lp] = v;
            total += v;
            lp = lp_n;
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int f = 0; f < MAXF; ++f) cur[u][f] = nxt[u][f];
        }
    } else {
        for (uint32_t p = 0; p < d.n_plate; ++p) {
            int32_t pb[MAXF], lp;
            plate_offsets(p, pb, lp);
            float m = Num<float>::ninf(), s = 0.f;
            for (uint32_t r0 = gl; r0 < d.n_red; r0 += UNR * G) {
                float val[UNR][MAXF];
                load_round(pb, r0, val);
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    accumulate<float, ALAN_MODE_LSE>(m, s, val[u], 0.f, sc, d.nf, r0 + (uint32_t)u * G < d.n_red);
            }
            combine_lanes<float, ALAN_MODE_LSE, false>(m, s, G);
            const float v = lse_finish(m, s);
            if (active && gl == 0 && d.lse) d.lse[lp] = v;
            total += v;
        }
    }
    if (active && gl == 0) d.out[obase] = total + d.add_const;
}

int launch_small_plate(const SmallPlateDesc &sd, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev) {
    if (gl.grid == 0) return ALAN_OK;
    ALAN_LAUNCH_EXT(reduce_small_plate_kernel, dim3(gl.grid), dim3(256), 0, stream, ev.start, ev.stop, 0, sd, gl.logG);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}


template <bool BLOCK>
__global__ __launch_bounds__(256) void bernoulli_linear_kernel(const LinDesc d, const int logG) {
    lin_body<BLOCK>(d, logG, blockIdx.x);
}

// ALAN_MODE_BERNOULLI_LINEAR_GRAD: d out / d a for term 0's first operand,
//   da[keep, e] = out_scale * sum_R G[keep] * (y - sigmoid(l)) * b[..., e],        l as in the forward.
// FOUR lanes per output row, eight events each (round 4; a thread per row was 36 workgroups at movielens -- 9,000 rows x 5 films
// x 18 events -- each thread walking the summed dims one element at a time with 32 dependent-on-nothing loads it waited for
// before the next element's: 10 us).  A lane keeps its eight events of the row of `a`; the summed dims go by in rounds of up to
// four elements (eight at a time: no faster) whose loads -- 8 of b each, the values, the plain terms -- are all requested
// before the first is used; the logit is
// the four lanes' partial dots added by two shuffles.
__global__ __launch_bounds__(256) void bernoulli_linear_grad_kernel(const LinDesc d) {
    constexpr int EW = 8, RU = 4;
    const uint32_t t0 = blockIdx.x * 256u + threadIdx.x, o0 = t0 >> 2;
    const int e0 = (int)(t0 & 3u) * EW;
    const bool active = o0 < d.n_out;
    uint32_t o = active ? o0 : d.n_out - 1u;
    int32_t abase[LIN_T], bbase[LIN_T], vbase = 0, obase = 0, gbase = 0;
#pragma unroll
    for (int tm = 0; tm < LIN_T; ++tm) abase[tm] = bbase[tm] = 0;
#pragma unroll
    for (int k = LIN_NK - 1; k >= 0; --k) {
        const uint32_t q = fd_div(o, d.kdiv[k]);
        const int32_t idx = (int32_t)(o - q * d.kdiv[k].d);
        o = q;
#pragma unroll
        for (int tm = 0; tm < LIN_T; ++tm) {
            abase[tm] += idx * d.aks[tm][k];
            bbase[tm] += idx * d.bks[tm][k];
        }
        vbase += idx * d.vks[k];
        obase += idx * d.oks[k];
        gbase += idx * d.gks[k];
    }
    const int len = d.len[0], as0 = d.ads[0], bs0 = d.bds[0];
    float arow[EW], acc[EW];
#pragma unroll
    for (int e = 0; e < EW; ++e) {
        arow[e] = e0 + e < len ? d.a[0][abase[0] + min(e0 + e, len - 1) * as0] : 0.f;
        acc[e] = 0.f;
    }
    const float gw = d.g[gbase] * d.out_scale;
    for (uint32_t r0 = 0; r0 < d.n_red; r0 += RU) {
        float brow[RU][EW], y[RU], extra[RU];
        int32_t aoff[RU][LIN_T], boff[RU][LIN_T];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (u > 0 && r0 + (uint32_t)u >= d.n_red) {                          // (uniform: nothing to load)
#pragma unroll
                for (int e = 0; e < EW; ++e) brow[u][e] = 0.f;
                y[u] = extra[u] = 0.f;
#pragma unroll
                for (int tm = 0; tm < LIN_T; ++tm) aoff[u][tm] = abase[tm], boff[u][tm] = bbase[tm];
                continue;
            }
            const uint32_t r = r0 + (uint32_t)u;
            int32_t voff = vbase;
#pragma unroll
            for (int tm = 0; tm < LIN_T; ++tm) aoff[u][tm] = abase[tm], boff[u][tm] = bbase[tm];
            uint32_t rr = r;
#pragma unroll
            for (int k = LIN_NR - 1; k >= 0; --k) {
                const uint32_t q = fd_div(rr, d.rdiv[k]);
                const int32_t idx = (int32_t)(rr - q * d.rdiv[k].d);
                rr = q;
#pragma unroll
                for (int tm = 0; tm < LIN_T; ++tm) {
                    aoff[u][tm] += idx * d.ars[tm][k];
                    boff[u][tm] += idx * d.brs[tm][k];
                }
                voff += idx * d.vrs[k];
            }
#pragma unroll
            for (int e = 0; e < EW; ++e) brow[u][e] = d.b[0][boff[u][0] + min(e0 + e, len - 1) * bs0];
            y[u] = d.val[voff];
            extra[u] = 0.f;
#pragma unroll
            for (int tm = 1; tm < LIN_T; ++tm)
                if (tm < d.nt && d.b[tm] == nullptr) extra[u] += d.a[tm][aoff[u][tm]];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (u > 0 && r0 + (uint32_t)u >= d.n_red) break;                   // (uniform)
            float xl = 0.f;
#pragma unroll
            for (int e = 0; e < EW; ++e) xl = fmaf(arow[e], brow[u][e], xl);   // (arow is 0 beyond len)
            xl += __shfl_xor(xl, 1);
            xl += __shfl_xor(xl, 2);
            xl += extra[u];
#pragma unroll
            for (int tm = 1; tm < LIN_T; ++tm) {
                if (tm >= d.nt || d.b[tm] == nullptr) continue;
                const float *pa = d.a[tm] + aoff[u][tm], *pb = d.b[tm] + boff[u][tm];
                xl += lin_dot<16>(pa, pb, d.len[tm], d.ads[tm], d.bds[tm]);
            }
            // y - sigmoid(x), sigmoid on the fast transcendental instructions (rcp of 1 + 2^(-x log2 e))
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-xl * 1.44269504088896340736f));
            const float c = gw * (y[u] - sg);
#pragma unroll
            for (int e = 0; e < EW; ++e) acc[e] = fmaf(c, brow[u][e], acc[e]);
        }
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < EW; ++e)
            if (e0 + e < len) d.out[obase + (e0 + e) * d.ods] = acc[e];
    }
}

int launch_lin_grad(const LinDesc &ld, hipStream_t stream, const EvPair &ev) {
    if (ld.n_out == 0) return ALAN_OK;
    if (4ull * ld.n_out >= (1ull << 32)) return ALAN_ERR_UNSUPPORTED;             // (four lanes per row on a 32-bit thread index)
    ALAN_LAUNCH_EXT(bernoulli_linear_grad_kernel, dim3((uint32_t)((4ull * ld.n_out + 255) / 256)), dim3(256), 0, stream, ev.start,
                    ev.stop, 0, ld);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int launch_lin(const LinDesc &ld, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev) {
    if (gl.grid == 0) return ALAN_OK;
    if (gl.block)
        ALAN_LAUNCH_EXT((bernoulli_linear_kernel<true>), dim3(gl.grid), dim3(256), 0, stream, ev.start, ev.stop, 0, ld, 8);
    else
        ALAN_LAUNCH_EXT((bernoulli_linear_kernel<false>), dim3(gl.grid), dim3(256), 0, stream, ev.start, ev.stop, 0,
                              ld, gl.logG);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}


__global__ __launch_bounds__(256) void reduce_small_multi_kernel(const SmallMulti m) {
    small_multi_block<>(0, blockIdx.x);
}

#ifdef ALAN_TIMELINE
}  // namespace alan
extern "C" int alan_small_timeline_read(unsigned long long *host_out, int n_wgs) {
    if (n_wgs > alan::SM_TL_WGS) n_wgs = alan::SM_TL_WGS;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(alan::sm_timeline), sizeof(unsigned long long) * n_wgs * alan::SM_TL_SLOTS) ==
                   hipSuccess
               ? alan::SM_TL_SLOTS : -1;
}
namespace alan {
#endif

int launch_small_multi(const SmallDesc *sd, const GroupLaunch *gl, const int *mode, int n, hipStream_t stream,
                       const LinDesc *lin, const alan_noise_t *noise, bool advance) {
    if (n < 1 || n > SMALL_MULTI) return ALAN_ERR_BAD_DESC;
    SmallMulti m;
    const uint32_t blocks = fill_small_multi(m, sd, gl, mode, n, lin);
    if (blocks == 0) return ALAN_OK;
    m.noise = noise_launch(noise, advance);
    ALAN_LAUNCH(reduce_small_multi_kernel, dim3(blocks), dim3(256), 0, stream, m);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

template <int MODE>
static void launch_small_T(const SmallDesc &sd, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev) {
    if (gl.block)
        ALAN_LAUNCH_EXT((reduce_small_kernel<MODE, true>), dim3(gl.grid), dim3(256), 0, stream, ev.start, ev.stop,
                              0, sd, 8);
    else
        ALAN_LAUNCH_EXT((reduce_small_kernel<MODE, false>), dim3(gl.grid), dim3(256), 0, stream, ev.start,
                              ev.stop, 0, sd, gl.logG);
}

int build_small(const Canon &c, const GroupDesc &gd, int mode, int compute_dtype, SmallDesc &sd) {
    if (compute_dtype != ALAN_F32 || gd.out_dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    if (c.nk > SMALL_NK || c.nr > SMALL_NR || c.nf < 1) return ALAN_ERR_UNSUPPORTED;
    if (c.n_out * c.n_red > (1ll << 22)) return ALAN_ERR_UNSUPPORTED;   // big problems are not launch-bound
    if (mode == ALAN_MODE_WEXPSUM && (!c.w.p || c.w.dtype != ALAN_F32)) return ALAN_ERR_UNSUPPORTED;
    const int64_t lim = (1ll << 31) - 1;
    // the kernel accumulates element offsets in int32: besides each stride, the LARGEST reachable offset of every
    // tensor -- sum of (size - 1) * |stride| -- must fit (a small strided view into a tensor of > 2^31 elements, e.g. a
    // slice of a K=100 factor, does not: it goes to the int64 group kernel instead)
    auto reach_ok = [&](const KTensor &x, bool reduce_dims) {
        int64_t reach = 0;
        for (int j = 0; j < c.nk; ++j) {
            const int64_t st = x.ks[j] < 0 ? -x.ks[j] : x.ks[j];
            if (st > lim) return false;
            reach += (c.ksize[j] - 1) * st;
            if (reach > lim) return false;
        }
        for (int j = 0; reduce_dims && j < c.nr; ++j) {
            const int64_t st = x.rs[j] < 0 ? -x.rs[j] : x.rs[j];
            if (st > lim) return false;
            reach += (c.rsize[j] - 1) * st;
            if (reach > lim) return false;
        }
        return true;
    };
    for (int f = 0; f < c.nf; ++f)
        if (!reach_ok(c.f[f], true)) return ALAN_ERR_UNSUPPORTED;
    if (mode == ALAN_MODE_WEXPSUM && !reach_ok(c.w, true)) return ALAN_ERR_UNSUPPORTED;
    if (!reach_ok(c.o, false)) return ALAN_ERR_UNSUPPORTED;
    std::memset(&sd, 0, sizeof(sd));
    for (int k = 0; k < SMALL_NK; ++k) sd.kdiv[k] = make_fastdiv(1);
    for (int k = 0; k < SMALL_NR; ++k) sd.rdiv[k] = make_fastdiv(1);
    // right-align the problem's dims in the fixed-size arrays (leading slots: size 1, stride 0)
    const int ko = SMALL_NK - c.nk, ro = SMALL_NR - c.nr;
    for (int j = 0; j < c.nk; ++j) sd.kdiv[ko + j] = make_fastdiv((uint32_t)c.ksize[j]);
    for (int j = 0; j < c.nr; ++j) sd.rdiv[ro + j] = make_fastdiv((uint32_t)c.rsize[j]);
    for (int f = 0; f < MAXF; ++f) {
        const KTensor &src = c.f[f < c.nf ? f : 0];
        if (src.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
        sd.f[f] = (const float *)src.p;
        sd.fscale[f] = f < c.nf ? src.scale : 0.f;
        if (f >= c.nf) continue;                       // alias of factor 0, strides stay 0
        for (int j = 0; j < c.nk; ++j) {
            if (src.ks[j] > lim || src.ks[j] < -lim) return ALAN_ERR_UNSUPPORTED;
            sd.fks[f][ko + j] = (int32_t)src.ks[j];
        }
        for (int j = 0; j < c.nr; ++j) {
            if (src.rs[j] > lim || src.rs[j] < -lim) return ALAN_ERR_UNSUPPORTED;
            sd.frs[f][ro + j] = (int32_t)src.rs[j];
        }
    }
    if (mode == ALAN_MODE_WEXPSUM) {
        sd.w = (const float *)c.w.p;
        for (int j = 0; j < c.nk; ++j) {
            if (c.w.ks[j] > lim || c.w.ks[j] < -lim) return ALAN_ERR_UNSUPPORTED;
            sd.wks[ko + j] = (int32_t)c.w.ks[j];
        }
        for (int j = 0; j < c.nr; ++j) {
            if (c.w.rs[j] > lim || c.w.rs[j] < -lim) return ALAN_ERR_UNSUPPORTED;
            sd.wrs[ro + j] = (int32_t)c.w.rs[j];
        }
    }
    for (int j = 0; j < c.nk; ++j) {
        if (c.o.ks[j] > lim || c.o.ks[j] < -lim) return ALAN_ERR_UNSUPPORTED;
        sd.oks[ko + j] = (int32_t)c.o.ks[j];
    }
    sd.out = (float *)const_cast<void *>(c.o.p);
    sd.n_out = gd.n_out;
    sd.n_red = gd.n_red;
    sd.nf = c.nf;
    sd.out_scale = gd.out_scale;
    sd.add_const = (float)gd.add_const;
    return ALAN_OK;
}

int try_launch_small(const Canon &c, const GroupDesc &gd, const GroupLaunch &gl, int mode, int compute_dtype,
                     hipStream_t stream, const EvPair &ev, int64_t presum_n, int64_t presum_stride, bool dry) {
    SmallDesc sd;
    const int rc = build_small(c, gd, mode, compute_dtype, sd);
    if (rc != ALAN_OK) return rc;
    if (presum_n > 1) {
        if (mode != ALAN_MODE_LSE && mode != ALAN_MODE_SUM) return ALAN_ERR_UNSUPPORTED;
        const int64_t st = presum_stride < 0 ? -presum_stride : presum_stride;
        if (presum_n > (1 << 16) || st * presum_n >= (1ll << 30)) return ALAN_ERR_UNSUPPORTED;      // (int32 offsets)
        sd.presum_n = (int32_t)presum_n, sd.presum_stride = (int32_t)presum_stride;
    }
    if (dry) return (ev.ring_n && (gd.n_out != 1 || gl.grid != 1)) ? ALAN_ERR_UNSUPPORTED : ALAN_OK;
    if (ev.ring_n) {
        if (gd.n_out != 1 || gl.grid != 1) return ALAN_ERR_UNSUPPORTED;
        sd.ring_slots = (float *const *)ev.ring_slots;
        sd.ring_counter = (int32_t *)ev.ring_counter;
        sd.ring_n = ev.ring_n, sd.ring_and_out = ev.ring_and_out;
    }
    if (gd.n_out == 0) return ALAN_OK;
    // one output, thousands of loads: the 1024-thread kernel (every load of a thread's share in flight in two rounds)
    const int64_t slices = sd.presum_n > 1 ? sd.presum_n : 1;
    if (gd.n_out == 1 && gl.block && (mode == ALAN_MODE_LSE || mode == ALAN_MODE_SUM) &&
        (int64_t)sd.n_red * (slices + sd.nf - 1) >= 4096) {
        const int per_thread = (int)((sd.n_red + 1023) / 1024);
        auto wide = [&](auto kern) { ALAN_LAUNCH_EXT(kern, dim3(1), dim3(1024), 0, stream, ev.start, ev.stop, 0, sd); };
        const bool lse = mode == ALAN_MODE_LSE;
        if (slices >= 16 || per_thread <= 1)
            lse ? wide(reduce_wide_kernel<ALAN_MODE_LSE, 1, 40>) : wide(reduce_wide_kernel<ALAN_MODE_SUM, 1, 40>);
        else if (slices >= 6 || per_thread <= 4)
            lse ? wide(reduce_wide_kernel<ALAN_MODE_LSE, 4, 10>) : wide(reduce_wide_kernel<ALAN_MODE_SUM, 4, 10>);
        else
            lse ? wide(reduce_wide_kernel<ALAN_MODE_LSE, 10, 4>) : wide(reduce_wide_kernel<ALAN_MODE_SUM, 10, 4>);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    }
    switch (mode) {
        case ALAN_MODE_LSE: launch_small_T<ALAN_MODE_LSE>(sd, gl, stream, ev); break;
        case ALAN_MODE_SUM: launch_small_T<ALAN_MODE_SUM>(sd, gl, stream, ev); break;
        case ALAN_MODE_WEXPSUM: launch_small_T<ALAN_MODE_WEXPSUM>(sd, gl, stream, ev); break;
        case ALAN_MODE_DOT: launch_small_T<ALAN_MODE_DOT>(sd, gl, stream, ev); break;
        case ALAN_MODE_AFFINE: launch_small_T<ALAN_MODE_AFFINE>(sd, gl, stream, ev); break;
        case ALAN_MODE_NORMAL: launch_small_T<ALAN_MODE_NORMAL>(sd, gl, stream, ev); break;
        case ALAN_MODE_NORMAL_LOGSCALE: launch_small_T<ALAN_MODE_NORMAL_LOGSCALE>(sd, gl, stream, ev); break;
        case ALAN_MODE_BERNOULLI: launch_small_T<ALAN_MODE_BERNOULLI>(sd, gl, stream, ev); break;
        case ALAN_MODE_PRODUCER_GRAD: launch_small_T<ALAN_MODE_PRODUCER_GRAD>(sd, gl, stream, ev); break;
        default: return ALAN_ERR_BAD_DESC;
    }
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

template <typename T, int MODE>
static int launch_group_T(const GroupDesc &gd, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev) {
    if (gl.block) {
        ALAN_LAUNCH_EXT((reduce_group_kernel<T, MODE, true>), dim3(gl.grid), dim3(256), 0, stream, ev.start,
                              ev.stop, 0, gd, 8);
    } else {
        ALAN_LAUNCH_EXT((reduce_group_kernel<T, MODE, false>), dim3(gl.grid), dim3(256), 0, stream, ev.start,
                              ev.stop, 0, gd, gl.logG);
    }
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int launch_group(const GroupDesc &gd, const GroupLaunch &gl, int mode, int compute_dtype, hipStream_t stream,
                 const EvPair &ev) {
    if (gd.n_out == 0) return ALAN_OK;
#define ALAN_DISPATCH(T)                                                              \
    switch (mode) {                                                                   \
        case ALAN_MODE_LSE: return launch_group_T<T, ALAN_MODE_LSE>(gd, gl, stream, ev);  \
        case ALAN_MODE_SUM: return launch_group_T<T, ALAN_MODE_SUM>(gd, gl, stream, ev);  \
        case ALAN_MODE_WEXPSUM: return launch_group_T<T, ALAN_MODE_WEXPSUM>(gd, gl, stream, ev); \
        case ALAN_MODE_DOT: return launch_group_T<T, ALAN_MODE_DOT>(gd, gl, stream, ev); \
        case ALAN_MODE_AFFINE: return launch_group_T<T, ALAN_MODE_AFFINE>(gd, gl, stream, ev); \
        case ALAN_MODE_NORMAL: return launch_group_T<T, ALAN_MODE_NORMAL>(gd, gl, stream, ev); \
        case ALAN_MODE_BERNOULLI: return launch_group_T<T, ALAN_MODE_BERNOULLI>(gd, gl, stream, ev); \
        case ALAN_MODE_NORMAL_LOGSCALE: return launch_group_T<T, ALAN_MODE_NORMAL_LOGSCALE>(gd, gl, stream, ev); \
        case ALAN_MODE_PRODUCER_GRAD: return launch_group_T<T, ALAN_MODE_PRODUCER_GRAD>(gd, gl, stream, ev); \
    }
    if (compute_dtype == ALAN_F32) {
        ALAN_DISPATCH(float)
    } else {
        ALAN_DISPATCH(double)
    }
#undef ALAN_DISPATCH
    return ALAN_ERR_BAD_DESC;
}

}  // namespace alan
