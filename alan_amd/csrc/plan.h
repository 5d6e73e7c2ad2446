// Host-side planning: canonicalise an alan_reduce_desc_t (drop unit dims, order dims for
// coalescing, merge contiguous dims) and pick a kernel + launch geometry.
#pragma once
#include <cstring>
#include <utility>

#include "common.h"

namespace alan {

struct KTensor {
    const void *p;
    int32_t dtype;
    float scale;
    int64_t ks[MAXD];  // element stride per canonical keep dim
    int64_t rs[MAXD];  // element stride per canonical reduce dim
};

// Kernel argument of the group kernel (passed by value; lives in the kernarg segment).
struct GroupDesc {
    int32_t nf, nk, nr, out_dtype;
    uint32_t n_out, n_red;
    FastDiv kdiv[MAXD], rdiv[MAXD];  // sizes, outermost first; index decomposition walks backwards
    KTensor f[MAXF];
    KTensor w;
    void *out;
    int64_t oks[MAXD];
    double add_const;
    float out_scale;   // producer modes: out = out_scale * sum + add_const
};

// Compact argument of the small-problem kernel: at most 3 keep and 2 reduce dims (after merging), fp32
// everywhere, 32-bit strides.  Everything a thread needs arrives in one round of scalar loads -- the generic
// kernel walks its 1.4 KB descriptor with run-time loops, i.e. a chain of ~10 dependent scalar loads, which
// is most of the duration of a launch that reads a few hundred elements.
constexpr int SMALL_NK = 4, SMALL_NR = 2;
constexpr int SMALL_MULTI = 8;     // problems per reduce_small_multi_kernel launch (its kernel argument: 3.3 KB of the 4 KB)

struct SmallDesc {
    const float *f[MAXF];
    const float *w;
    float *out;
    uint32_t n_out, n_red;
    int32_t nf;
    float out_scale, add_const;
    FastDiv kdiv[SMALL_NK], rdiv[SMALL_NR];
    int32_t fks[MAXF][SMALL_NK], frs[MAXF][SMALL_NR];
    int32_t wks[SMALL_NK], wrs[SMALL_NR], oks[SMALL_NK];
    float fscale[MAXF];
    float *const *ring_slots;      // result ring (alan_reduce_desc_t.ring_*); ring_n = 0: off
    int32_t *ring_counter;
    int32_t ring_n, ring_and_out;
    int32_t presum_n, presum_stride;   // role ALAN_PRESUM: factor 0 is the sum of presum_n slices presum_stride apart (0: off)
    int32_t noise_on;                  // alan_noise_t: factor 1 is generated (ALAN_MODE_AFFINE / ALAN_MODE_DOT)
    uint64_t noise_off;
};

// What the problems of ONE launch share of alan_noise_t (a kernel argument of its own: eight SmallDescs fill the 4 KB)
struct NoiseLaunch {
    uint64_t seed;
    const unsigned long long *cell;
    unsigned long long *receipt, *advance;
    uint32_t advance_by, pad_;
};

// Kernel argument of the linear-logits Bernoulli producer (ALAN_MODE_BERNOULLI_LINEAR): fp32, 32-bit offsets, at
// most LIN_NK keep dims, LIN_NR summed dims and LIN_T terms; a term is a[...] or sum_e a[... + e ads] * b[... + e bds].
constexpr int LIN_T = 3, LIN_NK = 4, LIN_NR = 2;

struct LinDesc {
    const float *val;
    float *out;
    const float *a[LIN_T], *b[LIN_T];           // b = nullptr: a plain term
    uint32_t n_out, n_red;
    int32_t nt;
    float out_scale, add_const;
    FastDiv kdiv[LIN_NK], rdiv[LIN_NR];         // right-aligned; unused leading slots: size 1
    int32_t vks[LIN_NK], vrs[LIN_NR], oks[LIN_NK];
    int32_t aks[LIN_T][LIN_NK], ars[LIN_T][LIN_NR], bks[LIN_T][LIN_NK], brs[LIN_T][LIN_NR];
    int32_t len[LIN_T], ads[LIN_T], bds[LIN_T];
    // ALAN_MODE_BERNOULLI_LINEAR_GRAD: upstream gradient (strides over the keep dims), the gradient's dot stride
    const float *g;
    int32_t gks[LIN_NK], ods;
};

// Several INDEPENDENT small problems in one launch (alan_reduce_batch): the per-variable log-prob producers of a
// plate are a handful of launch-latency-bound kernels (4-5 us each inside a replayed graph) that do not depend on
// each other.  Workgroups are dealt to the problems in order; mode and lane-group shape are run-time here.
template <int NP>
struct SmallMultiT {
    int32_t n;
    uint32_t first_block[NP];    // of problem k; 0xffffffff beyond the last: a workgroup finds its problem from ONE fetch of these
                                 // (a loop over them was a dependent scalar load per problem in front of everything else)
    struct Head {
        int32_t mode, logG, block;
        uint32_t first_block;
    } head[NP];                  // fetched with the problem's descriptor
    SmallDesc d[NP];
    LinDesc lin;                 // of the (at most one) ALAN_MODE_BERNOULLI_LINEAR problem
    NoiseLaunch noise;
};
typedef SmallMultiT<SMALL_MULTI> SmallMulti;

// Kernel argument of the small log-sum-exp + plate-sum kernel: out[keep] = sum_plate LSE_red(sum_f factor_f) + add_const
// in ONE launch (reduce_Ks.py:249-251 then logpq.py:149) for problems too small for the rows kernel -- a lane group per
// output walks the plate elements one after the other.  fp32, 32-bit offsets, <= SP_NK keep, SP_NP plate, SP_NR reduce dims.
constexpr int SP_NK = 3, SP_NP = 2, SP_NR = 2;
struct GroupLaunch;
struct EvPair;

struct SmallPlateDesc {
    const float *f[MAXF];
    float *out, *lse;                           // lse: optional per-(keep, plate) log-sum-exp values (the backward's input)
    uint32_t n_out, n_plate, n_red;
    int32_t nf;
    float add_const;
    FastDiv kdiv[SP_NK], pdiv[SP_NP], rdiv[SP_NR];     // right-aligned; unused leading slots: size 1
    int32_t fks[MAXF][SP_NK], fps[MAXF][SP_NP], frs[MAXF][SP_NR];
    int32_t oks[SP_NK], lks[SP_NK], lps[SP_NP];
    float fscale[MAXF];
};
int launch_small_plate(const SmallPlateDesc &sd, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev);

struct GroupLaunch {
    int logG;
    bool block;
    uint32_t grid;
};

// Canonical problem: keep dims and reduce dims, each ordered outermost -> innermost.
struct Canon {
    int nk = 0, nr = 0;
    int64_t ksize[MAXD], rsize[MAXD];
    int nf = 0;
    KTensor f[MAXF];
    KTensor w;       // w.p == nullptr when absent
    KTensor o;       // output: ks only
    KTensor l;       // optional per-(KEEP,PLATE) log-sum-exp output (l.p == nullptr when absent): ks only
    bool kplate[MAXD];     // canonical keep dim j is a PLATE dim (summed after the log-sum-exp)
    int dominant = 0;      // index of the largest factor
    int64_t n_out = 1, n_red = 1;
    bool red_contig = false;   // dominant factor's unit-stride dim is the innermost reduce dim
    bool keep_contig = false;  // ... is the innermost keep dim
};

// keep_mask / red_mask: bit d set => dim d of the descriptor is a keep / reduce dim.  Dims in
// neither mask must have size 1 for every tensor that is read (they are ignored).
int canonicalise(const alan_reduce_desc_t &d, uint32_t keep_mask, uint32_t red_mask,
                 const alan_tensor_t &out, Canon &c, uint32_t plate_mask = 0,
                 const alan_tensor_t *lse_out = nullptr);

// Optional hipEvent pair (alan_reduce_desc_t.ev_*): handed to hipExtLaunchKernelGGL for the dominant kernel
// of a call, so the events carry that kernel's own start / stop timestamps (what rocprofv3 reports).
struct EvPair {
    hipEvent_t start = nullptr, stop = nullptr;
    // result ring of the call (alan_reduce_desc_t.ring_*): only the single-workgroup small kernel honours it
    void *ring_slots = nullptr, *ring_counter = nullptr;
    int ring_n = 0, ring_and_out = 0;
};
inline NoiseLaunch noise_launch(const alan_noise_t *n, bool advance) {
    NoiseLaunch r;
    std::memset(&r, 0, sizeof(r));
    if (n && n->on) {
        r.seed = n->seed, r.cell = (const unsigned long long *)n->cell, r.receipt = (unsigned long long *)n->receipt;
        r.advance = advance ? (unsigned long long *)n->advance : nullptr;
        r.advance_by = n->advance_by;
    }
    return r;
}
int plan_group(const Canon &c, int out_dtype, double add_const, GroupDesc &gd, GroupLaunch &gl,
               float out_scale = 1.f);

int launch_group(const GroupDesc &gd, const GroupLaunch &gl, int mode, int compute_dtype, hipStream_t stream,
                 const EvPair &ev = EvPair());
// Small-problem variant of the group kernel; ALAN_ERR_UNSUPPORTED when the problem does not fit SmallDesc.
// (presum_n > 1: factor 0 is a sum of slices, role ALAN_PRESUM; dry: build and check only, launch nothing)
int try_launch_small(const Canon &c, const GroupDesc &gd, const GroupLaunch &gl, int mode, int compute_dtype,
                     hipStream_t stream, const EvPair &ev, int64_t presum_n = 0, int64_t presum_stride = 0, bool dry = false);
// The two halves of it, for alan_reduce_batch: fill a SmallDesc; launch up to SMALL_MULTI of them as one kernel.
int build_small(const Canon &c, const GroupDesc &gd, int mode, int compute_dtype, SmallDesc &sd);
// `lin`: the LinDesc of the (at most one) problem whose mode[] entry is ALAN_MODE_BERNOULLI_LINEAR; its sd[] slot is unused
int launch_small_multi(const SmallDesc *sd, const GroupLaunch *gl, const int *mode, int n, hipStream_t stream,
                       const LinDesc *lin = nullptr, const alan_noise_t *noise = nullptr, bool advance = false);
// (its kernel argument alone; returns the number of workgroups the problems take together)
template <int NP>
inline uint32_t fill_small_multi(SmallMultiT<NP> &m, const SmallDesc *sd, const GroupLaunch *gl, const int *mode, int n,
                                 const LinDesc *lin) {
    std::memset(&m, 0, sizeof(m));
    m.n = n;
    if (lin) m.lin = *lin;
    // What the launch's own timeline showed (tools/small_timeline.py, movielens K = 30: 1,416 workgroups, 8.95 us): every problem
    // had been given the lanes that suit it ALONE on an idle chip -- 32 lanes per output for a sum of 18 -- so the batch did not
    // fit the chip at once (5-6 workgroups of 256 threads per CU), and the workgroups dealt last, the linear-logits
    // producer's with the longest life of all (5.5 us), started 4-5 us late.  So (1) the biggest problems give up lanes until the
    // launch is about one chipful, and (2) behind the problems of a few workgroups (chains of dependent latencies that should
    // start with the launch) the problems go out by the work of one of their lanes, most first.
    GroupLaunch g2[NP];
    for (int i = 0; i < n; ++i) g2[i] = gl[i];
    auto n_out_of = [&](int i) { return mode[i] == ALAN_MODE_BERNOULLI_LINEAR && lin ? lin->n_out : sd[i].n_out; };
    auto n_red_of = [&](int i) { return mode[i] == ALAN_MODE_BERNOULLI_LINEAR && lin ? lin->n_red : sd[i].n_red; };
    constexpr uint32_t CHIPFUL = 1024;
    // (lanes that cost nothing to give up: a lane of the small kernel has four elements' loads in flight at once, so down to
    // four elements per lane the walk is still one round of loads)
    for (int i = 0; i < n; ++i) {
        if (g2[i].block || mode[i] == ALAN_MODE_BERNOULLI_LINEAR || mode[i] == ALAN_MODE_NORMAL_TABLE || g2[i].grid <= 4) continue;
        const uint64_t nr = n_red_of(i);
        while (g2[i].logG > 0 && ((nr + (1ull << (g2[i].logG - 1)) - 1) >> (g2[i].logG - 1)) <= 4) g2[i].logG -= 1;
        g2[i].grid = (uint32_t)((((uint64_t)n_out_of(i) << g2[i].logG) + 255) / 256);
    }
    for (;;) {
        uint32_t total = 0;
        int big = -1;
        for (int i = 0; i < n; ++i) {
            total += g2[i].grid;
            if (!g2[i].block && g2[i].logG > 2 && g2[i].grid > 64 && (big < 0 || g2[i].grid > g2[big].grid)) big = i;
        }
        if (total <= CHIPFUL || big < 0) break;
        g2[big].logG -= 1;
        g2[big].grid = (uint32_t)((((uint64_t)n_out_of(big) << g2[big].logG) + 255) / 256);
    }
    int order[NP];
    for (int i = 0; i < n; ++i) order[i] = i;
    auto weight = [&](int i) -> int64_t {
        if (g2[i].grid <= 4) return (int64_t)1 << 40;
        const int64_t per_lane = g2[i].block ? (int64_t)n_red_of(i) / 256 + 1 : ((int64_t)n_red_of(i) >> g2[i].logG) + 1;
        return per_lane * (mode[i] == ALAN_MODE_BERNOULLI_LINEAR ? 8 : 1);       // (an element of it: a dot product)
    };
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && weight(order[j]) > weight(order[j - 1]); --j) std::swap(order[j], order[j - 1]);
    uint32_t blocks = 0;
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        m.head[k].mode = mode[i];
        m.head[k].logG = g2[i].logG;
        m.head[k].block = g2[i].block ? 1 : 0;
        m.head[k].first_block = m.first_block[k] = blocks;
        blocks += g2[i].grid;
        m.d[k] = sd[i];
    }
    for (int k = n; k < NP; ++k) m.first_block[k] = 0xffffffffu;
    return blocks;
}
int launch_lin(const LinDesc &ld, const GroupLaunch &gl, hipStream_t stream, const EvPair &ev);
int launch_lin_grad(const LinDesc &ld, hipStream_t stream, const EvPair &ev);

// rows.hip: LDS-staged fast path.  Returns ALAN_ERR_UNSUPPORTED when the canonical problem does not
// fit it (caller then falls back to the group kernel).  With PLATE dims in the canonical problem the
// plate sum is fused (per-workgroup partial sums over a chunk of the plate + a tiny second stage).
struct RowsPlan {
    bool ok = false;
    int L = 0, RB = 0, threads = 0, logG = 0, gs_off = 0;
    uint32_t NO = 0, P = 1, n_windows = 0, n_chunks = 1, p_chunk = 1;
    size_t lds_bytes = 0, partial_bytes = 0;
    bool rot = false, vec2 = false;
};
RowsPlan plan_rows(const Canon &c, int mode, int compute_dtype);
int launch_rows(const Canon &c, const RowsPlan &rp, int mode, double add_const, void *workspace,
                size_t workspace_bytes, hipStream_t stream, const EvPair &ev = EvPair());

// pair.hip: a log-sum-exp whose output is bigger than every factor (two factors that meet only in the reduce dim), tiles
// of the output with the factors' rows staged in LDS; optional plate sum inside the workgroup.
constexpr int PAIR_NB = 3, PAIR_NP = 2, PAIR_T = 8, PAIR_RMAX = 256;

struct PairDesc {
    // the factors by side -- row side: carries the tile dim i (or neither tile dim), col side: carries j -- in MAXF / 2 slots
    // each; a slot nobody uses repeats the side's first factor with weight 0 (no branch around a load)
    const float *f[2][MAXF / 2];
    float w[2][MAXF / 2];
    int32_t ns[2];                            // slots in use per side
    int32_t sa[2][MAXF / 2], sx[2][MAXF / 2];             // strides over a and over the side's tile dim
    int32_t sb[2][MAXF / 2][PAIR_NB], sp[2][MAXF / 2][PAIR_NP];
    float *out;
    int32_t R, NI, NJ, n_plate, ntj;
    float add_const;
    FastDiv bdiv[PAIR_NB], pdiv[PAIR_NP];     // batch (kept) dims and plate dims, right-aligned; unused: size 1
    FastDiv rdiv;                             // by R
    int32_t a_fast[2];                        // staging: lanes along a (else along the tile dim)
    int32_t split;                            // a plate element per workgroup (gridDim.z), partial results in ws
    float *ws;
    int32_t osb[PAIR_NB], osi, osj;
    int32_t bern;                             // the Bernoulli-of-linear-logits variant: observations y[b, a]
    const float *y;
    int32_t y_sa, y_sb[PAIR_NB];
    float out_scale;
};


bool pair_prepare(const alan_reduce_desc_t &d, uint32_t keep, uint32_t red, uint32_t plate, PairDesc &pd, dim3 &grid,
                  size_t &lds_bytes);
size_t pair_workspace_bytes(const PairDesc &pd, dim3 grid);
int launch_pair(const PairDesc &pd, dim3 grid, size_t lds_bytes, void *workspace, size_t workspace_bytes,
                hipStream_t stream, const EvPair &ev);

// normal.hip: register-blocked Normal producer (value / loc / scale on disjoint dims).
// (dry: only answer whether it would take the problem)
int try_launch_normal_outer(const Canon &c, bool log_scale, float out_scale, double add_const, hipStream_t stream,
                            const EvPair &ev, bool dry = false);

}  // namespace alan
