// The fused plate step on the bf16 matrix instructions (the default kernel of alan_normal_lse; launched by normal_lse.hip).
// gfx950 only.
#pragma once
#include "common.h"
#include "normal_lse_table.h"

namespace alan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// The same tile on the bf16 matrix instructions with 3-way split operands (the default since round 3).
//
// fp32 MFMA runs at the vector rate and holds the vector unit while it does (tools/mfma_f32_probe.hip: the log-sum-exp's
// VALU work ADDS to the matrix time, 442 ns per tile per SIMD).  v_mfma_f32_32x32x16_bf16 is 16 x faster per flop and
// leaves the vector unit free for 24 of its 32 cycles.  Each f32 operand is split exactly into three bf16 pieces
// (x = h + m + l, round-to-nearest at every step) and the product a b is taken as the six terms of order >= 2^-16,
//     ah bh + ah bm + am bh + ah bl + al bh + am bm        (every bf16 x bf16 product is exact in f32; f32 accumulate),
// the terms dropped being <= 2^-23 |a b|: measured against fp64 the result is as close as the f32 fma chain
// (tools/mfma_bf16x3_probe.hip: 2.35e-7 against 2.51e-7 of sum |a b|).  The six terms are laid out ALONG the contraction
// dim -- 6 (E + 1) slots, 114 for E = 18 -- so a tile takes 8 matrix instructions (256 cycles) where the f32 form takes
// 10 of 64 cycles: per event a lane holds three packed registers
//     A: (ah, ah) (am, ah) (al, am)        B: (bh, bm) (bh, bl) (bh, bm)        [low half, high half]
// and lane half h takes the events 2 q + h; MFMA step t consumes registers 4 t .. 4 t + 3 of both.  B is split once per
// workgroup into an LDS table (NST tiles x NSTEP steps x 64 lanes x 16 bytes, read with one ds_read_b128 per step; in
// registers when the wave has one scale tile -- and then loaded READY-MADE when the caller had it built ahead of the launch:
// normal_lse_table.h, template argument TBL); A costs 9 VALU per element and tile of 32 rows and is shared by the
// wave's NST scale tiles.  Rows beyond NK and -inf small factors carry NL_BIG (finite: inf - inf would poison the split)
// in the small-factor slot: 2^(min - NL_BIG log2e) = 0, and a column whose rows are ALL masked ends with a minimum
// >= 1e29, which is reported as the NaN utils.py:219 gives.
typedef short bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
constexpr float NL_BIG = 1e30f;

// Diagnostic build (make TIMELINE=1 -> tools/_build/libalan_timeline.so; tools/nlse_timeline.py): every wave stamps
// s_memtime at the phases of its life and stores them, with s_memrealtime of its start and its hardware id, into a
// buffer of the library's own that no other code reads.  In the default build no stamp executes.
#ifdef ALAN_TIMELINE
constexpr int NL_TL_SLOTS = 16, NL_TL_WAVES = 8192;
__device__ unsigned long long nl_timeline[NL_TL_WAVES * NL_TL_SLOTS];
#define NL_STAMP(i) (tl[i] = __builtin_amdgcn_s_memtime())
#else
#define NL_STAMP(i) ((void)0)
#endif

// Its kernel argument: 32-bit sizes and strides only, and what every wave would otherwise derive with integer divisions
// (the round-3 timeline: 744 instructions, half of them scalar address arithmetic on 64-bit strides, before a wave's
// first tile -- 2.3 of the 8 us a wave lives at K = 30).  Filled by the launcher, which takes this kernel only when
// every offset fits 31 bits and no stride is negative.
struct X3Desc {
    const float *val, *loc, *scl;
    const u32x4v *tbl;                // the scale table built ahead of the launch (TBL kernels; normal_lse_table.h)
    float *part, *lse;
    const float *small[4];
    int32_t M, NK, NL, NS, E, n_sub, n_small, log_scale;   // n_sub: the plate in this many slices, four per workgroup
    int32_t v_sm, l_sl, l_se, s_ss, s_se;
    int32_t small_sm[4], small_sk[4];
    int32_t nkt, nlg;                 // k tiles per plate element; groups of NLW loc rows
    int32_t m_q, m_r;                 // M = n_sub m_q + m_r: slice c = m_q (+ 1 where c < m_r) plate elements
    uint32_t rcp_e;                   // ceil(2^16 / E): floor(f / E) = (f * rcp_e) >> 16 for f < 2048
};

// A prepared launch of it (nl_x3_prepare, normal_lse.hip).
struct X3Prep {
    X3Desc x;
    uint32_t gx, gy, gz;              // (scale-tile groups, loc-row groups, groups of four plate slices)
    int eq, nst, nlw, n_chunks;
    bool flat, tbl;
    size_t lds;
};
int nl_x3_prepare(const alan_normal_lse_desc_t &a, void *part, X3Prep &o);

// EQ: events per lane half incl. the small-factor slot (2 EQ >= E + 1).  Grid: x = group of NST scale tiles, y = group
// of NLW loc rows, z = group of four slices of the plate, one per wave: the waves of a workgroup share the B table and
// the loc rows and add up their partial sums through LDS, so a launch leaves gridDim.z partial results per output (34
// at K = 30 where the first build left 150) for the consumer to add.  Slice c: M / n_sub plate elements, one more
// in the first M % n_sub slices.  The value tile is staged as in the f32 kernel (STAGE): contiguous rows only.
template <int EQ, int NST, int NLW, bool FLAT, bool TBL = false>
__device__ __forceinline__ void normal_lse_x3_body(const X3Desc &d, const int bx, const int by, const int bz, const int gx,
                                                   const int gy) {
    static_assert(!FLAT || NLW == 1, "flat row tiling: one loc row per wave");
    static_assert(!TBL || NST == 1, "a ready-made scale table: one scale tile per wave");
    constexpr int NSTEP = (3 * EQ + 3) / 4, NV = 4 * NSTEP;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: what depends on it alone stays on the scalar unit)
    const int j = lane & 31, h = lane >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS, nkt = d.nkt;
    const int ES = E | 1;                                            // row stride of the staged value tile: odd, conflict-free
    const int sg = bx, l = by * NLW, sub = bz * 4 + wave;
    const int slot_h = E > 2 * (EQ - 1) ? 1 : 0;                   // the small-factor slot: event pair EQ - 1, this half
    const float inf = __builtin_huge_valf();
#ifdef ALAN_TIMELINE
    unsigned long long tl[NL_TL_SLOTS] = {};
    const unsigned long long tl_real = __builtin_amdgcn_s_memrealtime();
    NL_STAMP(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(NK), "s"(E), "s"(NS) : "memory");       // (the kernel arguments have arrived)
    NL_STAMP(13);
#endif
    // ---- LDS: B table | partial log-normalisers | per wave: value tile [32][ES], loc rows [NLW][32]
    u32x4v *bt = reinterpret_cast<u32x4v *>(lds);
    float *lgp_l = lds + NST * NSTEP * 64 * 4;
    float *tile = lgp_l + NST * 4 * 64 + wave * (32 * 33 + NLW * 36);
    float *locl = tile + 32 * 33;                                   // [NLW][36]: events 0 .. 2 EQ - 1 <= 33
    // ---- everything the wave needs from memory is requested before anything waits -- the scale rows FIRST: the B table
    // built from them, and the barrier behind it, are every wave's critical path (round 4: they were requested last, behind
    // ~150 instructions of address arithmetic for the value tile).  Thread (wave w, lane) takes the event pairs w, w + 4,
    // ... of its lane's (scale row, half) in every scale tile.
    constexpr int NQI = (EQ + 3) / 4;
    float xs[NST][NQI];
    u32x4v breg[NSTEP];                               // B of the unit about to be multiplied (one scale tile: for good)
    float lgn[NST];
    auto load_table = [&]() {
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) breg[step] = d.tbl[step * 64 + lane];
        lgn[0] = reinterpret_cast<const float *>(d.tbl + NSTEP * 64)[j];
        asm volatile("" ::: "memory");
    };
    if constexpr (!TBL) {
        const int joff = j * d.s_ss, hoff = h ? d.s_se : 0, last_s = (NS - 1) * d.s_ss, last_e = (E - 1) * d.s_se;
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int qi = 0; qi < NQI; ++qi) {
                const int q = wave + 4 * qi;                                                           // (scalar)
                xs[st][qi] = d.scl[(uint32_t)(min(joff + 32 * (sg * NST + st) * d.s_ss, last_s) + min(hoff + q * 2 * d.s_se, last_e))];
            }
        asm volatile("" ::: "memory");                // (keep them in front of the loads below)
    }
    // The loc rows: one coalesced load per row, handed to the lanes through LDS (lane (j, h) wants events 2 q + h: ten
    // reads at immediate offsets)
    float lrow[NLW];
#pragma unroll
    for (int lw = 0; lw < NLW; ++lw)
        lrow[lw] = d.loc[(uint32_t)(min(l + lw, d.NL - 1) * d.l_sl + min(lane, E - 1) * d.l_se)];
    // (the longer slices first: a workgroup's four waves then have equally many, bar one workgroup, and the workgroups with the
    // most work are dispatched first.  c M / n_sub as the boundary cost two 64-bit divisions -- ~200 dependent scalar
    // instructions, 0.4 us -- in front of the first value-tile load)
    const int m0 = sub * d.m_q + min(sub, d.m_r), m1 = m0 + d.m_q + (sub < d.m_r ? 1 : 0);
    const int rows_total = (m1 - m0) * NK;
    const int n_tiles = FLAT ? (rows_total + 31) >> 5 : (m1 - m0) * nkt;
    // the value tile: 32 rows of E floats are one contiguous run; lane i takes floats i, i + 64, ... of it (buffer loads:
    // a scalar descriptor per tile, immediate offsets, floats beyond the run read as 0) and puts float f at row f / E
    constexpr int NX = EQ;
    uint32_t soff[NX];
#pragma unroll
    for (int qq = 0; qq < NX; ++qq) {
        const uint32_t f = lane + 64 * qq, row = (f * d.rcp_e) >> 16;
        soff[qq] = row < 32 ? f + row * (uint32_t)(ES - E) : 32 * 33 - 1;             // (beyond: a slot nobody reads)
    }
    const uint32_t lane4 = lane * 4;
    auto load_run = [&](const float *vp, int rows, float (&x)[NX]) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)vp, 0, rows * E * 4, 0x00020000);
#pragma unroll
        for (int qq = 0; qq < NX; ++qq)
            x[qq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + 256 * qq, 0, 0));
    };
    auto small_tile = [&](int m, int kt_, float (&hs)[4]) {
        const uint32_t kk = (uint32_t)min(32 * kt_ + j, NK - 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {                 // (the launcher points unused slots at valid memory, stride 0)
            const float *sp = d.small[f] + (int64_t)m * d.small_sm[f];                       // (scalar)
            hs[f] = sp[kk * (uint32_t)d.small_sk[f]];
        }
        asm volatile("" ::: "memory");
    };
    auto load_tile = [&](int m, int kt_, float (&x)[NX], float (&hs)[4]) {
        load_run(d.val + (int64_t)m * d.v_sm + 32 * kt_ * E, min(32, NK - 32 * kt_), x);
        small_tile(m, kt_, hs);
    };
    // FLAT: tile tt = rows 32 tt .. of the chunk's run; (pm, pk) = the plate element and k of the FIRST row of the tile
    // being loaded (scalar), advanced by 32 rows per tile (NK > 32: at most one wrap; a lane's row wraps once more)
    int pm = m0, pk = 0;
    auto small_flat = [&](float (&hs)[4]) {
        int lm = pm, lk = pk + j;
        if (lk >= NK) lk -= NK, ++lm;
        const bool in = lm < m1;
        const uint32_t mm = (uint32_t)((in ? lm : m1 - 1) - m0), kk = (uint32_t)(in ? lk : NK - 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float *sp = d.small[f] + (int64_t)m0 * d.small_sm[f];                      // (scalar)
            hs[f] = sp[mm * (uint32_t)d.small_sm[f] + kk * (uint32_t)d.small_sk[f]];
        }
        pk += 32;
        if (pk >= NK) pk -= NK, ++pm;
        asm volatile("" ::: "memory");
    };
    auto load_flat = [&](int tt, float (&x)[NX], float (&hs)[4]) {
        load_run(d.val + (int64_t)m0 * d.v_sm + 32 * tt * E, min(32, rows_total - 32 * tt), x);
        small_flat(hs);
    };
    float zc[NX], zn[NX], hc[4], hn[4];
    if (FLAT) {
        if (n_tiles > 0) load_flat(0, zc, hc);
    } else {
        // (unconditional -- an empty slice reads a valid tile nobody uses: behind a branch the compiler waits for the loads
        // inside it, in front of ~100 instructions of set-up that should run while they are in flight)
        load_tile(min(m0, d.M - 1), 0, zc, hc);
    }
    // (ready-made: requested LAST -- it is needed last, behind the first unit's A operand, and its 8 KB per wave keep the
    // CU's load path busy for hundreds of cycles that the loc rows and the value tile would otherwise queue behind: 7.84 us
    // against 7.97 requested first)
    if constexpr (TBL) load_table();
    // ---- the workgroup's B table: thread (wave w, lane) takes the event pairs w, w + 4, ... of its lane's (scale row,
    // half) in every scale tile; every load of a thread is issued before the first is used.  The log-normaliser
    // sum_e log(scale[s, e]) is collected on the way as per-(wave, lane) partial sums, added up in a fixed order behind
    // the barrier.  Branch-free and on the fast transcendental instructions (1 ulp): this is every wave's critical path.
    {
        unsigned *bw = reinterpret_cast<unsigned *>(lds);
#ifdef ALAN_TIMELINE
        NL_STAMP(14);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NL_STAMP(15);
#endif
        // (the loc rows on their way to the lanes, zero beyond the last event)
#pragma unroll
        for (int lw = 0; lw < NLW; ++lw)
            if (lane < 36) locl[lw * 36 + lane] = lane < E ? lrow[lw] : 0.f;
        // (the value tile's rows beyond the run are written every tile: the buffer loads read them as 0)
        if constexpr (!TBL) {
        const bool lsc = d.log_scale != 0;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const bool s_ok = 32 * (sg * NST + st) + j < NS;
            float lgp = 0.f;
#pragma unroll
            for (int qi = 0; qi < NQI; ++qi) {
                const int q = wave + 4 * qi;                                                           // (scalar)
                if (4 * qi + 3 < EQ || q < EQ) {
                    unsigned r[3];
                    nl_b_entry(xs[st][qi], lsc, s_ok && 2 * q + h < E, s_ok && q == EQ - 1 && h == slot_h, r, lgp);
                    // registers 3 q .. 3 q + 2 of the lane's operand: dword (3 q + i) & 3 of step (3 q + i) >> 2
                    const int v0 = 3 * q;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        bw[(st * NSTEP * 64 + lane) * 4 + ((v0 + i) >> 2) * 256 + ((v0 + i) & 3)] = r[i];
                }
            }
            lgp_l[(st * 4 + wave) * 64 + lane] = lgp;
        }
        if (NV > 3 * EQ) {                            // the registers beyond 3 EQ: zero
#pragma unroll
            for (int st = 0; st < NST; ++st)
#pragma unroll
                for (int v = 3 * EQ; v < NV; ++v)
                    if ((st * (NV - 3 * EQ) + v) % 4 == wave) bw[((st * NSTEP + (v >> 2)) * 64 + lane) * 4 + (v & 3)] = 0u;
        }
        }
    }
    NL_STAMP(1);
    if constexpr (TBL) {                              // (the loc rows in LDS are the wave's own: no workgroup barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
    NL_STAMP(2);
    if constexpr (!TBL) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            float lg = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) lg += lgp_l[(st * 4 + w) * 64 + j] + lgp_l[(st * 4 + w) * 64 + 32 + j];
            lgn[st] = lg + (float)E * NL_HALF_LOG_2PI;
        }
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) breg[step] = bt[step * 64 + lane];
    }
    float mreg[NLW][EQ];
#pragma unroll
    for (int lw = 0; lw < NLW; ++lw)
#pragma unroll
        for (int q = 0; q < EQ; ++q) mreg[lw][q] = locl[lw * 36 + h + 2 * q];
    constexpr int NU = NLW * NST;
    float accm[NU], mn[NU], sm[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) accm[u] = 0.f, mn[u] = inf, sm[u] = 0.f;
    int rem_a = NK;
    const float n_small_mask[4] = {d.n_small > 0 ? 1.f : 0.f, d.n_small > 1 ? 1.f : 0.f, d.n_small > 2 ? 1.f : 0.f,
                                   d.n_small > 3 ? 1.f : 0.f};
    int kt = 0, m = m0;
    NL_STAMP(3);
    for (int t = 0; t < n_tiles; ++t) {
        if (FLAT) {
            if (t + 1 < n_tiles) load_flat(t + 1, zn, hn);
        } else {
            int kt_n = kt + 1, m_n = m;
            if (kt_n == nkt) kt_n = 0, ++m_n;
            if (t + 1 < n_tiles) load_tile(m_n, kt_n, zn, hn);
        }
        const int valid = FLAT ? min(32, rows_total - 32 * t) : 32;
        const int bnd = FLAT ? min(rem_a, valid) : 32;
        const bool split = FLAT && bnd < valid;       // (wave-uniform)
        float mnb[NU], smb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) mnb[u] = inf, smb[u] = 0.f;
        const bool k_ok = FLAT ? j < valid : 32 * kt + j < NK;
        float hsum = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) hsum += n_small_mask[f] != 0.f ? hc[f] : 0.f;
#ifdef ALAN_TIMELINE
        if (t == 0) {                                 // (the first tile's loads have landed)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NL_STAMP(4);
        }
#endif
        float zv[EQ];
#pragma unroll
        for (int qq = 0; qq < NX; ++qq) tile[soff[qq]] = zc[qq];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < EQ; ++q) zv[q] = tile[j * ES + min(2 * q + h, E - 1)];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float nh = -hsum;
        const float slot = k_ok ? (nh > NL_BIG ? NL_BIG : nh) : NL_BIG;       // (a NaN small factor stays a NaN)
        // Units u = (loc row lw, scale tile st) of this value tile, software-pipelined: the matrix instructions of unit
        // u + 1 (and the A build that precedes them when it starts a new loc row) are issued in the same basic block as
        // the log-sum-exp of unit u, whose vector work then runs beside them; B of the unit after that is fetched from
        // the LDS table behind them (one unit ahead: the reads land during a whole log-sum-exp).
        unsigned areg[NV];
        auto build_a = [&](int lw) {
            // two events at a time on the packed f32 instructions (v_pk_add / v_pk_mul: two lanes' worth per issue slot; the
            // loop is bound by vector issue): 7 instructions per element where one at a time takes 9
#pragma unroll
            for (int q = 0; q + 1 < EQ; q += 2) {
                const f32x2v z2 = {zv[q], zv[q + 1]}, m2 = {mreg[lw][q], mreg[lw][q + 1]};
                const f32x2v df = z2 - m2;
                f32x2v a = df * df;
                if (q + 1 == EQ - 1) a[1] = h == slot_h ? slot : a[1];
                nl_split_a2(a, &areg[3 * q], &areg[3 * q + 3]);
            }
            if (EQ & 1) {
                const float df = zv[EQ - 1] - mreg[lw][EQ - 1];
                const float a = h == slot_h ? slot : df * df;
                nl_split_a(a, areg[3 * EQ - 3], areg[3 * EQ - 2], areg[3 * EQ - 1]);
            }
#pragma unroll
            for (int v = 3 * EQ; v < NV; ++v) areg[v] = 0u;
        };
        auto chain = [&]() {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                const u32x4v av = {areg[4 * step], areg[4 * step + 1], areg[4 * step + 2], areg[4 * step + 3]};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, av),
                                                              __builtin_bit_cast(bf16x8v, breg[step]), acc, 0, 0, 0);
                // The first instruction of a chain has C = 0 and a destination of its own: under -amdgpu-mfma-vgpr-form
                // hipcc (ROCm 7.2) lets that destination overlap a dead A or B operand (seen: v_mfma v[0:15], v[68:71],
                // v[0:3], 0), which a multi-pass MFMA does not survive.  An empty asm that takes the result and both operands keeps
                // them alive past it.
                if (step == 0) asm volatile("" ::"v"(acc[0]), "v"(av), "v"(breg[0]));
            }
            return acc;
        };
        auto fetch_b = [&](int st) {                  // (one scale tile: B never leaves its registers)
            if (NST > 1) {
#pragma unroll
                for (int step = 0; step < NSTEP; ++step) breg[step] = bt[(st * NSTEP + step) * 64 + lane];
            }
        };
        // acc[r] = -log2(e) (log-prob + small) of row (r & 3) + 8 (r >> 2) + 4 h of the tile, normaliser apart
        auto lse_plain = [&](int u, const f32x16 &acc) {
            float tmin = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[r]);
            const float mnew = fminf(mn[u], tmin);
            const float mf = mnew == inf ? 0.f : mnew;
            float ssum = sm[u] * __builtin_amdgcn_exp2f(mf - (mn[u] == inf ? mf : mn[u]));
            const f32x2v mf2 = {mf, mf};
            f32x2v part = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2v a2 = {acc[r], acc[r + 1]};
                const f32x2v d2 = mf2 - a2;
                const f32x2v e2 = {__builtin_amdgcn_exp2f(d2[0]), __builtin_amdgcn_exp2f(d2[1])};
                part += e2;
            }
            ssum += part[0] + part[1];
            mn[u] = mnew, sm[u] = ssum;
        };
        auto lse_split = [&](int u, const f32x16 &acc) {
            float ta = inf, tb = inf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool isa = 8 * g + 4 * h < bnd;
                const float t4 = fminf(fminf(acc[4 * g], acc[4 * g + 1]), fminf(acc[4 * g + 2], acc[4 * g + 3]));
                ta = isa ? fminf(ta, t4) : ta;
                tb = isa ? tb : fminf(tb, t4);
            }
            const float mnew = fminf(mn[u], ta);
            const float mfa = mnew == inf ? 0.f : mnew, mfb = tb == inf ? 0.f : tb;
            float ssa = sm[u] * __builtin_amdgcn_exp2f(mfa - (mn[u] == inf ? mfa : mn[u])), ssb = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool isa = 8 * g + 4 * h < bnd;
                const float mfx = isa ? mfa : mfb;
                float part = 0.f;
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) part += __builtin_amdgcn_exp2f(mfx - acc[r]);
                ssa += isa ? part : 0.f;
                ssb += isa ? 0.f : part;
            }
            mn[u] = mnew, sm[u] = ssa, mnb[u] = tb, smb[u] = ssb;
        };
        constexpr int VPG = (64 + NSTEP - 1) / NSTEP;    // vector instructions per matrix instruction (a log-sum-exp has ~64)
        auto units = [&](auto lse) {
            build_a(0);
            f32x16 cur = chain();                     // (breg: scale tile 0, fetched behind the previous tile's last unit)
            fetch_b(NST > 1 ? 1 : 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x16 nxt = cur;
                if (u + 1 < NU) {
                    const int lw1 = (u + 1) / NST, st1 = (u + 1) - lw1 * NST;
                    if (st1 == 0) build_a(lw1);
                    nxt = chain();
                    fetch_b((st1 + 1) % NST);
                }
                lse(u, cur);
                // the order the scheduler is asked for: one matrix instruction, the B read that refills its operand, then a
                // share of the log-sum-exp's vector work -- left alone it emits the eight dependent MFMAs back to back
                // (the wave then stalls through all of them) and the vector instructions behind
                if (u + 1 < NU) {
#pragma unroll
                    for (int i = 0; i < NSTEP; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (NST > 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
                    }
                }
                cur = nxt;
            }
        };
        if (split)
            units(lse_split);
        else
            units(lse_plain);
#ifdef ALAN_TIMELINE
        if (t == 0) NL_STAMP(5);
        if (t == n_tiles - 1) NL_STAMP(6);
#endif
        if (FLAT) rem_a -= bnd;
        if (FLAT ? rem_a == 0 : ++kt == nkt) {        // plate element done: join the two half-waves, add to the plate sum
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int lw = u / NST, st = u - lw * NST;
                // (v_permlane32_swap with both operands the same register: every lane gets the lower half-wave's value
                // and the upper half-wave's -- one vector instruction where a shuffle is an LDS round trip)
                const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mn[u]), __float_as_uint(mn[u]), false, false);
                const auto ps = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm[u]), __float_as_uint(sm[u]), false, false);
                const float mn1 = __uint_as_float(pm[0]), mn2 = __uint_as_float(pm[1]);
                const float sm1 = __uint_as_float(ps[0]), sm2 = __uint_as_float(ps[1]);
                const float mm = fminf(mn1, mn2);
                const float mf = mm == inf ? 0.f : mm;
                const float tot = sm1 * __builtin_amdgcn_exp2f(mf - (mn1 == inf ? mf : mn1)) +
                                  sm2 * __builtin_amdgcn_exp2f(mf - (mn2 == inf ? mf : mn2));
                // log(tot + eps) + max, in base 2 until the end (tot + eps >= eps: v_log_f32 needs no denormal care)
                float lse_m = (__builtin_amdgcn_logf(tot + Num<float>::eps) - mm) * NL_LN2 - lgn[st];
                if (mm >= 1e29f || mm == -inf) lse_m = __builtin_nanf("");      // every row masked / -inf, or a +inf term
                accm[u] += lse_m;
                const int s = 32 * (sg * NST + st) + j;
                if (d.lse && h == 0 && s < NS && l + lw < d.NL) d.lse[((int64_t)m * d.NL + l + lw) * NS + s] = lse_m;
                mn[u] = mnb[u], sm[u] = smb[u];
            }
            kt = 0, ++m;
            if (FLAT) rem_a = NK - (valid - bnd);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) zc[i] = zn[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) hc[f] = hn[f];
    }
    // ---- the four slices of the workgroup, added in slice order by wave 0 (both half-waves hold the sums: lanes 0-31 write)
    float *red = lgp_l + NST * 4 * 64;                // (the waves' tile areas: every wave is past its last tile read)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if (h == 0) red[(wave * NU + u) * 32 + j] = accm[u];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int lw = u / NST, st = u - lw * NST;
            const int s = 32 * (sg * NST + st) + j;
            float tot = ((red[u * 32 + j] + red[(NU + u) * 32 + j]) + red[(2 * NU + u) * 32 + j]) + red[(3 * NU + u) * 32 + j];
            if (h == 0 && s < NS && l + lw < d.NL) {
                d.part[((int64_t)bz * d.NL + l + lw) * NS + s] = tot;
            }
        }
    }
#ifdef ALAN_TIMELINE
    NL_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NL_STAMP(8);
    const int wid = ((bz * gy + by) * gx + bx) * 4 + wave;
    if (lane == 0 && wid < NL_TL_WAVES) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl[9] = tl_real, tl[10] = ((unsigned long long)xcc << 32) | hwid, tl[11] = (unsigned long long)n_tiles * NU;
        tl[12] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < NL_TL_SLOTS; ++i) nl_timeline[wid * NL_TL_SLOTS + i] = tl[i];
    }
#endif
}


}  // namespace alan
