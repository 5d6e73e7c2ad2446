// alan_normal_lse: the S-ML plate step with the factor producer fused in --
//
//   out[l, s] = sum_m  LSE_k( log N(value[m,k,:]; loc[l,:], scale[s,:]) + sum_f small_f[m,k] )  + add_const
//
// i.e. movielens' plate_1:  z[plate_1, K_z, d] ~ N(mu_z[K_mu, d], exp(psi_z)[K_psi, d]) with the -(log Q + log K)
// and data-likelihood terms as the small factors, log-sum-exp over K_z, sum over plate_1
// (TorchDimDist.py:127-162 + utils.py:147-152 + reduce_Ks.py:249-251 + utils.py:218-220 + logpq.py:149).
// The factor F[plate_1, K_mu, K_psi, K_z] (32 MB at K=30, 1.2 GB at K=100) is never written or read: the kernel is
// bound by its exps and MFMAs, not by HBM.
//
// Per-chunk partial sums go to the workspace; a small second stage adds the chunks (no float atomics).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "plan.h"
#include "normal_lse_x3.h"

namespace alan {

struct NLDesc {
    const float *val, *loc, *scl;
    float *part;                      // [n_chunks][NL][NS]
    float *lse;                       // optional [M][NL][NS]: the per-plate-element log-sum-exp (the backward's input)
    int32_t M, NK, NL, NS, E, m_chunk, n_small, log_scale;
    int64_t v_sm, v_sk, v_se, l_sl, l_se, s_ss, s_se;
    const float *small[4];
    int64_t small_sm[4], small_sk[4];
};

// On the matrix cores: for one (plate element m, loc row l) the block F[m, l, :, :] is a GEMM over the event dim;
// with v_mfma_f32_32x32x2_f32 computing D[i = k][j = s] = sum_e d2[(m,k), e] * w[s, e] the log-sum-exp over
// k runs DOWN the accumulator registers of a lane (16 rows per lane + one exchange between the two half-waves), the
// sum over m stays in a register, and nothing but out[l, s] partials is ever stored:
//   A (one VGPR per step): d2 of this lane's k row, rebuilt per (plate element, k tile) with two VALU ops per event
//      pair and used for NST chains of MFMAs -- the wave owns one loc row and NST tiles of 32 scale rows;
//   B (one VGPR per step and scale tile, held for the whole kernel): log2(e) * w of this lane's scale row, built by
//      the wave itself in registers (no LDS tables, no barriers);  C = 0 (an inline constant): the log-normaliser
//      does not depend on k, so it is subtracted once per plate element instead;
//   one extra event slot carries the small factors: A = -sum_f small_f[m,k], B = log2(e).
// D is therefore -log2(e) * (log-prob + small) up to the normaliser: the online log-sum-exp across k tiles runs on
// v_exp_f32 directly (base 2), and log(sum + eps) + max (utils.py:218-220) is taken in natural units at the end.
// The value rows of the next (m, k tile) are loaded while the current ones are multiplied.


// STAGE: value[m, 32 kt .. 32 kt + 31, :] is one contiguous run of 32 * E floats (unit event stride, rows E apart):
// the wave copies it into its own LDS tile with fully coalesced loads (lane i takes floats i, i + 64, ...) and reads
// its A operand from there.  Per-lane row loads of the same bytes -- 64 lanes, 64 different rows 72 bytes apart, one
// dword each, ten times per tile -- keep the CU's texture addresser busy for longer than the MFMAs take.

// NLW: loc rows per wave (2 when one scale tile covers all scale rows, K <= 32: the value tile's loads, its LDS
// stage and the small-factor sum are then shared by two tiles of work -- 110 of the 250 instructions of a tile).
// RAG: the last of the wave's NST scale tiles holds at most 16 scale rows (K = 100: 3 full tiles + 4 rows): it is
// computed as two 16 x 16 blocks by v_mfma_f32_16x16x4_f32 -- half the matrix time and half the exps of a 32 x 32 tile
// that would be 7/8 padding.  (STAGE only: its A operand has the event index spread over the four lane quarters and is
// read from the wave's LDS tile.)

//
// FLAT (STAGE, NLW = 1, NK > 32 not a multiple of 32, plate elements contiguous in memory): the k rows of a wave's
// plate elements are walked as ONE run of (m1 - m0) * NK rows in tiles of 32, so only the run's last tile is padded --
// per-element tiles leave the last of ceil(NK / 32) mostly empty (NK = 100: 4 rows of 32; a quarter of the launch).  A
// tile then holds the end of one plate element (rows below `bnd`) and the start of the next: two log-sum-exp states
// per unit, rows assigned by a compare per group of four accumulator registers (NK % 4 == 0), the finished element's
// state retired after the tile.
template <int EH, int NST, bool STAGE, int NLW = 1, bool RAG = false, bool FLAT = false>   // EH: MFMA steps >= ceil((E + 1) / 2) -- the event dim plus the small-factor slot
__global__ __launch_bounds__(256) void normal_lse_mfma_kernel(const NLDesc d) {
    static_assert(!FLAT || (STAGE && NLW == 1), "flat row tiling: staged loads, one loc row per wave");
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int NK = d.NK, E = d.E, NS = d.NS;
    const int nkt = (NK + 31) >> 5, nsg = (((NS + 31) >> 5) + NST - 1) / NST;
    const int nlg = (d.NL + NLW - 1) / NLW;           // groups of NLW loc rows
    const int q = blockIdx.x * 4 + wave;              // (group of loc rows, group of NST scale tiles) of this wave
    const bool wave_on = q < nlg * nsg;
    if (!wave_on) return;
    const int lgp = wave_on ? q / nsg : 0, sg = wave_on ? q - lgp * nsg : 0;
    const int l = lgp * NLW;                          // first loc row of the wave
    // the small factors ride in the LAST step's spare element: half-wave 0 when the events leave both elements of
    // that step free, else half-wave 1 (E = 2 EH - 1) -- one select per tile instead of one per step
    const bool slot_lane = h == (E > 2 * (EH - 1) ? 1 : 0);
    const float inf = __builtin_huge_valf();
    // ---- the wave's operands, in registers: B and the log-normaliser per scale tile, the loc row
    float breg[NST][EH], lgn[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int s = 32 * (sg * NST + st) + j;
        const bool s_ok = s < NS;
        float lg = 0.f;
#pragma unroll
        for (int step = 0; step < EH; ++step) {
            const int e = 2 * step + h;
            const float x = d.scl[(int64_t)min(s, NS - 1) * d.s_ss + (int64_t)min(e, E - 1) * d.s_se];
            const float w = d.log_scale ? 0.5f * expf(-2.f * x) : 0.5f / (x * x);
            breg[st][step] = (s_ok && e < E) ? w * NL_LOG2E : (s_ok && step == EH - 1 && slot_lane) ? NL_LOG2E : 0.f;
            lg += e < E ? (d.log_scale ? x : logf(x)) : 0.f;
        }
        lgn[st] = lg + __shfl_xor(lg, 32) + (float)E * 0.91893853320467274178f;
    }
    float mreg[NLW][EH];
#pragma unroll
    for (int lw = 0; lw < NLW; ++lw)
#pragma unroll
        for (int step = 0; step < EH; ++step) {
            const int e = 2 * step + h;
            const float x = d.loc[(int64_t)min(l + lw, d.NL - 1) * d.l_sl + (int64_t)min(e, E - 1) * d.l_se];
            mreg[lw][step] = e < E ? x : 0.f;
        }
    // RAG: operands of the 16-wide tile -- lane (r16 = lane & 15, q4 = lane >> 4) holds column s16 of B and event 4 t + q4
    constexpr int T16 = (EH + 1) / 2;                 // K = 4 steps covering the same 2 EH event slots
    const int r16 = lane & 15, q4 = lane >> 4;
    const int s16 = 32 * (sg * NST + NST - 1) + r16;
    float b16[T16], m16[T16], lgn16 = 0.f;
    if (RAG) {
        const bool ok = s16 < NS;
        float lg = 0.f;
#pragma unroll
        for (int tt = 0; tt < T16; ++tt) {
            const int e = 4 * tt + q4;
            const float x = d.scl[(int64_t)min(s16, NS - 1) * d.s_ss + (int64_t)min(e, E - 1) * d.s_se];
            const float w = d.log_scale ? 0.5f * expf(-2.f * x) : 0.5f / (x * x);
            b16[tt] = (ok && e < E) ? w * NL_LOG2E : (ok && e == E) ? NL_LOG2E : 0.f;
            lg += e < E ? (d.log_scale ? x : logf(x)) : 0.f;
            const float mu = d.loc[(int64_t)min(l, d.NL - 1) * d.l_sl + (int64_t)min(e, E - 1) * d.l_se];
            m16[tt] = e < E ? mu : 0.f;
        }
        lg += __shfl_xor(lg, 16);
        lg += __shfl_xor(lg, 32);
        lgn16 = lg + (float)E * 0.91893853320467274178f;
    }
    const int m0 = blockIdx.y * d.m_chunk, m1 = min(d.M, m0 + d.m_chunk);
    const int rows_total = (m1 - m0) * NK;            // FLAT: the wave's run of k rows
    const int n_tiles = wave_on ? (FLAT ? (rows_total + 31) >> 5 : (m1 - m0) * nkt) : 0;
    // Tile t = (plate element m0 + t / nkt, k tile t % nkt), walked with counters (no division in the loop).  Everything
    // that addresses a tile is the same for all four waves (they share blockIdx.y): a scalar base plus a 32-bit lane
    // offset.  Loads are issued from clamped, always valid addresses, all of them before anything waits; masks are
    // applied where the values are used, one iteration later.
    // STAGE: x[] holds this lane's share of the tile's contiguous run (floats lane + 64 q): EH loads cover the longest
    // run of this EH bucket (32 rows x (2 EH - 1) floats / 64 lanes); a shorter run re-reads its last float
    constexpr int NX = EH;
    float *tile = lds + wave * (32 * 33);
    uint32_t soff[NX];                                // STAGE: LDS slot of float lane + 64 q: row * (E + 1) + column
    if (STAGE) {
        if (FLAT || (NK & 31))                        // rows beyond NK are never written: keep them finite
            for (int i = lane; i < 32 * 33; i += 64) tile[i] = 0.f;
        // (row, column) of float lane + 64 q, stepped by 64 floats = (64 / E) rows + (64 % E) columns: one division
        const int r64 = 64 / E, c64 = 64 - r64 * E;
        int row = lane / E, col = lane - row * E;
#pragma unroll
        for (int qq = 0; qq < NX; ++qq) {
            soff[qq] = row < 32 ? row * (E + 1) + col : 32 * 33 - 1;                    // (beyond: a slot nobody reads)
            row += r64, col += c64;
            if (col >= E) col -= E, ++row;
        }
    }
    const uint32_t row_off = (uint32_t)j * (uint32_t)d.v_sk;               // !STAGE: this lane's row inside a tile
    auto load_tile = [&](int m, int kt_, float (&x)[NX], float (&hs)[4]) {
        const float *vp = d.val + (int64_t)m * d.v_sm + (int64_t)(32 * kt_) * d.v_sk;      // (uniform)
        const int rows = min(32, NK - 32 * kt_);
        if (STAGE) {
            const uint32_t lim = (uint32_t)(rows * E - 1);                                  // last valid float of the run
#pragma unroll
            for (int qq = 0; qq < NX; ++qq) x[qq] = vp[min((uint32_t)(lane + 64 * qq), lim)];
        } else {
            const uint32_t ro = j < rows ? row_off : (uint32_t)(rows - 1) * (uint32_t)d.v_sk;
#pragma unroll
            for (int step = 0; step < EH; ++step) x[step] = vp[ro + (uint32_t)min(2 * step + h, E - 1) * (uint32_t)d.v_se];
        }
        const uint32_t kk = (uint32_t)min(32 * kt_ + j, NK - 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {                 // (the launcher points unused slots at valid memory, stride 0)
            const float *sp = d.small[f] + (int64_t)m * d.small_sm[f];                       // (uniform)
            hs[f] = sp[kk * (uint32_t)d.small_sk[f]];
        }
        asm volatile("" ::: "memory");
    };
    // FLAT: tile tt = rows 32 tt .. of the run; (pm, pk) = the plate element and k of the FIRST row of the tile being
    // loaded (wave-uniform), advanced by 32 rows per tile (NK > 32: at most one wrap; a lane's row wraps once more)
    int pm = m0, pk = 0;
    auto load_flat = [&](int tt, float (&x)[NX], float (&hs)[4]) {
        const float *vp = d.val + (int64_t)m0 * d.v_sm + (int64_t)(32 * tt) * E;          // (uniform)
        const int rows = min(32, rows_total - 32 * tt);
        const uint32_t lim = (uint32_t)(rows * E - 1);
#pragma unroll
        for (int qq = 0; qq < NX; ++qq) x[qq] = vp[min((uint32_t)(lane + 64 * qq), lim)];
        int lm = pm, lk = pk + j;                     // this lane's row
        if (lk >= NK) lk -= NK, ++lm;
        const bool in = lm < m1;
        const uint32_t mm = (uint32_t)((in ? lm : m1 - 1) - m0), kk = (uint32_t)(in ? lk : NK - 1);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float *sp = d.small[f] + (int64_t)m0 * d.small_sm[f];                      // (uniform)
            hs[f] = sp[mm * (uint32_t)d.small_sm[f] + kk * (uint32_t)d.small_sk[f]];
        }
        pk += 32;
        if (pk >= NK) pk -= NK, ++pm;
        asm volatile("" ::: "memory");
    };
    float zc[NX], zn[NX], hc[4], hn[4];
    if (n_tiles > 0) {
        if (FLAT)
            load_flat(0, zc, hc);
        else
            load_tile(m0, 0, zc, hc);
    }
    constexpr int NU = NLW * NST;                     // units of work per value tile: (loc row, scale tile) pairs
    float accm[NU], mn[NU], sm[NU];                   // plate sum; running minimum of D (= -max, base 2) and sum 2^(mn - D)
#pragma unroll
    for (int u = 0; u < NU; ++u) accm[u] = 0.f, mn[u] = inf, sm[u] = 0.f;
    int rem_a = NK;                                   // FLAT: rows the current plate element still has to come
    const float n_small_mask[4] = {d.n_small > 0 ? 1.f : 0.f, d.n_small > 1 ? 1.f : 0.f, d.n_small > 2 ? 1.f : 0.f,
                                   d.n_small > 3 ? 1.f : 0.f};
    int kt = 0, m = m0;
    for (int t = 0; t < n_tiles; ++t) {
        if (FLAT) {
            if (t + 1 < n_tiles) load_flat(t + 1, zn, hn);
        } else {
            int kt_n = kt + 1, m_n = m;
            if (kt_n == nkt) kt_n = 0, ++m_n;
            if (t + 1 < n_tiles) load_tile(m_n, kt_n, zn, hn);
        }
        // FLAT: rows of this tile, and how many of them belong to the current plate element (the rest start the next)
        const int valid = FLAT ? min(32, rows_total - 32 * t) : 32;
        const int bnd = FLAT ? min(rem_a, valid) : 32;
        const bool split = FLAT && bnd < valid;       // (wave-uniform)
        // the state of the plate element that begins inside this tile (it becomes the current one when the tile is done)
        float mnb[NU], smb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) mnb[u] = inf, smb[u] = 0.f;
        // A operand.  No masks: pad events meet a zero in B; rows beyond NK (and plate elements' -inf small factors)
        // put +inf into the small-factor slot, which makes their whole row of D +inf = a log-prob of -inf
        const bool k_ok = FLAT ? j < valid : 32 * kt + j < NK;
        float hsum = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) hsum += n_small_mask[f] != 0.f ? hc[f] : 0.f;
        float a[EH], zv[EH];
        if (STAGE) {
            // (one wave, in-order LDS queue: last tile's reads precede these writes, these writes the reads below)
#pragma unroll
            for (int qq = 0; qq < NX; ++qq) tile[soff[qq]] = zc[qq];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int step = 0; step < EH; ++step) zv[step] = tile[j * (E + 1) + min(2 * step + h, E - 1)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int step = 0; step < EH; ++step) zv[step] = zc[step];
        }
        const float slot = k_ok ? -hsum : inf;
        if (RAG) {
            // the 16-wide tile: rows 16 b + r16 of the k tile, events 4 t + q4 (slot E carries the small factors)
            constexpr int u = NU - 1;
            float tmin = inf;
            f32x4v acc4[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float hsb = __shfl(slot, 16 * b + r16);           // (slot is per k row: lanes j = row, either half)
                acc4[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < T16; ++tt) {
                    const int e = 4 * tt + q4;
                    const float df = tile[(16 * b + r16) * (E + 1) + min(e, E - 1)] - m16[tt];
                    const float av = e == E ? hsb : df * df;
                    acc4[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b16[tt], acc4[b], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) tmin = fminf(tmin, acc4[b][r]);
            }
            if (split) {
                // rows 16 b + 4 q4 + r below bnd end the current plate element, the others begin the next
                float ta = inf, tb = inf;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool isa = 16 * b + 4 * q4 < bnd;            // (NK % 4 == 0: a lane's four rows go together)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ta = isa ? fminf(ta, acc4[b][r]) : ta;
                        tb = isa ? tb : fminf(tb, acc4[b][r]);
                    }
                }
                const float mnew = fminf(mn[u], ta);
                const float mfa = mnew == inf ? 0.f : mnew, mfb = tb == inf ? 0.f : tb;
                float ssa = sm[u] * __builtin_amdgcn_exp2f(mfa - (mn[u] == inf ? mfa : mn[u])), ssb = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool isa = 16 * b + 4 * q4 < bnd;
                    const float mfx = isa ? mfa : mfb;
                    float part = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) part += __builtin_amdgcn_exp2f(mfx - acc4[b][r]);
                    ssa += isa ? part : 0.f;
                    ssb += isa ? 0.f : part;
                }
                mn[u] = mnew, sm[u] = ssa, mnb[u] = tb, smb[u] = ssb;
            } else {
            const float mnew = fminf(mn[u], tmin);
            const float mf = mnew == inf ? 0.f : mnew;
            float ssum = sm[u] * __builtin_amdgcn_exp2f(mf - (mn[u] == inf ? mf : mn[u]));
            const f32x2v mf2 = {mf, mf};
            f32x2v part = {0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2v a2 = {acc4[b][r], acc4[b][r + 1]};
                    const f32x2v d2 = mf2 - a2;
                    const f32x2v e2 = {__builtin_amdgcn_exp2f(d2[0]), __builtin_amdgcn_exp2f(d2[1])};
                    part += e2;
                }
            ssum += part[0] + part[1];
            mn[u] = mnew, sm[u] = ssum;
            }
        }
#pragma unroll
        for (int u = 0; u < NU - (RAG ? 1 : 0); ++u) {
            const int lw = u / NST, st = u - lw * NST;
            if (st == 0) {
#pragma unroll
                for (int step = 0; step < EH; ++step) {
                    const float df = zv[step] - mreg[lw][step];
                    a[step] = df * df;
                }
                if (slot_lane) a[EH - 1] = slot;
            }
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int step = 0; step < EH; ++step)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step], breg[st][step], acc, 0, 0, 0);
            // acc[r] = -log2(e) (log-prob + small) of row k = 32 kt + (r & 3) + 8 (r >> 2) + 4 h (normaliser apart):
            // online log-sum-exp down the registers, branch-free (an infinite minimum is replaced by 0 where it
            // enters a difference: 2^(0 - inf) = 0, and the sum it would scale is 0)
            if (split) {
                // rows (r & 3) + 8 (r >> 2) + 4 h below bnd end the current plate element, the others begin the next
                float ta = inf, tb = inf;
#pragma unroll
                for (int g = 0; g < 4; ++g) {                 // (NK % 4 == 0: the four rows of a register group go together)
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
            } else {
            float tmin = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[r]);
            const float mnew = fminf(mn[u], tmin);
            const float mf = mnew == inf ? 0.f : mnew;
            float ssum = sm[u] * __builtin_amdgcn_exp2f(mf - (mn[u] == inf ? mf : mn[u]));
            // (register pairs: v_pk_add_f32 takes the differences and the partial sums two at a time)
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
            }
        }
        if (FLAT) rem_a -= bnd;
        if (FLAT ? rem_a == 0 : ++kt == nkt) {        // plate element done: join the two half-waves, add to the plate sum
            if (RAG) {                                // (the 16-wide tile's rows are spread over the four lane quarters)
                constexpr int u = NU - 1;
                float mm = mn[u], tot = sm[u];
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float mn2 = __shfl_xor(mm, o), sm2 = __shfl_xor(tot, o);
                    const float mx = fminf(mm, mn2);
                    const float mf = mx == inf ? 0.f : mx;
                    tot = tot * __builtin_amdgcn_exp2f(mf - (mm == inf ? mf : mm)) +
                          sm2 * __builtin_amdgcn_exp2f(mf - (mn2 == inf ? mf : mn2));
                    mm = mx;
                }
                float lse_m = logf(tot + Num<float>::eps) - mm * NL_LN2 - lgn16;
                if (mm == inf || mm == -inf) lse_m = __builtin_nanf("");
                accm[u] += lse_m;
                if (d.lse && lane < 16 && s16 < NS) d.lse[((int64_t)m * d.NL + l) * NS + s16] = lse_m;
                mn[u] = mnb[u], sm[u] = smb[u];           // (empty unless the tile was split)
            }
#pragma unroll
            for (int u = 0; u < NU - (RAG ? 1 : 0); ++u) {
                const int lw = u / NST, st = u - lw * NST;
                const float mn2 = __shfl_xor(mn[u], 32), sm2 = __shfl_xor(sm[u], 32);
                const float mm = fminf(mn[u], mn2);
                const float mf = mm == inf ? 0.f : mm;
                const float tot = sm[u] * __builtin_amdgcn_exp2f(mf - (mn[u] == inf ? mf : mn[u])) +
                                  sm2 * __builtin_amdgcn_exp2f(mf - (mn2 == inf ? mf : mn2));
                // log(sum + eps) + max in natural units (utils.py:218-220); an all -inf (or +inf) column gives NaN there
                float lse_m = logf(tot + Num<float>::eps) - mm * NL_LN2 - lgn[st];
                if (mm == inf || mm == -inf) lse_m = __builtin_nanf("");
                accm[u] += lse_m;
                const int s = 32 * (sg * NST + st) + j;
                if (d.lse && h == 0 && s < NS && l + lw < d.NL) d.lse[((int64_t)m * d.NL + l + lw) * NS + s] = lse_m;
                mn[u] = mnb[u], sm[u] = smb[u];
            }
            kt = 0, ++m;
            if (FLAT) rem_a = NK - (valid - bnd);     // (the rows of this tile that already belong to the next element)
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) zc[i] = zn[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) hc[f] = hn[f];
    }
#pragma unroll
    for (int u = 0; u < NU - (RAG ? 1 : 0); ++u) {
        const int lw = u / NST, st = u - lw * NST;
        const int s = 32 * (sg * NST + st) + j;
        if (wave_on && h == 0 && s < NS && l + lw < d.NL) d.part[((int64_t)blockIdx.y * d.NL + l + lw) * NS + s] = accm[u];
    }
    if (RAG && wave_on && lane < 16 && s16 < NS) d.part[((int64_t)blockIdx.y * d.NL + l) * NS + s16] = accm[NU - 1];
}

template <int EQ, int NST, int NLW, bool FLAT, bool TBL = false>
__global__ __launch_bounds__(256, 2) void normal_lse_x3_kernel(const X3Desc d) {
    normal_lse_x3_body<EQ, NST, NLW, FLAT, TBL>(d, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

}  // namespace alan

using namespace alan;

namespace {

struct NLPlan {
    int eh = 0, nst = 1, nlw = 1, m_chunk = 1, n_chunks = 1;
    bool rag = false, x3 = false;
    size_t part_bytes = 0;
    dim3 grid;
};

int plan_nl(const alan_normal_lse_desc_t &a, NLPlan &p) {
    if (!a.value || !a.loc || !a.scale || !a.out) return ALAN_ERR_BAD_DESC;
    if (a.M < 1 || a.NK < 1 || a.NL < 1 || a.NS < 1 || a.E < 1) return ALAN_ERR_BAD_DESC;
    if (a.n_small < 0 || a.n_small > 4) return ALAN_ERR_BAD_DESC;
    for (int f = 0; f < a.n_small; ++f)
        if (!a.small[f]) return ALAN_ERR_BAD_DESC;
    if (a.E > 32 || a.NK > 4096 || a.NS > 4096 || a.NL > (1 << 20) || a.M > (1 << 24)) return ALAN_ERR_UNSUPPORTED;
    // a wave per (loc row, group of up to 4 tiles of 32 scale rows), 4 per workgroup; the plate in chunks so that the
    // chip holds every wave at once (3 per SIMD)
    p.eh = nl_eq_for((int)a.E);
    const int64_t nst_total = (a.NS + 31) / 32;
    p.nst = nst_total >= 4 ? 4 : nst_total >= 2 ? 2 : 1;
    const int64_t nsg = (nst_total + p.nst - 1) / p.nst;
    p.nlw = (p.nst == 1 && a.NL >= 8) ? 2 : 1;
    // the last scale tile as a 16-wide one: a single group of tiles whose last holds at most 16 rows (NS = 100: 4)
    p.rag = p.nst > 1 && nsg == 1 && nst_total == p.nst && a.NS - 32 * (nst_total - 1) <= 16;
    // the bf16x3 kernel (default): contiguous value rows (its staged loads), the chunks added by a second launch
    static const int f32_knob = env_knob("ALAN_NLSE_F32");                            // ablation knob: 1 = the f32 MFMA kernel
    p.x3 = a.v_se == 1 && a.v_sk == a.E && f32_knob != 1 &&
           a.l_sl >= 0 && a.l_se >= 0 && a.s_ss >= 0 && a.s_se >= 0 &&                   // (its 32-bit lane offsets)
           a.NL * a.l_sl + a.E * a.l_se < (1ll << 31) && a.NS * a.s_ss + a.E * a.s_se < (1ll << 31) &&
           a.v_sm >= 0 && a.v_sm < (1ll << 31) && a.NL <= 65535;
    for (int f = 0; f < a.n_small; ++f)
        p.x3 = p.x3 && a.small_sm[f] >= 0 && a.small_sk[f] >= 0 && a.small_sm[f] < (1ll << 31) && a.small_sk[f] < (1ll << 31) &&
               a.NK * a.small_sk[f] < (1ll << 31);
    if (p.x3 && p.eh == 17 && p.nst == 4) p.nst = 2;                                  // (its B table: 13 KB per scale tile)
    const int64_t nsg_x = (nst_total + p.nst - 1) / p.nst;
    if (p.x3) {
        // One workgroup per (scale-tile group, loc-row group, four slices of the plate).  The kernel holds two workgroups
        // per CU (its registers: two waves per SIMD): a workgroup that has to wait for a free slot starts a whole wave
        // lifetime late (the round-3 timeline at K = 30: 385 of 2,250 waves entered 9 us after the rest), and one wave
        // per SIMD has nothing to hide its own latencies behind (668 ns per unit against 2 x 244) -- so as many slices
        // as give at most 512 workgroups.
        const int64_t per = nsg_x * ((a.NL + p.nlw - 1) / p.nlw);
        static const int blocks_knob_x = env_knob("ALAN_NLSE_BLOCKS");                // tuning knob: target workgroups
        const int64_t target_x = blocks_knob_x != ENV_UNSET ? std::max(1, blocks_knob_x) : 512;
        int64_t ncg = std::max<int64_t>(1, target_x / std::max<int64_t>(1, per));
        ncg = std::min<int64_t>(ncg, std::min<int64_t>(65535, (a.M + 3) / 4));
        p.n_chunks = (int)ncg;
        p.m_chunk = (int)((a.M + 4 * ncg - 1) / (4 * ncg));       // (the largest slice: what flat row tiling's offset check needs)
        p.part_bytes = (size_t)p.n_chunks * a.NL * a.NS * sizeof(float);
        p.grid = dim3((uint32_t)nsg_x, (uint32_t)((a.NL + p.nlw - 1) / p.nlw), (uint32_t)ncg);
        return ALAN_OK;
    }
    const int64_t gx = (((a.NL + p.nlw - 1) / p.nlw) * nsg + 3) / 4;
    int64_t target = 768;                                                    // workgroups (x 4 waves)
    static const int blocks_knob = env_knob("ALAN_NLSE_BLOCKS");                      // tuning knob
    if (blocks_knob != ENV_UNSET) target = std::max(1, blocks_knob);
    int64_t nch = std::max<int64_t>(1, std::min<int64_t>(a.M, target / std::max<int64_t>(1, gx)));
    nch = std::min<int64_t>(nch, 65535);
    p.m_chunk = (int)((a.M + nch - 1) / nch);
    if (blocks_knob == ENV_UNSET) {
        // Balance: workgroups spread evenly over the 256 CUs, so the launch lasts as long as the CU with the most of
        // them -- ceil(workgroups / 256) x (plate elements per workgroup).  Among the chunkings that keep the chip
        // between ~2 and ~4 waves per SIMD take the cheapest, fewer workgroups on a tie (K=100, M=75: 475 workgroups of
        // 4 users instead of 625 of 3 -- 98 against 112 us; M=300 and the K=30 launch keep their measured optimum).
        const int64_t lo = 450, hi = 1100;
        int64_t best = -1, best_cost = 0;
        for (int64_t mc = a.M; mc >= 1; --mc) {
            const int64_t chunks = (a.M + mc - 1) / mc, wgs = gx * chunks;
            if (chunks > 65535 || wgs < lo || wgs > hi) continue;
            const int64_t cost = ((wgs + 255) / 256) * mc;
            if (best < 0 || cost < best_cost) best = mc, best_cost = cost;
        }
        if (best > 0) p.m_chunk = (int)best;
    }
    p.n_chunks = (int)((a.M + p.m_chunk - 1) / p.m_chunk);
    p.part_bytes = (size_t)p.n_chunks * a.NL * a.NS * sizeof(float);
    p.grid = dim3((uint32_t)gx, (uint32_t)p.n_chunks);
    return ALAN_OK;
}

}  // namespace

namespace alan {

// What a launch of the bf16x3 kernel needs:
// ALAN_ERR_UNSUPPORTED when the kernel declines the call (the f32 kernel, or the caller's other route, takes it).
int nl_x3_prepare(const alan_normal_lse_desc_t &a, void *part, X3Prep &o) {
    NLPlan p;
    const int rc = plan_nl(a, p);
    if (rc != ALAN_OK) return rc;
    if (!p.x3) return ALAN_ERR_UNSUPPORTED;
    bool flat = p.nlw == 1 && a.NK > 32 && (a.NK & 31) != 0 && (a.NK & 3) == 0 && a.v_sm == a.NK * a.E;
    for (int f = 0; f < a.n_small; ++f)
        flat = flat && (int64_t)p.m_chunk * a.small_sm[f] + a.NK * a.small_sk[f] < (1ll << 31);
    X3Desc &x = o.x;
    std::memset(&x, 0, sizeof(x));
    x.val = (const float *)a.value, x.loc = (const float *)a.loc, x.scl = (const float *)a.scale;
    o.tbl = a.scale_table != nullptr && p.nst == 1;                 // (a table the kernel cannot use is ignored: scale is there)
    x.tbl = o.tbl ? (const u32x4v *)a.scale_table : nullptr;
    x.part = (float *)part, x.lse = (float *)a.lse_out;
    x.M = (int)a.M, x.NK = (int)a.NK, x.NL = (int)a.NL, x.NS = (int)a.NS, x.E = (int)a.E, x.n_sub = 4 * p.n_chunks;
    x.n_small = a.n_small, x.log_scale = a.log_scale;
    x.v_sm = (int32_t)a.v_sm, x.l_sl = (int32_t)a.l_sl, x.l_se = (int32_t)a.l_se, x.s_ss = (int32_t)a.s_ss, x.s_se = (int32_t)a.s_se;
    for (int f = 0; f < 4; ++f) {
        const bool used = f < a.n_small;
        x.small[f] = used ? (const float *)a.small[f] : (const float *)a.value;      // (unused: any valid address)
        x.small_sm[f] = used ? (int32_t)a.small_sm[f] : 0;
        x.small_sk[f] = used ? (int32_t)a.small_sk[f] : 0;
    }
    x.nkt = (x.NK + 31) / 32, x.nlg = (x.NL + p.nlw - 1) / p.nlw;
    x.m_q = x.M / x.n_sub, x.m_r = x.M % x.n_sub;
    x.rcp_e = (65536u + (uint32_t)x.E - 1) / (uint32_t)x.E;
    o.gx = p.grid.x, o.gy = p.grid.y, o.gz = p.grid.z;
    o.eq = p.eh, o.nst = p.nst, o.nlw = p.nlw, o.flat = flat, o.n_chunks = p.n_chunks;
    const int nstep = (3 * o.eq + 3) / 4;
    o.lds = ((size_t)o.nst * nstep * 64 * 4 + (size_t)o.nst * 4 * 64 + 4 * (32 * 33 + o.nlw * 36)) * sizeof(float);
    return ALAN_OK;
}

}  // namespace alan

#ifdef ALAN_TIMELINE
extern "C" int alan_nlse_timeline_read(unsigned long long *host_out, int n_waves) {
    if (n_waves > NL_TL_WAVES) n_waves = NL_TL_WAVES;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(nl_timeline), sizeof(unsigned long long) * n_waves * NL_TL_SLOTS) == hipSuccess
               ? NL_TL_SLOTS : -1;
}
#endif

extern "C" size_t alan_normal_lse_workspace_bytes(const alan_normal_lse_desc_t *a) {
    if (!a) return 0;
    NLPlan p;
    if (plan_nl(*a, p) != ALAN_OK) return 0;
    return (p.part_bytes + 255) & ~(size_t)255;
}

extern "C" size_t alan_normal_lse_table_bytes(const alan_normal_lse_desc_t *a) {
    if (!a) return 0;
    NLPlan p;
    if (plan_nl(*a, p) != ALAN_OK || !p.x3 || p.nst != 1) return 0;
    return nl_table_bytes_for(p.eh);
}

extern "C" int64_t alan_normal_lse_n_partials(const alan_normal_lse_desc_t *a) {
    if (!a) return 0;
    NLPlan p;
    if (plan_nl(*a, p) != ALAN_OK) return 0;
    return p.n_chunks;
}

extern "C" int alan_normal_lse(const alan_normal_lse_desc_t *a, void *workspace, size_t workspace_bytes,
                               void *stream_) {
    if (!a) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    NLPlan p;
    int rc = plan_nl(*a, p);
    if (rc != ALAN_OK) return rc;
    const bool keep = a->keep_partials != 0;
    if ((uintptr_t)a->scale_table & 15) return ALAN_ERR_BAD_DESC;
    if (!keep && (!workspace || workspace_bytes < p.part_bytes)) return ALAN_ERR_WORKSPACE;
    NLDesc d;
    std::memset(&d, 0, sizeof(d));
    d.val = (const float *)a->value;
    d.loc = (const float *)a->loc;
    d.scl = (const float *)a->scale;
    d.part = keep ? (float *)a->out : (float *)workspace;
    d.lse = (float *)a->lse_out;
    d.M = (int)a->M, d.NK = (int)a->NK, d.NL = (int)a->NL, d.NS = (int)a->NS, d.E = (int)a->E;
    d.m_chunk = p.m_chunk, d.n_small = a->n_small, d.log_scale = a->log_scale;
    d.v_sm = a->v_sm, d.v_sk = a->v_sk, d.v_se = a->v_se;
    d.l_sl = a->l_sl, d.l_se = a->l_se, d.s_ss = a->s_ss, d.s_se = a->s_se;
    for (int f = 0; f < 4; ++f) {
        const bool used = f < a->n_small;
        d.small[f] = used ? (const float *)a->small[f] : (const float *)a->value;    // (unused: any valid address)
        d.small_sm[f] = used ? a->small_sm[f] : 0;
        d.small_sk[f] = used ? a->small_sk[f] : 0;
    }
    // the staged variant needs the tile's 32 rows to be one contiguous run of the value tensor
    const bool stage = a->v_se == 1 && a->v_sk == a->E;
    const bool rag = p.rag && stage;
    const size_t lds = 4 * 32 * 33 * sizeof(float);               // the waves' value tiles
    auto launch = [&](auto kern) {
        ALAN_LAUNCH_EXT(kern, p.grid, dim3(256), lds, stream, (hipEvent_t)a->ev_start, (hipEvent_t)a->ev_stop, 0, d);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
    // flat row tiling (see the kernels): the plate elements' k rows as one run
    bool flat = stage && p.nlw == 1 && a->NK > 32 && (a->NK & 31) != 0 && (a->NK & 3) == 0 && a->v_sm == a->NK * a->E;
                                                  // (NK % 4: a lane's groups of four accumulator rows never straddle two elements)
    for (int f = 0; f < a->n_small; ++f)          // (32-bit lane offsets into a small factor, inside one chunk of the plate)
        flat = flat && a->small_sm[f] >= 0 && a->small_sk[f] >= 0 &&
               (int64_t)p.m_chunk * a->small_sm[f] + a->NK * a->small_sk[f] < (1ll << 31);
    if (p.x3) {
        X3Prep xp;
        rc = nl_x3_prepare(*a, d.part, xp);
        if (rc != ALAN_OK) return rc;
        const X3Desc &x = xp.x;
        flat = xp.flat;
        const dim3 grid3 = p.grid;
        auto launch_x3 = [&](auto kern, int eq, int nst, int nlw) {
            const size_t lds_x = xp.lds;
            if (lds_x > 64 * 1024 &&
                hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x) != hipSuccess)
                return ALAN_ERR_LAUNCH;
            ALAN_LAUNCH_EXT(kern, grid3, dim3(256), lds_x, stream, (hipEvent_t)a->ev_start, (hipEvent_t)a->ev_stop, 0, x);
            return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
        };
#define X3_PICK(EQV)                                                                                                   \
    case EQV:                                                                                                          \
        rc = p.nst == 4   ? (flat ? launch_x3(normal_lse_x3_kernel<EQV, 4, 1, true>, EQV, 4, 1)                        \
                                  : launch_x3(normal_lse_x3_kernel<EQV, 4, 1, false>, EQV, 4, 1))                      \
             : p.nst == 2 ? (flat ? launch_x3(normal_lse_x3_kernel<EQV, 2, 1, true>, EQV, 2, 1)                        \
                                  : launch_x3(normal_lse_x3_kernel<EQV, 2, 1, false>, EQV, 2, 1))                      \
             : xp.tbl ? (p.nlw == 2 ? launch_x3(normal_lse_x3_kernel<EQV, 1, 2, false, true>, EQV, 1, 2)               \
                                    : (flat ? launch_x3(normal_lse_x3_kernel<EQV, 1, 1, true, true>, EQV, 1, 1)        \
                                            : launch_x3(normal_lse_x3_kernel<EQV, 1, 1, false, true>, EQV, 1, 1)))     \
             : p.nlw == 2 ? launch_x3(normal_lse_x3_kernel<EQV, 1, 2, false>, EQV, 1, 2)                               \
                          : (flat ? launch_x3(normal_lse_x3_kernel<EQV, 1, 1, true>, EQV, 1, 1)                        \
                                  : launch_x3(normal_lse_x3_kernel<EQV, 1, 1, false>, EQV, 1, 1));                     \
        break;
        switch (p.eh) {
            X3_PICK(4)
            X3_PICK(8)
            X3_PICK(10)
            X3_PICK(12)
            default:
                X3_PICK(17)
        }
#undef X3_PICK
    } else {
#define NL_PICK(EHV, NSTV) (rag && NSTV > 1 ? (flat ? launch(normal_lse_mfma_kernel<EHV, NSTV, true, 1, (NSTV > 1), true>) \
                                                    : launch(normal_lse_mfma_kernel<EHV, NSTV, true, 1, (NSTV > 1)>)) \
                            : stage ? (flat ? launch(normal_lse_mfma_kernel<EHV, NSTV, true, 1, false, true>) \
                                            : launch(normal_lse_mfma_kernel<EHV, NSTV, true>)) \
                                    : launch(normal_lse_mfma_kernel<EHV, NSTV, false>))
#define NL_PICK2(EHV) (stage ? launch(normal_lse_mfma_kernel<EHV, 1, true, 2>) : launch(normal_lse_mfma_kernel<EHV, 1, false, 2>))
#define NL_CASE(EHV)                                                                                   \
    case EHV:                                                                                          \
        rc = p.nst == 4 ? NL_PICK(EHV, 4) : p.nst == 2 ? NL_PICK(EHV, 2) : p.nlw == 2 ? NL_PICK2(EHV) : NL_PICK(EHV, 1); \
        break;
    switch (p.eh) {
        NL_CASE(4)
        NL_CASE(8)
        NL_CASE(10)
        NL_CASE(12)
        default:
            rc = p.nst == 4 ? NL_PICK(17, 4) : p.nst == 2 ? NL_PICK(17, 2) : p.nlw == 2 ? NL_PICK2(17) : NL_PICK(17, 1);
    }
#undef NL_PICK
#undef NL_PICK2
#undef NL_CASE
    }
    if (rc != ALAN_OK) return rc;
    if (keep) return ALAN_OK;                         // (the chunks are the caller's to add)

    // ---- second stage: out[l, s] = sum_chunk part[chunk, l, s] + add_const
    Canon s2;
    s2.nf = 1;
    s2.dominant = 0;
    s2.f[0].p = workspace;
    s2.f[0].dtype = ALAN_F32;
    s2.f[0].scale = 1.f;
    s2.w.p = nullptr;
    s2.l.p = nullptr;
    s2.o.p = a->out;
    s2.o.dtype = ALAN_F32;
    s2.o.scale = 1.f;
    for (int j = 0; j < MAXD; ++j) s2.f[0].ks[j] = s2.f[0].rs[j] = s2.o.ks[j] = 0;
    s2.nk = 0, s2.nr = 0, s2.n_out = 1, s2.n_red = 1;
    auto keep_dim = [&](int64_t size, int64_t src, int64_t dst) {
        if (size <= 1) return;
        s2.ksize[s2.nk] = size, s2.f[0].ks[s2.nk] = src, s2.o.ks[s2.nk] = dst, s2.kplate[s2.nk] = false;
        ++s2.nk;
        s2.n_out *= size;
    };
    if (p.n_chunks > 1) {
        s2.rsize[0] = p.n_chunks;
        s2.f[0].rs[0] = a->NL * a->NS;
        s2.nr = 1;
        s2.n_red = p.n_chunks;
    }
    keep_dim(a->NL, a->NS, a->o_sl);
    keep_dim(a->NS, 1, a->o_ss);
    s2.red_contig = false;
    s2.keep_contig = true;
    GroupDesc gd;
    GroupLaunch gl;
    rc = plan_group(s2, ALAN_F32, a->add_const, gd, gl);
    if (rc != ALAN_OK) return rc;
    rc = try_launch_small(s2, gd, gl, ALAN_MODE_SUM, ALAN_F32, stream, EvPair());
    if (rc == ALAN_ERR_UNSUPPORTED) rc = launch_group(gd, gl, ALAN_MODE_SUM, ALAN_F32, stream);
    return rc;
}
