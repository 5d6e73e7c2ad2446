// The scale table of the fused plate step: the B operand of normal_lse_x3_kernel (normal_lse_x3.h) for ONE tile of 32 scale
// rows, built ahead of the launch by a problem of mode ALAN_MODE_NORMAL_TABLE that rides in the producers' launch
// (reduce_small_multi_kernel) -- the plate step's waves then start with eight 16-byte loads where they built the table in
// LDS behind a barrier (round 4's timeline at K = 30: 1.4 us of a 6.6 us wave lifetime).  gfx950 only.
//
// Layout (what alan_normal_lse_table_bytes sizes): NSTEP x 64 lanes x 16 bytes -- lane (j = scale row, h = half) of MFMA
// step t holds registers 4 t .. 4 t + 3 of its B operand -- then 32 floats: the rows' log-normalisers
// sum_e log(scale[s, e]) + E log(2 pi) / 2.  Both builders (here and the kernel's own) share the arithmetic below, term by
// term and in the same order: a launch with a table gives the bits of one without.
#pragma once
#include "common.h"

namespace alan {

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
constexpr float NL_LOG2E = 1.44269504088896340736f, NL_LN2 = 0.69314718055994530942f;
constexpr float NL_HALF_LOG_2PI = 0.91893853320467274178f;

// Events per lane half incl. the small-factor slot, rounded up to the kernel's instantiations (2 EQ >= E + 1).
__host__ __device__ inline int nl_eq_for(int E) {
    const int need = (E + 2) / 2;
    return need <= 4 ? 4 : need <= 8 ? 8 : need <= 10 ? 10 : need <= 12 ? 12 : 17;
}
__host__ __device__ inline int nl_nstep_for(int eq) { return (3 * eq + 3) / 4; }
inline size_t nl_table_bytes_for(int eq) { return (size_t)nl_nstep_for(eq) * 64 * 16 + 32 * sizeof(float); }

// (lo, hi) -> one register of two bf16, round-to-nearest-even: v_cvt_pk_bf16_f32.  A vector cast, not inline asm: the
// compiler pads the wait states between a vector instruction's result and a matrix instruction that reads it as A or
// B only when it knows what wrote the register -- behind an asm the MFMA read a stale operand (seen: the first tile
// of a pipelined chain wrong, the later tiles, which reuse the same A registers, right).
__device__ __forceinline__ unsigned nl_cvt_pk(float lo, float hi) {
    const f32x2v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ void nl_split_a(float x, unsigned &r1, unsigned &r2, unsigned &r3) {
    r1 = nl_cvt_pk(x, x);                                         // (h, h)
    const float hf = __uint_as_float(r1 & 0xffff0000u);
    const float e1 = x - hf;
    r2 = nl_cvt_pk(e1, hf);                                       // (m, h)
    const float mf = __uint_as_float(r2 << 16);
    const float e2 = e1 - mf;
    r3 = nl_cvt_pk(e2, e1);                                       // (l, m)
}
// (two elements: the subtractions packed)
__device__ __forceinline__ void nl_split_a2(f32x2v x, unsigned *ra, unsigned *rb) {
    ra[0] = nl_cvt_pk(x[0], x[0]);
    rb[0] = nl_cvt_pk(x[1], x[1]);
    const f32x2v hf = {__uint_as_float(ra[0] & 0xffff0000u), __uint_as_float(rb[0] & 0xffff0000u)};
    const f32x2v e1 = x - hf;
    ra[1] = nl_cvt_pk(e1[0], hf[0]);
    rb[1] = nl_cvt_pk(e1[1], hf[1]);
    const f32x2v mf = {__uint_as_float(ra[1] << 16), __uint_as_float(rb[1] << 16)};
    const f32x2v e2 = e1 - mf;
    ra[2] = nl_cvt_pk(e2[0], e1[0]);
    rb[2] = nl_cvt_pk(e2[1], e1[1]);
}
__device__ __forceinline__ void nl_split_b(float x, unsigned &r1, unsigned &r2, unsigned &r3) {
    const unsigned hh = nl_cvt_pk(x, x);
    const float hf = __uint_as_float(hh & 0xffff0000u);
    const float e1 = x - hf;
    const unsigned mm = nl_cvt_pk(e1, e1);
    const float mf = __uint_as_float(mm & 0xffff0000u);
    const unsigned ll = nl_cvt_pk(e1 - mf, e1 - mf);
    r1 = (hh & 0xffffu) | (mm & 0xffff0000u);                     // (h, m)
    r2 = (hh & 0xffffu) | (ll & 0xffff0000u);                     // (h, l)
    r3 = r1;                                                      // (h, m)
}

// One entry of the table: x = scale[s, 2 q + h] (or its log) -> the three registers 3 q .. 3 q + 2 of lane (s, h), and the
// entry's share of the row's log-normaliser added to lgp.  ev: a real event of a real row; slot: the small-factor slot of a
// real row (B = log2(e): the A operand there holds -(sum of the small factors)).
__device__ __forceinline__ void nl_b_entry(float x, bool lsc, bool ev, bool slot, unsigned (&r)[3], float &lgp) {
    // log2(e) / (2 sigma^2); from log(sigma) = x: 2^(-2 log2(e) x) log2(e) / 2
    const float w = lsc ? __builtin_amdgcn_exp2f(-2.f * NL_LOG2E * x) * (0.5f * NL_LOG2E)
                        : (0.5f * NL_LOG2E) * __builtin_amdgcn_rcpf(x * x);
    const float bval = ev ? w : slot ? NL_LOG2E : 0.f;
    lgp += ev ? (lsc ? x : __builtin_amdgcn_logf(x) * NL_LN2) : 0.f;
    nl_split_b(bval, r[0], r[1], r[2]);
}

// The table of one tile of scale rows, by ONE workgroup of 256 threads (all of them call; a barrier inside).  Thread (wave w,
// lane) takes the event pairs w, w + 4, ... of its lane's (row, half), as the kernel's own builder does.
__device__ inline void nl_table_block(const float *scl, int s_ss, int s_se, int NS, int E, bool lsc, unsigned *tbl) {
    __shared__ float lgp_s[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int EQ = nl_eq_for(E), NSTEP = nl_nstep_for(EQ), NV = 4 * NSTEP;
    const int slot_h = E > 2 * (EQ - 1) ? 1 : 0;
    const bool s_ok = j < NS;
    const int joff = min(j, NS - 1) * s_ss;
    float lgp = 0.f;
    for (int q = wave; q < EQ; q += 4) {
        const float x = scl[joff + min(2 * q + h, E - 1) * s_se];
        unsigned r[3];
        nl_b_entry(x, lsc, s_ok && 2 * q + h < E, s_ok && q == EQ - 1 && h == slot_h, r, lgp);
        const int v0 = 3 * q;
#pragma unroll
        for (int i = 0; i < 3; ++i) tbl[lane * 4 + ((v0 + i) >> 2) * 256 + ((v0 + i) & 3)] = r[i];
    }
    for (int v = 3 * EQ + wave; v < NV; v += 4) tbl[((v >> 2) * 64 + lane) * 4 + (v & 3)] = 0u;    // the registers beyond 3 EQ
    lgp_s[tid] = lgp;
    __syncthreads();
    if (tid < 32) {
        float lg = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) lg += lgp_s[w * 64 + tid] + lgp_s[w * 64 + 32 + tid];
        reinterpret_cast<float *>(tbl)[NSTEP * 256 + tid] = lg + (float)E * NL_HALF_LOG_2PI;
    }
}

}  // namespace alan
