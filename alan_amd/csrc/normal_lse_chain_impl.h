// The chained launch's kernel (see normal_lse_chain.hip) and its per-event-length launcher: one translation unit per
// event-length bucket (normal_lse_chain_eq*.hip) so that they compile side by side.
#pragma once
#include <cstring>

#include "plan.h"
#include "small_device.h"
#include "normal_lse_x3.h"

namespace alan {

constexpr int CHAIN_TAIL = 2;         // steps of the parent's contraction

struct ChainArgs {
    int32_t n_aux;                    // workgroups [0, n_aux): the prelude
    int32_t body_waits;               // a small factor of the body is written by the prelude: its workgroups poll
    uint32_t pre_blocks;              // its problems' workgroups, dealt round-robin to those
    uint32_t gx, gy, n_main;
    FastDiv gxd, gyd;
    int32_t *state;                   // [0] prelude arrivals, [1] arrivals of all, [2] sticky: a poll ran out
    int32_t n_tail;
    int32_t tail_mode[CHAIN_TAIL], tail_logG[CHAIN_TAIL], tail_block[CHAIN_TAIL];
    uint32_t tail_blocks[CHAIN_TAIL];
};

template <bool BLOCK>
__device__ __forceinline__ void chain_tail_step(const SmallDesc &d, int mode, int logG, uint32_t vb) {
    if (mode == ALAN_MODE_LSE)
        small_body<ALAN_MODE_LSE, BLOCK>(d, logG, vb);
    else
        small_body<ALAN_MODE_SUM, BLOCK>(d, logG, vb);
}

// (the kernel argument: body descriptor | chain arguments | prelude | tail steps -- the last two are read through the
// kernel-argument segment at their byte offsets, see small_multi_block)
struct ChainKernArg {
    X3Desc d;
    ChainArgs c;
    X3Recipes rec;
    SmallMultiT<CHAIN_MULTI> pre;
    SmallDesc tail[CHAIN_TAIL];
};
static_assert(sizeof(ChainKernArg) <= 4096, "kernel arguments: 4 KB");

// SYNC = false: nothing inside the launch waits for anything else in it -- every small factor of the body is computed
// in its tiles (REC) and there is no tail, so the prelude's workgroups only have to finish before the launch does: plain
// stores, no counters, no polls.  (What the hand-offs cost, measured: tools/chain_parts.py.)
template <int EQ, int NST, int NLW, bool FLAT, bool REC, bool SYNC = true>
__global__ __launch_bounds__(256, 2) void normal_lse_x3_chain_kernel(const ChainKernArg a) {
    const ChainArgs &c = a.c;
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x;
    __shared__ int is_last;
    if (!SYNC) {
        if ((int)wg < c.n_aux) {
            for (uint32_t vb = wg; vb < c.pre_blocks; vb += (uint32_t)c.n_aux) {
                small_multi_block<false, CHAIN_MULTI>(offsetof(ChainKernArg, pre), vb);
                __syncthreads();
            }
        } else {
            const uint32_t r = wg - (uint32_t)c.n_aux;
            const uint32_t q1 = fd_div(r, c.gxd), bx = r - q1 * c.gx;
            const uint32_t bz = fd_div(q1, c.gyd), by = q1 - bz * c.gy;
            normal_lse_x3_body<EQ, NST, NLW, FLAT, false, REC>(a.d, (int)bx, (int)by, (int)bz, (int)c.gx, (int)c.gy, X3Chain(), a.rec);
        }
        return;
    }
    if ((int)wg < c.n_aux) {
        for (uint32_t vb = wg; vb < c.pre_blocks; vb += (uint32_t)c.n_aux) {
            small_multi_block<true, CHAIN_MULTI>(offsetof(ChainKernArg, pre), vb);
            __syncthreads();                                  // (the block-wide combine's LDS words, before the next problem)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains, then one lane signals
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(c.state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const uint32_t r = wg - (uint32_t)c.n_aux;
        const uint32_t q1 = fd_div(r, c.gxd), bx = r - q1 * c.gx;
        const uint32_t bz = fd_div(q1, c.gyd), by = q1 - bz * c.gy;
        X3Chain ch;
        ch.pre_done = c.state, ch.pre_fail = c.state + 2, ch.pre_n = c.body_waits ? c.n_aux : 0;
        normal_lse_x3_body<EQ, NST, NLW, FLAT, true, REC>(a.d, (int)bx, (int)by, (int)bz, (int)c.gx, (int)c.gy, ch, a.rec);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (wave 0 stored the partial sums)
    }
    if (c.n_tail == 0) {
        // nobody reads the arrivals: the last workgroup of the prelude's readers only has to leave the counters zero
        if (c.n_aux == 0) return;
    }
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(c.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == (int)(c.n_aux + c.n_main) - 1;
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    const char *base = (const char *)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainKernArg, tail));
    for (int s = 0; s < c.n_tail; ++s) {
        const SmallDesc &td = *reinterpret_cast<const SmallDesc *>(base + (size_t)s * sizeof(SmallDesc));
        for (uint32_t vb = 0; vb < c.tail_blocks[s]; ++vb) {
            if (c.tail_block[s])
                chain_tail_step<true>(td, c.tail_mode[s], 8, vb);
            else
                chain_tail_step<false>(td, c.tail_mode[s], c.tail_logG[s], vb);
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (tid == 0) {
        __hip_atomic_store(c.state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c.state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// What plan_chain (normal_lse_chain.hip) hands a launcher.
struct ChainPlan {
    ChainKernArg k;
    X3Prep xp;
    uint32_t grid = 0;
    bool rec = false;
    bool syncfree = false;                // no hand-off inside the launch: no tail, no body small factor from the prelude
};

template <int EQV>
int chain_launch_eq(const ChainPlan &p, hipStream_t stream) {
    auto launch = [&](auto kern) {
        if (p.xp.lds > 64 * 1024 &&
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.xp.lds) != hipSuccess)
            return ALAN_ERR_LAUNCH;
        ALAN_LAUNCH(kern, dim3(p.grid), dim3(256), p.xp.lds, stream, p.k);
        return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
    };
    const bool flat = p.xp.flat;
    return p.xp.nst == 4   ? (flat ? launch(normal_lse_x3_chain_kernel<EQV, 4, 1, true, false>)
                                   : launch(normal_lse_x3_chain_kernel<EQV, 4, 1, false, false>))
           : p.xp.nst == 2 ? (flat ? launch(normal_lse_x3_chain_kernel<EQV, 2, 1, true, false>)
                                   : launch(normal_lse_x3_chain_kernel<EQV, 2, 1, false, false>))
           : p.xp.nlw == 2 ? (p.syncfree ? launch(normal_lse_x3_chain_kernel<EQV, 1, 2, false, true, false>)
                              : p.rec    ? launch(normal_lse_x3_chain_kernel<EQV, 1, 2, false, true>)
                                         : launch(normal_lse_x3_chain_kernel<EQV, 1, 2, false, false>))
                           : (flat ? launch(normal_lse_x3_chain_kernel<EQV, 1, 1, true, false>)
                                   : launch(normal_lse_x3_chain_kernel<EQV, 1, 1, false, false>));
}

int chain_launch_eq4(const ChainPlan &p, hipStream_t stream);
int chain_launch_eq8(const ChainPlan &p, hipStream_t stream);
int chain_launch_eq10(const ChainPlan &p, hipStream_t stream);
int chain_launch_eq12(const ChainPlan &p, hipStream_t stream);
int chain_launch_eq17(const ChainPlan &p, hipStream_t stream);

}  // namespace alan
