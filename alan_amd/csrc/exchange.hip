// One-shot sum of the ranks' partial log-marginals over xGMI (the collective of a sharded Split, logpq.py:149-153
// across ranks): every rank WRITES its [K_parents...] partial (40 KB at K = 100) into a slot of every peer's inbox,
// raises a flag there, waits for the peers' flags in its own inbox and sums the slots in rank order -- one launch, no
// ring, no second hop, the same bits on every rank.  The inboxes are device memory of this library (the one exception to
// "the library never allocates": an inbox has to be a whole allocation to be exported to the peers' processes through a
// HIP IPC handle).
//
// Protocol (per rank; s = the exchange's running number, kept on the device so that a replayed HIP graph advances it):
//   inbox   float [2][world][capacity]      slot [s & 1][q] is written by rank q, and only by rank q
//   flags   u32   [world]                   flags[q] = the number of the last exchange rank q has delivered here
//   state   u32   [4]                       [0] exchanges completed here, [1] ticket of the running launch, [2] error
// A launch has `world` workgroups.  Workgroup q: (1) copies this rank's partial into peer q's slot [p][rank] with
// system-scope stores, fences, and stores s into peer q's flags[rank] (release, system scope); (2) waits -- for a BOUNDED
// time -- until every flags[.] here has reached s (acquire, system scope); (3) adds slice q of the world slots, read with
// system-scope loads, in rank order into `out`.  The last workgroup to finish publishes state[0] = s.
// Why two slots are enough: a rank finishes exchange s only after every peer has delivered s, and a peer delivers s + 1
// only after its own launch of s has finished and its consumer has been enqueued behind it; so while this rank still
// reads slot [p] of exchange s, a peer can be at most writing slot [p ^ 1] of s + 1.
// A wait that runs out (a peer died, or never launched) sets state[2], fills this rank's result with NaN and ends the
// launch: nothing spins for ever.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace alan {

constexpr int EX_MAX = ALAN_EXCHANGE_MAX_RANKS;
constexpr int EX_THREADS = 1024;

struct ExDesc {
    float* inbox[EX_MAX];        // every rank's inbox as mapped into THIS process ([rank] = the local one)
    unsigned* flags[EX_MAX];
    unsigned* state;             // local
    const float* src;
    float* out;
    int world, rank, n, capacity;
    long long spin_ticks;        // wall_clock64 ticks (100 MHz) a wait may last
};

__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(EX_THREADS) void exchange_sum_kernel(const ExDesc d) {
    __shared__ unsigned s_step, s_bad;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        s_step = __hip_atomic_load(d.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        // a rank that has timed out once stays failed: its peers may have completed that exchange with a valid sum, so its
        // later results are NaN too (it still delivers and waits: the peers' protocol goes on) until the caller resets
        s_bad = __hip_atomic_load(d.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    }
    __syncthreads();
    const unsigned step = s_step;
    const int p = step & 1u;
    // (1) deliver to peer q
    float* dst = d.inbox[q] + ((size_t)p * d.world + d.rank) * d.capacity;
    for (int i = tid; i < d.n; i += EX_THREADS) st_sys(dst + i, d.src[i]);
    __atomic_thread_fence(__ATOMIC_RELEASE);      // (system scope: the default of the builtin)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(d.flags[q] + d.rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (2) wait for every peer's delivery here
    if (tid < d.world) {
        const unsigned* f = d.flags[d.rank] + tid;
        const long long t0 = wall_clock64();
        bool ok = false;
        for (;;) {
            unsigned v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - step) >= 0) { ok = true; break; }
            if (wall_clock64() - t0 > d.spin_ticks) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (!ok) atomicOr(&s_bad, 1u);
    }
    __syncthreads();
    const bool bad = s_bad != 0;
    // (3) slice q of the sum, in rank order
    const int per = (d.n + d.world - 1) / d.world, lo = q * per, hi = min(d.n, lo + per);
    const float* mine = d.inbox[d.rank] + (size_t)p * d.world * d.capacity;
    for (int i = lo + tid; i < hi; i += EX_THREADS) {
        float v[EX_MAX];
#pragma unroll
        for (int r = 0; r < EX_MAX; ++r) v[r] = r < d.world ? ld_sys(mine + (size_t)r * d.capacity + i) : 0.f;
        float s = v[0];
#pragma unroll
        for (int r = 1; r < EX_MAX; ++r) if (r < d.world) s += v[r];
        d.out[i] = bad ? __builtin_nanf("") : s;
    }
    __syncthreads();
    if (tid == 0) {
        if (bad) {                                // (the FIRST exchange that failed; later ones leave it)
            unsigned expect = 0u;
            __hip_atomic_compare_exchange_strong(d.state + 2, &expect, step, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
        unsigned ticket = __hip_atomic_fetch_add(d.state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned)d.world - 1u) {   // every workgroup has read state[0] and finished: publish the number
            __hip_atomic_store(d.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.state, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct Exchange {
    int world, rank, capacity;
    void* base;                    // the local allocation: inbox, then flags, then state
    void* peer_base[EX_MAX];       // the peers' allocations as opened here (null for the local one / before connect)
    bool connected;
    long long spin_ticks;
};

static size_t inbox_bytes(int world, int capacity) { return (size_t)2 * world * capacity * sizeof(float); }
static size_t region_bytes(int world, int capacity) { return inbox_bytes(world, capacity) + 256; }

}  // namespace alan

using namespace alan;

extern "C" {

int alan_exchange_create(int32_t world, int32_t rank, int32_t capacity, unsigned char* handle_out, void** exchange) {
    if (!handle_out || !exchange || world < 1 || world > EX_MAX || rank < 0 || rank >= world || capacity < 1 ||
        capacity > (1 << 24) || sizeof(hipIpcMemHandle_t) != ALAN_EXCHANGE_HANDLE_BYTES)
        return ALAN_ERR_BAD_DESC;
    Exchange* e = (Exchange*)calloc(1, sizeof(Exchange));
    if (!e) return ALAN_ERR_WORKSPACE;
    e->world = world, e->rank = rank, e->capacity = capacity;
    const size_t bytes = region_bytes(world, capacity);
    // uncached device memory (what RCCL gives its own xGMI buffers): peers' stores and this rank's flag polls go to the
    // memory itself, not to a line held by some XCD's L2.  ALAN_EXCHANGE_ALLOC=plain asks for ordinary hipMalloc memory
    const char* how = getenv("ALAN_EXCHANGE_ALLOC");
    hipError_t rc = (how && !strcmp(how, "plain")) ? hipMalloc(&e->base, bytes)
                                                   : hipExtMallocWithFlags(&e->base, bytes, hipDeviceMallocUncached);
    if (rc != hipSuccess) { (void)hipGetLastError(); free(e); return ALAN_ERR_WORKSPACE; }
    if (hipMemset(e->base, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, e->base) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(e->base); free(e); return ALAN_ERR_LAUNCH;
    }
    const char* ms = getenv("ALAN_EXCHANGE_SPIN_MS");
    long long spin_ms = ms ? atoll(ms) : 2000;
    if (spin_ms < 1) spin_ms = 1;
    if (spin_ms > 60000) spin_ms = 60000;
    e->spin_ticks = spin_ms * 100000ll;           // wall_clock64: 100 MHz
    *exchange = e;
    return ALAN_OK;
}

int alan_exchange_connect(void* exchange, const unsigned char* handles) {
    Exchange* e = (Exchange*)exchange;
    if (!e || !handles || e->connected) return ALAN_ERR_BAD_DESC;
    for (int q = 0; q < e->world; ++q) {
        if (q == e->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)q * ALAN_EXCHANGE_HANDLE_BYTES, sizeof(h));
        if (hipIpcOpenMemHandle(&e->peer_base[q], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            for (int r = 0; r < q; ++r)
                if (e->peer_base[r]) { (void)hipIpcCloseMemHandle(e->peer_base[r]); e->peer_base[r] = nullptr; }
            return ALAN_ERR_LAUNCH;
        }
    }
    e->connected = true;
    return ALAN_OK;
}

int alan_exchange_sum(void* exchange, const void* src, void* out, int64_t n, void* stream) {
    Exchange* e = (Exchange*)exchange;
    if (!e || !e->connected || !src || !out || n < 1) return ALAN_ERR_BAD_DESC;
    if (n > e->capacity) return ALAN_ERR_UNSUPPORTED;
    ExDesc d;
    memset(&d, 0, sizeof(d));
    const size_t ib = inbox_bytes(e->world, e->capacity);
    for (int q = 0; q < e->world; ++q) {
        char* b = (char*)(q == e->rank ? e->base : e->peer_base[q]);
        d.inbox[q] = (float*)b;
        d.flags[q] = (unsigned*)(b + ib);
    }
    d.state = (unsigned*)((char*)e->base + ib + 128);
    d.src = (const float*)src, d.out = (float*)out;
    d.world = e->world, d.rank = e->rank, d.n = (int)n, d.capacity = e->capacity, d.spin_ticks = e->spin_ticks;
    ALAN_LAUNCH(exchange_sum_kernel, dim3(e->world), dim3(EX_THREADS), 0, (hipStream_t)stream, d);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_exchange_status(void* exchange, uint32_t* completed, uint32_t* failed_at) {
    Exchange* e = (Exchange*)exchange;
    if (!e) return ALAN_ERR_BAD_DESC;
    unsigned st[4];
    const size_t ib = inbox_bytes(e->world, e->capacity);
    if (hipMemcpy(st, (char*)e->base + ib + 128, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return ALAN_ERR_LAUNCH;
    }
    if (completed) *completed = st[0];
    if (failed_at) *failed_at = st[2];
    return ALAN_OK;
}

int alan_exchange_destroy(void* exchange) {
    Exchange* e = (Exchange*)exchange;
    if (!e) return ALAN_ERR_BAD_DESC;
    for (int q = 0; q < e->world; ++q)
        if (e->peer_base[q]) (void)hipIpcCloseMemHandle(e->peer_base[q]);
    (void)hipFree(e->base);
    (void)hipGetLastError();
    free(e);
    return ALAN_OK;
}

}  // extern "C"
