// A recorded sequence of this library's launches, issued again by ONE call: what a caller that evaluates the same
// contraction over and over (Sample.elbo_nograd on a fixed sample: the reference's basic_runner loop, logpq.py:68-155 per
// call) would otherwise replay as a captured HIP graph -- whose launch leaves the GPU idle for several microseconds
// (tools/replay_trace.sh; tools/direct_replay_probe.py: 27.2 us per evaluation as a graph, 23.2 us with the three library
// calls issued again from the host).  The list holds copies of the descriptors, so the device pointers in them must stay
// valid (the caller keeps the tensors -- e.g. the private pool of the graph it captured the evaluation into).
#include <hip/hip_runtime.h>
#include <string.h>
#include <new>
#include "common.h"

namespace alan {

thread_local LaunchRecorder *g_launch_recorder = nullptr;

// The launches of the recorded calls, each with its grid and a copy of its arguments (common.h: alan_launch): a replay
// plans nothing, it launches.
struct Calls {
    LaunchRecorder rec;
    int n_calls = 0;
};

template <typename F>
static int record(void *calls, F call) {
    Calls *c = (Calls *)calls;
    if (g_launch_recorder) return ALAN_ERR_BAD_DESC;          // (not while another list is being recorded)
    const size_t before = c->rec.launches.size();
    g_launch_recorder = &c->rec;
    const int rc = call();
    g_launch_recorder = nullptr;
    if (rc != ALAN_OK)
        c->rec.launches.resize(before);
    else
        ++c->n_calls;
    return rc;
}

// {counter, seed} of a generator slot copied to another one (alan_noise_t.cell): what closes the ring of a captured graph
// that holds a single launch with generated noise -- that launch may not hand its state on to its own slot.
__global__ void noise_handon_kernel(const unsigned long long *from, unsigned long long *to) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long c = __hip_atomic_load(from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long s = __hip_atomic_load(from + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(to, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(to + 1, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace alan

using namespace alan;

extern "C" {

int alan_calls_create(void **calls) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    *calls = new (std::nothrow) Calls();
    return *calls ? ALAN_OK : ALAN_ERR_WORKSPACE;
}

int alan_calls_add_reduce(void *calls, const alan_reduce_desc_t *desc, void *workspace, size_t workspace_bytes) {
    if (!calls || !desc) return ALAN_ERR_BAD_DESC;
    if (desc->ev_start || desc->ev_stop) return ALAN_ERR_UNSUPPORTED;         // (events belong to one launch)
    return record(calls, [&] { return alan_reduce(desc, workspace, workspace_bytes, nullptr); });
}

int alan_calls_add_reduce_batch(void *calls, const alan_reduce_desc_t *const *descs, int32_t n) {
    if (!calls || !descs || n < 0 || n > 64) return ALAN_ERR_BAD_DESC;
    for (int i = 0; i < n; ++i)
        if (!descs[i] || descs[i]->ev_start || descs[i]->ev_stop) return ALAN_ERR_BAD_DESC;
    return record(calls, [&] { return alan_reduce_batch(descs, n, nullptr); });
}

int alan_calls_add_normal_lse(void *calls, const alan_normal_lse_desc_t *desc, void *workspace, size_t workspace_bytes) {
    if (!calls || !desc) return ALAN_ERR_BAD_DESC;
    if (desc->ev_start || desc->ev_stop) return ALAN_ERR_UNSUPPORTED;
    return record(calls, [&] { return alan_normal_lse(desc, workspace, workspace_bytes, nullptr); });
}

int alan_calls_add_chain_terms_final(void *calls, const void *const *terms, const int64_t *strides, int32_t n_terms,
                                     const alan_chain_normal_t *normal, const alan_chain_final_t *fin, int32_t dtype,
                                     int64_t B, int64_t T, int64_t K, void *out_vec, void *workspace,
                                     size_t workspace_bytes) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    // (no out_chain: delivering it is a copy, not a launch -- a list holds launches only)
    return record(calls, [&] {
        return alan_chain_logmmexp_terms_final(terms, strides, n_terms, normal, fin, dtype, B, T, K, nullptr, out_vec,
                                               workspace, workspace_bytes, nullptr);
    });
}

int alan_calls_add_normal_lse_chained(void *calls, const alan_normal_lse_desc_t *desc,
                                      const alan_reduce_desc_t *const *prelude, int32_t n_prelude,
                                      const alan_reduce_desc_t *const *tail, int32_t n_tail, void *state) {
    if (!calls || !desc) return ALAN_ERR_BAD_DESC;
    if (desc->ev_start || desc->ev_stop) return ALAN_ERR_UNSUPPORTED;
    return record(calls, [&] { return alan_normal_lse_chained(desc, prelude, n_prelude, tail, n_tail, state, nullptr); });
}

int alan_calls_add_exchange_sum(void *calls, void *exchange, const void *src, void *out, int64_t n) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    return record(calls, [&] { return alan_exchange_sum(exchange, src, out, n, nullptr); });
}

int alan_noise_handon(const void *from, void *to, void *stream) {
    if (!from || !to || from == to) return ALAN_ERR_BAD_DESC;
    ALAN_LAUNCH(noise_handon_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long *)from,
                (unsigned long long *)to);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_calls_add_noise_handon(void *calls, const void *from, void *to) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    return record(calls, [&] { return alan_noise_handon(from, to, nullptr); });
}

int alan_calls_replay(void *calls, void *stream) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    for (const auto &launch : ((Calls *)calls)->rec.launches) launch((hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_calls_destroy(void *calls) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    delete (Calls *)calls;
    return ALAN_OK;
}

}  // extern "C"
