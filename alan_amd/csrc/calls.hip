// Recorded launch lists and the evaluation pipeline.
//
// alan_calls_*: a recorded sequence of this library's launches, issued again by ONE call: what a caller that evaluates the
// same contraction over and over (Sample.elbo_nograd on a fixed sample: the reference's basic_runner loop, logpq.py:68-155
// per call) would otherwise replay as a captured HIP graph -- whose launch leaves the GPU idle for several microseconds
// (tools/replay_trace.sh; tools/direct_replay_probe.py: 27.2 us per evaluation as a graph, 23.2 us with the three library
// calls issued again from the host).  The list holds copies of the kernel arguments, so the device pointers in them must
// stay valid (the caller keeps the tensors -- e.g. the private pool of the graph it captured the evaluation into).
//
// alan_pipeline_*: INDEPENDENT evaluations overlapped (round 4).  Consecutive ELBO evaluations of the reference's loop
// (basic_runner.py:81-112) do not depend on each other, and one evaluation is a chain of dependent launches most of which
// fill a fraction of the chip (movielens K=30: producers on a few workgroups, the fused plate step, a one-workgroup
// log-sum-exp -- 23 us end to end, three launch boundaries).  A pipeline holds n copies of the evaluation ("lanes": each
// a launch list recorded over intermediates of its own) and issues evaluation i on lane i % n, each lane on a stream of
// its own, so that evaluation i + 1's producers and evaluation i - 1's last log-sum-exp run beside evaluation i's plate
// step.  With the chip no longer waiting, the HOST's launch calls are the bound (3 x 2.4 us per evaluation from one
// thread), so the lanes are issued by threads of the library's own, one per lane by default
// (tools/launch_cost_probe.hip: four threads issuing to four streams launch 3.3 x as fast as one).
#include <hip/hip_runtime.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include "common.h"

namespace alan {

thread_local LaunchRecorder *g_launch_recorder = nullptr;

struct Calls {
    LaunchRecorder rec;
};

static inline int issue_all(const Calls *c, hipStream_t st) {
    int bad = 0;
    for (const auto &k : c->rec.launches) bad |= k->issue(st) != hipSuccess;
    return bad;
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

// ---- the pipeline --------------------------------------------------------------------------------------------------
struct Lane {
    Calls *calls = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    std::atomic<int64_t> submitted{0}, issued{0};
    int64_t fenced = 0;                       // the fence this lane's stream has been ordered behind (its issuing thread only)
};

struct Pipeline {
    int device = 0, n_lanes = 0, n_threads = 0;
    Lane *lanes = nullptr;
    std::thread *workers = nullptr;
    hipEvent_t fence_ev = nullptr;
    std::atomic<int64_t> fence_seq{0};        // fences recorded so far: a lane's issuing thread orders its stream behind the latest
    int64_t total = 0;                        // evaluations submitted so far (caller's thread only)
    std::atomic<bool> stop{false};
    std::atomic<int> err{0}, sleeping{0};
    std::mutex mu;
    std::condition_variable cv;
};

// An issuing thread: lanes w, w + T, ...  It spins for a few milliseconds after its last launch (a submit that follows soon
// -- a loop of evaluations with some host work between the batches -- finds it awake: a thread woken from the condition
// variable starts ~100 us late, a third of a 20-evaluation batch), then sleeps: an idle pipeline costs no CPU.
static void worker(Pipeline *p, int w) {
    (void)hipSetDevice(p->device);
    auto last = std::chrono::steady_clock::now();
    while (!p->stop.load(std::memory_order_acquire)) {
        bool did = false;
        for (int l = w; l < p->n_lanes; l += p->n_threads) {
            Lane &ln = p->lanes[l];
            while (ln.issued.load(std::memory_order_relaxed) < ln.submitted.load(std::memory_order_acquire)) {
                // (the fence's stream wait, by the lane's own thread: four of them one after the other were 15 us of the
                // caller's time in front of every batch)
                const int64_t fs = p->fence_seq.load(std::memory_order_acquire);
                if (ln.fenced != fs) {
                    if (hipStreamWaitEvent(ln.stream, p->fence_ev, 0) != hipSuccess) p->err.store(1);
                    ln.fenced = fs;
                }
                if (issue_all(ln.calls, ln.stream)) p->err.store(1);
                ln.issued.fetch_add(1, std::memory_order_release);
                did = true;
                if (p->n_lanes > p->n_threads) break;             // (several lanes per thread: take turns)
            }
        }
        if (did) {
            last = std::chrono::steady_clock::now();
            continue;
        }
        if (std::chrono::steady_clock::now() - last < std::chrono::microseconds(3000)) {
            __builtin_ia32_pause();
            continue;
        }
        std::unique_lock<std::mutex> lk(p->mu);
        p->sleeping.fetch_add(1);
        p->cv.wait_for(lk, std::chrono::milliseconds(50), [&] {
            if (p->stop.load()) return true;
            for (int l = w; l < p->n_lanes; l += p->n_threads)
                if (p->lanes[l].issued.load() < p->lanes[l].submitted.load()) return true;
            return false;
        });
        p->sleeping.fetch_sub(1);
        last = std::chrono::steady_clock::now();
    }
}

static void wait_issued(Pipeline *p) {
    for (int l = 0; l < p->n_lanes; ++l)
        while (p->lanes[l].issued.load(std::memory_order_acquire) < p->lanes[l].submitted.load(std::memory_order_relaxed))
            __builtin_ia32_pause();
}

}  // namespace alan

using namespace alan;

extern "C" {

int alan_calls_create(void **calls) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    *calls = new (std::nothrow) Calls();
    return *calls ? ALAN_OK : ALAN_ERR_WORKSPACE;
}

int alan_calls_begin(void *calls) {
    if (!calls || g_launch_recorder) return ALAN_ERR_BAD_DESC;            // (not while another list is being recorded)
    g_launch_recorder = &((Calls *)calls)->rec;
    return ALAN_OK;
}

int alan_calls_end(void *calls) {
    if (!calls || g_launch_recorder != &((Calls *)calls)->rec) return ALAN_ERR_BAD_DESC;
    g_launch_recorder = nullptr;
    return ((Calls *)calls)->rec.failed ? ALAN_ERR_UNSUPPORTED : ALAN_OK;
}

int64_t alan_calls_count(void *calls) { return calls ? (int64_t)((Calls *)calls)->rec.launches.size() : -1; }

int alan_noise_handon(const void *from, void *to, void *stream) {
    if (!from || !to || from == to) return ALAN_ERR_BAD_DESC;
    ALAN_LAUNCH(noise_handon_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long *)from,
                (unsigned long long *)to);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_calls_replay(void *calls, void *stream) {
    if (!calls || ((Calls *)calls)->rec.failed) return ALAN_ERR_BAD_DESC;
    if (issue_all((Calls *)calls, (hipStream_t)stream)) {
        (void)hipGetLastError();
        return ALAN_ERR_LAUNCH;
    }
    return ALAN_OK;
}

int alan_calls_destroy(void *calls) {
    if (!calls) return ALAN_ERR_BAD_DESC;
    if (g_launch_recorder == &((Calls *)calls)->rec) g_launch_recorder = nullptr;
    delete (Calls *)calls;
    return ALAN_OK;
}

int alan_pipeline_create(void *const *calls, int32_t n_lanes, int32_t n_threads, void **pipeline) {
    if (!calls || !pipeline || n_lanes < 1 || n_lanes > ALAN_PIPELINE_MAX_LANES || n_threads < 0) return ALAN_ERR_BAD_DESC;
    for (int l = 0; l < n_lanes; ++l)
        if (!calls[l] || ((Calls *)calls[l])->rec.failed || ((Calls *)calls[l])->rec.launches.empty()) return ALAN_ERR_BAD_DESC;
    Pipeline *p = new (std::nothrow) Pipeline();
    if (!p) return ALAN_ERR_WORKSPACE;
    p->n_lanes = n_lanes, p->n_threads = n_threads > n_lanes ? n_lanes : n_threads;
    p->lanes = new (std::nothrow) Lane[n_lanes];
    bool ok = p->lanes && hipGetDevice(&p->device) == hipSuccess &&
              hipEventCreateWithFlags(&p->fence_ev, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < n_lanes; ++l) {
        p->lanes[l].calls = (Calls *)calls[l];
        ok = hipStreamCreateWithFlags(&p->lanes[l].stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&p->lanes[l].done, hipEventDisableTiming) == hipSuccess;
    }
    if (ok && p->n_threads > 0) {
        p->workers = new (std::nothrow) std::thread[p->n_threads];
        ok = p->workers != nullptr;
        for (int w = 0; ok && w < p->n_threads; ++w) p->workers[w] = std::thread(worker, p, w);
    }
    if (!ok) {
        (void)hipGetLastError();
        alan_pipeline_destroy(p);
        return ALAN_ERR_LAUNCH;
    }
    *pipeline = p;
    return ALAN_OK;
}

int alan_pipeline_submit(void *pipeline, int64_t count) {
    Pipeline *p = (Pipeline *)pipeline;
    if (!p || count < 0) return ALAN_ERR_BAD_DESC;
    const int64_t t0 = p->total, t1 = t0 + count;
    p->total = t1;
    if (p->n_threads == 0) {                          // the caller's thread issues, the lanes in turn
        for (int64_t i = t0; i < t1; ++i) {
            Lane &ln = p->lanes[i % p->n_lanes];
            if (issue_all(ln.calls, ln.stream)) p->err.store(1);
            ln.submitted.fetch_add(1), ln.issued.fetch_add(1);
        }
        return p->err.load() ? ALAN_ERR_LAUNCH : ALAN_OK;
    }
    for (int l = 0; l < p->n_lanes; ++l) {
        // evaluations i in [t0, t1) with i % n == l
        const int64_t n = p->n_lanes;
        const int64_t upto = [&](int64_t t) { return t <= l ? 0 : (t - l + n - 1) / n; }(t1) -
                             [&](int64_t t) { return t <= l ? 0 : (t - l + n - 1) / n; }(t0);
        if (upto) p->lanes[l].submitted.fetch_add(upto, std::memory_order_release);
    }
    if (p->sleeping.load()) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cv.notify_all();
    }
    return ALAN_OK;
}

int alan_pipeline_join(void *pipeline, void *stream) {
    Pipeline *p = (Pipeline *)pipeline;
    if (!p) return ALAN_ERR_BAD_DESC;
    wait_issued(p);
    bool ok = true;
    for (int l = 0; l < p->n_lanes; ++l)
        ok = ok && hipEventRecord(p->lanes[l].done, p->lanes[l].stream) == hipSuccess &&
             hipStreamWaitEvent((hipStream_t)stream, p->lanes[l].done, 0) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return (ok && !p->err.load()) ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_pipeline_fence(void *pipeline, void *stream) {
    Pipeline *p = (Pipeline *)pipeline;
    if (!p) return ALAN_ERR_BAD_DESC;
    wait_issued(p);                           // (nobody is between reading fence_seq and waiting for the event)
    bool ok = hipEventRecord(p->fence_ev, (hipStream_t)stream) == hipSuccess;
    if (p->n_threads == 0) {
        for (int l = 0; ok && l < p->n_lanes; ++l) ok = hipStreamWaitEvent(p->lanes[l].stream, p->fence_ev, 0) == hipSuccess;
    } else {
        p->fence_seq.fetch_add(1, std::memory_order_release);     // the lanes' threads wait for it before their next launch
    }
    if (!ok) (void)hipGetLastError();
    return ok ? ALAN_OK : ALAN_ERR_LAUNCH;
}

int alan_pipeline_destroy(void *pipeline) {
    Pipeline *p = (Pipeline *)pipeline;
    if (!p) return ALAN_ERR_BAD_DESC;
    p->stop.store(true);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cv.notify_all();
    }
    if (p->workers) {
        for (int w = 0; w < p->n_threads; ++w)
            if (p->workers[w].joinable()) p->workers[w].join();
        delete[] p->workers;
    }
    if (p->lanes) {
        for (int l = 0; l < p->n_lanes; ++l) {
            if (p->lanes[l].stream) {
                (void)hipStreamSynchronize(p->lanes[l].stream);
                (void)hipStreamDestroy(p->lanes[l].stream);
            }
            if (p->lanes[l].done) (void)hipEventDestroy(p->lanes[l].done);
        }
        delete[] p->lanes;
    }
    if (p->fence_ev) (void)hipEventDestroy(p->fence_ev);
    (void)hipGetLastError();
    delete p;
    return ALAN_OK;
}

}  // extern "C"
