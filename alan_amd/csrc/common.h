// Shared host/device helpers for libalan_mi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <memory>
#include <tuple>
#include <vector>
#include "alan_mi355.h"

namespace alan {

// A tuning / ablation knob from the environment, read ONCE per process (a `static const` at the use site): launch
// paths never call getenv again, and a later setenv() cannot change the library's behaviour under a running caller.
constexpr int ENV_UNSET = -2147483647 - 1;
inline int env_knob(const char *name) {
    const char *e = getenv(name);
    return e ? atoi(e) : ENV_UNSET;
}

// Every kernel launch of the library goes through alan_launch: normally straight to hipLaunchKernelGGL (or its event-
// carrying form), but while a call list is being recorded on this thread (calls.hip: alan_calls_begin .. alan_calls_end)
// the launch is NOT issued -- it is kept, kernel and grid and a copy of every argument, to be issued by alan_calls_replay
// (or by the issuing threads of an alan_pipeline): all the host-side planning of a call happens once.  A kept launch goes
// out through hipModuleLaunchKernel on the hipFunction_t looked up when it was recorded: 2.4 us of host time against 3.4
// for hipLaunchKernelGGL, which looks the function up on every call (tools/launch_cost_probe.hip).
struct KeptLaunch {
    hipFunction_t fn = nullptr;
    dim3 grid, block;
    uint32_t lds = 0;
    void **params = nullptr;
    virtual ~KeptLaunch() {}
    inline hipError_t issue(hipStream_t st) const {
        return hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds, st, params, nullptr);
    }
};
template <typename... KArgs>
struct KeptLaunchOf : KeptLaunch {
    std::tuple<std::decay_t<KArgs>...> held;
    void *ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    template <typename... Args>
    explicit KeptLaunchOf(const Args &...a) : held(a...) {
        int i = 0;
        std::apply([&](auto &...x) { ((ptrs[i++] = (void *)&x), ...); }, held);
        params = ptrs;
    }
};
struct LaunchRecorder {
    std::vector<std::unique_ptr<KeptLaunch>> launches;
    bool failed = false;              // a launch that cannot be kept (timing events, no function handle) was made
};
extern thread_local LaunchRecorder *g_launch_recorder;

template <typename... KArgs, typename... Args>
inline void alan_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, hipEvent_t e0,
                        hipEvent_t e1, const Args &...args) {
    if (g_launch_recorder) {
        auto k = std::make_unique<KeptLaunchOf<KArgs...>>(args...);
        if (e0 || e1 || hipGetFuncBySymbol(&k->fn, (const void *)kernel) != hipSuccess) {
            (void)hipGetLastError();
            g_launch_recorder->failed = true;
            return;
        }
        k->grid = grid, k->block = block, k->lds = (uint32_t)lds;
        g_launch_recorder->launches.emplace_back(std::move(k));
        return;
    }
    if (e0 || e1)
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0, e1, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}
#define ALAN_LAUNCH(kernel, grid, block, lds, stream, ...) \
    ::alan::alan_launch(kernel, grid, block, lds, stream, nullptr, nullptr, __VA_ARGS__)
#define ALAN_LAUNCH_EXT(kernel, grid, block, lds, stream, e0, e1, flags, ...) \
    ::alan::alan_launch(kernel, grid, block, lds, stream, e0, e1, __VA_ARGS__)

constexpr int MAXD = ALAN_MAX_DIMS;
constexpr int MAXF = ALAN_MAX_FACTORS;
constexpr int WAVE = 64;  // CDNA4 wavefront

// n / d for n < 2^31, d >= 1, by multiply-high (no hardware integer divide on CDNA).
struct FastDiv {
    uint32_t d, m, l;
};

inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.l = l;
    f.m = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
    return f;
}

__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv &f) {
    return (__umulhi(n, f.m) + n) >> f.l;
}

template <typename T>
struct Num;
template <>
struct Num<float> {
    static constexpr float eps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
    __device__ static __forceinline__ float exp(float x) { return __expf(x); }
    __device__ static __forceinline__ float log(float x) { return logf(x); }
    __device__ static __forceinline__ float exp_acc(float x) { return expf(x); }
    __device__ static __forceinline__ float log1p(float x) { return log1pf(x); }
    __device__ static __forceinline__ float ninf() { return -__builtin_huge_valf(); }
    __device__ static __forceinline__ float nan() { return __builtin_nanf(""); }
};
template <>
struct Num<double> {
    static constexpr double eps = 2.220446049250313e-16;  // torch.finfo(float64).eps
    __device__ static __forceinline__ double exp(double x) { return ::exp(x); }
    __device__ static __forceinline__ double log(double x) { return ::log(x); }
    __device__ static __forceinline__ double exp_acc(double x) { return ::exp(x); }
    __device__ static __forceinline__ double log1p(double x) { return ::log1p(x); }
    __device__ static __forceinline__ double ninf() { return -__builtin_huge_val(); }
    __device__ static __forceinline__ double nan() { return __builtin_nan(""); }
};

template <typename T>
__device__ __forceinline__ T load_as(const void *p, int dtype, int64_t off) {
    return dtype == ALAN_F32 ? (T)((const float *)p)[off] : (T)((const double *)p)[off];
}

template <typename T>
__device__ __forceinline__ void store_as(void *p, int dtype, int64_t off, T v) {
    if (dtype == ALAN_F32)
        ((float *)p)[off] = (float)v;
    else
        ((double *)p)[off] = (double)v;
}

// One-exp online log-sum-exp update of the running (max m, scaled sum s) with a new value x.
// -inf contributes nothing; NaN poisons s (as torch's amax/exp/sum would).
template <typename T>
__device__ __forceinline__ void lse_push(T &m, T &s, T x) {
    if (!(x == Num<T>::ninf())) {
        T d = x - m;                         // +inf on the first finite value (m = -inf)
        T e = Num<T>::exp(d > 0 ? -d : d);   // exp(-|d|)
        s = d > 0 ? s * e + T(1) : s + e;
        m = d > 0 ? x : m;
    }
}

// Merge another lane's (m, s) into ours.
template <typename T>
__device__ __forceinline__ void lse_merge(T &m, T &s, T m2, T s2) {
    T mm = m > m2 ? m : m2;
    // (a NaN sum survives the multiply by 0, as it must)
    T a = s * ((m == Num<T>::ninf()) ? T(0) : Num<T>::exp(m - mm));
    T b = s2 * ((m2 == Num<T>::ninf()) ? T(0) : Num<T>::exp(m2 - mm));
    s = a + b;
    m = mm;
}

// log(s + eps) + m with the reference's corner cases: an all -inf slice gives NaN (x - max = NaN in
// utils.py:219), and so does a +inf max.
template <typename T>
__device__ __forceinline__ T lse_finish(T m, T s) {
    if (m == Num<T>::ninf() || m == -Num<T>::ninf()) return Num<T>::nan();
    return Num<T>::log(s + Num<T>::eps) + m;
}

}  // namespace alan
