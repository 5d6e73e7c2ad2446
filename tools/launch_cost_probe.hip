// What does the HOST pay per kernel launch on this stack, by entry point and by number of issuing threads?  (Round 4: with
// independent evaluations issued on several streams the chip is no longer the bound at K = 30 -- the host's three launches
// per evaluation are: tools/overlap_probe.py, 10.6 us of host time per evaluation.)
//   hipcc --offload-arch=gfx950 -O3 -pthread tools/launch_cost_probe.hip -o tools/_build/launch_cost_probe
// Prints host microseconds per launch for: hipLaunchKernelGGL, hipModuleLaunchKernel on a hipFunction_t looked up once
// (hipGetFuncBySymbol), the same with the arguments packed into one buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER), a captured
// graph of three kernels per hipGraphLaunch -- each from 1, 2 and 4 host threads issuing to streams of their own.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

struct Arg {
    float *p;
    int n;
    char pad[200];                  // (the library's descriptors are a few hundred bytes of kernel argument)
};

__global__ void tiny(const Arg a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += 1.f;
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));              \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

int main() {
    const int N = 20000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int nthreads : {1, 2, 4}) {
            std::vector<hipStream_t> streams(nthreads);
            std::vector<float *> bufs(nthreads);
            for (int i = 0; i < nthreads; ++i) {
                CK(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
                CK(hipMalloc(&bufs[i], 256));
                CK(hipMemset(bufs[i], 0, 256));
            }
            hipFunction_t fn;
            CK(hipGetFuncBySymbol(&fn, (const void *)tiny));
            std::vector<hipGraphExec_t> execs(nthreads);
            if (mode == 3) {
                for (int i = 0; i < nthreads; ++i) {
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(streams[i], hipStreamCaptureModeThreadLocal));
                    Arg a;
                    memset(&a, 0, sizeof(a));
                    a.p = bufs[i], a.n = 1;
                    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, streams[i], a);
                    CK(hipStreamEndCapture(streams[i], &g));
                    CK(hipGraphInstantiate(&execs[i], g, nullptr, nullptr, 0));
                }
            }
            CK(hipDeviceSynchronize());
            std::vector<double> host(nthreads);
            auto work = [&](int i) {
                Arg a;
                memset(&a, 0, sizeof(a));
                a.p = bufs[i], a.n = 1;
                void *params[1] = {&a};
                size_t sz = sizeof(a);
                void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
                const int per = N / nthreads;
                const double t0 = now();
                for (int k = 0; k < per; ++k) {
                    if (mode == 0)
                        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, streams[i], a);
                    else if (mode == 1)
                        hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, streams[i], params, nullptr);
                    else if (mode == 2)
                        hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, streams[i], nullptr, extra);
                    else if (k % 3 == 0)
                        hipGraphLaunch(execs[i], streams[i]);
                }
                host[i] = now() - t0;
            };
            const double t0 = now();
            std::vector<std::thread> th;
            for (int i = 1; i < nthreads; ++i) th.emplace_back(work, i);
            work(0);
            for (auto &x : th) x.join();
            const double t1 = now();
            CK(hipDeviceSynchronize());
            const double t2 = now();
            float total = 0.f;
            for (int i = 0; i < nthreads; ++i) {
                float v;
                CK(hipMemcpy(&v, bufs[i], 4, hipMemcpyDeviceToHost));
                total += v;
            }
            const char *names[] = {"hipLaunchKernelGGL", "hipModuleLaunchKernel(params)", "hipModuleLaunchKernel(packed buffer)",
                                   "hipGraphLaunch of 3 kernels (per kernel)"};
            printf("%-42s %d thread(s): host %.2f us per launch per thread, %.2f us per launch overall; all done after %.2f us per launch  (ran %.0f)\n",
                   names[mode], nthreads, host[0] / (N / nthreads) * 1e6, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, total);
            for (int i = 0; i < nthreads; ++i) {
                CK(hipStreamDestroy(streams[i]));
                CK(hipFree(bufs[i]));
            }
        }
    }
    return 0;
}
