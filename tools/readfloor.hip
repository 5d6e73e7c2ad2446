// Floor probe: how fast can ANY kernel read (or write) N bytes once on this chip, at the literal movielens size
// (32.5 MB, cache-resident, ~1 residency wave) and in the bandwidth regime (2 GB)?  Sets the yardstick for
// rows_kernel's achieved GB/s.  Build: hipcc -O3 --offload-arch=gfx950 tools/readfloor.hip -o tools/_build/readfloor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNR>
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *__restrict__ p, size_t n4, float *out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) {
        const f32x4 v = __builtin_nontemporal_load(p + i);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5678f) out[blockIdx.x] = acc;
}

__global__ void fill_kernel(f32x4 *p, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

__global__ void empty_kernel() {}

#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

template <typename L>
static float time_us(L launch, int iters, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(nullptr, nullptr);
    CK(hipStreamSynchronize(s));
    float tot = 0.f;
    for (int i = 0; i < iters; ++i) {
        launch(a, b);
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    return tot / iters * 1e3f;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t sizes[] = {(size_t)32500000, (size_t)152000000, (size_t)2080000000};
    float *out;
    CK(hipMalloc(&out, 1 << 20));
    {
        const float us = time_us([&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, a, b, 0); }, 50, s);
        printf("empty kernel (kernel-exact events): %.2f us\n", us);
    }
    for (size_t bytes : sizes) {
        const size_t n4 = bytes / 16;
        f32x4 *p;
        CK(hipMalloc(&p, n4 * 16));
        fill_kernel<<<4096, 256, 0, s>>>(p, n4);
        CK(hipStreamSynchronize(s));
        for (int blocks : {256, 512, 1024, 1280, 2048, 4096, 8192, 16384}) {
            const float u1 = time_us([&](hipEvent_t a, hipEvent_t b) {
                hipExtLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, s, a, b, 0, p, n4, out); }, 30, s);
            const float u4 = time_us([&](hipEvent_t a, hipEvent_t b) {
                hipExtLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, s, a, b, 0, p, n4, out); }, 30, s);
            const float u8 = time_us([&](hipEvent_t a, hipEvent_t b) {
                hipExtLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(256), 0, s, a, b, 0, p, n4, out); }, 30, s);
            printf("%8.1f MB blocks=%5d  unr1 %8.2f us %7.0f GB/s | unr4 %8.2f us %7.0f GB/s | unr8 %8.2f us %7.0f GB/s\n",
                   bytes / 1e6, blocks, u1, bytes / u1 / 1e3, u4, bytes / u4 / 1e3, u8, bytes / u8 / 1e3);
            fflush(stdout);
        }
        // the same read inside a replayed HIP graph, 20 launches back to back (how the kernels of a captured ELBO
        // evaluation run: no idle gap, no per-launch host work): time per launch = replay time / 20
        if (bytes < 100000000) {
            for (int blocks : {600, 1200, 2048}) {
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, s, p, n4, out);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                hipEvent_t a, b2;
                CK(hipEventCreate(&a));
                CK(hipEventCreate(&b2));
                CK(hipEventRecord(a, s));
                for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(b2, s));
                CK(hipEventSynchronize(b2));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b2));
                printf("%8.1f MB read inside a replayed graph, blocks=%5d unr4: %8.2f us per launch %7.0f GB/s\n", bytes / 1e6,
                       blocks, ms * 1e3f / 200, bytes / (ms * 1e3f / 200) / 1e3);
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
            }
        }
        // write floor: a plain 16-byte-per-lane fill of the same buffer (what the factor PRODUCER is bounded by)
        for (int blocks : {1024, 2048, 4096, 8192}) {
            const float uw = time_us([&](hipEvent_t a, hipEvent_t b) {
                hipExtLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, a, b, 0, p, n4); }, 30, s);
            printf("%8.1f MB FILL blocks=%5d %8.2f us %7.0f GB/s\n", bytes / 1e6, blocks, uw, bytes / uw / 1e3);
        }
        // cold-ish: a fill (write) of the same buffer between reads, as the producer does in the ELBO
        {
            float tot = 0.f;
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            for (int i = 0; i < 20; ++i) {
                fill_kernel<<<4096, 256, 0, s>>>(p, n4);
                hipExtLaunchKernelGGL(read_kernel<4>, dim3(2048), dim3(256), 0, s, a, b, 0, p, n4, out);
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                tot += ms;
            }
            printf("%8.1f MB after a write of the buffer, blocks=2048 unr4: %8.2f us %7.0f GB/s\n", bytes / 1e6,
                   tot / 20 * 1e3f, bytes / (tot / 20 * 1e3f) / 1e3);
        }
        CK(hipFree(p));
    }
    return 0;
}
