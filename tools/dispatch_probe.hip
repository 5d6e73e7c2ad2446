// How long the dispatcher takes to get every wave of a launch started, as a function of the workgroup's size: the same
// number of waves as 4-wave and as 8-wave (and 16-wave) workgroups, each wave busy for ~5 us so that the CU's slots stay
// taken as in the fused plate step.   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/_build/dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <bool FAT>
__global__ void probe(unsigned long long* entry, unsigned long long* exit_, int busy_ticks) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (FAT) asm volatile("v_mov_b32 v230, 0" ::: "v230");       // (a wave of ~232 VGPRs: two per SIMD, as the plate step's)
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) entry[wid] = t0;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) exit_[wid] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    const int waves = 2048;
    unsigned long long *entry, *exit_;
    hipMalloc(&entry, waves * 8), hipMalloc(&exit_, waves * 8);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    std::vector<unsigned long long> he(waves), hx(waves);
    const int lds_per_wave = 5 * 1024;      // bytes: the fused plate step's ~4.5 KB per wave + its B table
    for (int fat = 0; fat < 2; ++fat)
    for (int wpg : {1, 2, 4, 8, 16}) {
        if (fat && wpg > 8) continue;
        for (int busy_us : {0, 5}) {
            double ev_us = 0, spread = 0, span = 0;
            const int reps = 20;
            for (int r = 0; r < reps + 3; ++r) {
                hipEventRecord(a, 0);
                if (fat)
                    hipLaunchKernelGGL(probe<true>, dim3(waves / wpg), dim3(64 * wpg), 8192 + lds_per_wave * wpg, 0, entry, exit_, busy_us * 100);
                else
                    hipLaunchKernelGGL(probe<false>, dim3(waves / wpg), dim3(64 * wpg), 8192 + lds_per_wave * wpg, 0, entry, exit_, busy_us * 100);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                if (r < 3) continue;
                float ms;
                hipEventElapsedTime(&ms, a, b);
                ev_us += ms * 1e3;
                hipMemcpy(he.data(), entry, waves * 8, hipMemcpyDeviceToHost);
                hipMemcpy(hx.data(), exit_, waves * 8, hipMemcpyDeviceToHost);
                const auto e0 = *std::min_element(he.begin(), he.end()), e1 = *std::max_element(he.begin(), he.end());
                const auto x1 = *std::max_element(hx.begin(), hx.end());
                spread += (e1 - e0) * 0.01, span += (x1 - e0) * 0.01;       // 100 MHz ticks -> us
            }
            printf("%s%2d waves per workgroup (%4d workgroups), waves busy %d us: last wave enters %.2f us after the first, "
                   "span first entry -> last exit %.2f us, events %.2f us\n", fat ? "232 VGPRs: " : "", wpg, waves / wpg, busy_us, spread / reps,
                   span / reps, ev_us / reps);
        }
    }
    return 0;
}
