// Probe: what does one "tile" of the fused plate step cost on a SIMD -- a dependent chain of 10 v_mfma_f32_32x32x2_f32
// followed by the log-sum-exp VALU work on its 16 results -- with 1, 2, 4 waves per SIMD?  Do the f32 matrix
// instructions of one wave overlap the vector instructions of another?  Sets the yardstick for normal_lse_mfma_kernel.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_f32_probe.hip -o tools/_build/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA chain only; 1: chain + LSE VALU; 2: VALU only; 3: chain + VALU without the exps; 4: 4 chains then 4 LSEs
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a[10], b[10];
    for (int i = 0; i < 10; ++i) a[i] = seed + 0.001f * (lane + i), b[i] = 0.5f + 0.01f * i;
    float mn = 1e30f, sm = 0.f, accm = 0.f;
    f32x16 keep = {0};
    for (int it = 0; it < iters; ++it) {
        constexpr int NCH = MODE == 4 ? 4 : 1;
        f32x16 acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (MODE != 2) {
                acc[c] = f32x16{0};
#pragma unroll
                for (int s = 0; s < 10; ++s) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s] + (float)c, b[s], acc[c], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = a[r % 10] * b[(r + it) % 10] + keep[r];
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (MODE == 0) {
                keep = acc[c];
            } else {
                float tmin = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[c][r]);
                const float mnew = fminf(mn, tmin);
                float ssum = sm * (MODE == 3 ? (mnew - mn) : __builtin_amdgcn_exp2f(mnew - mn));
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += MODE == 3 ? (mnew - acc[c][r]) : __builtin_amdgcn_exp2f(mnew - acc[c][r]);
                mn = mnew, sm = ssum;
                if (MODE == 2) keep[it & 15] = ssum;
            }
        }
#pragma unroll
        for (int s = 0; s < 10; ++s) a[s] += 0.25f;          // (the A operand changes every tile)
    }
    accm = mn + sm + keep[0] + keep[5];
    if (accm == 1234.5678f) out[blockIdx.x] = accm;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
void run(const char *name, float *out) {
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {                               // waves per SIMD: 256 CUs x wps workgroups of 4 waves
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double tiles_per_simd = (double)iters * wps * (MODE == 4 ? 4 : 1);
        printf("%-34s waves/SIMD %d: %8.1f us, %7.1f ns per tile per SIMD (= %6.0f cycles at 2.4 GHz; 10 MFMAs = 640)\n",
               name, wps, ms * 1e3, ms * 1e6 / tiles_per_simd, ms * 1e6 / tiles_per_simd * 2.4);
    }
}

int main() {
    float *out; CK(hipMalloc(&out, 1 << 20));
    run<0>("MFMA chain only", out);
    run<1>("chain + LSE (16 exp)", out);
    run<2>("LSE VALU only", out);
    run<3>("chain + VALU without exp", out);
    run<4>("4 chains, then 4 LSEs", out);
    return 0;
}
