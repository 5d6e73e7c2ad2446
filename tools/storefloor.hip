// Store-pattern probe for the factor producer: how fast can 152 MB (K=100, 38 users) be WRITTEN with
//   (a) 4 bytes per lane, a wave writing 256 contiguous bytes, rows of 400 bytes walked by successive stores (the
//       vector kernel's pattern),
//   (b) 4 bytes per lane, a wave writing 2 x 128 bytes (the MFMA kernel with lanes along the value rows),
//   (c) 16 bytes per lane, a wave writing 32 rows x 2 pieces of 16 bytes, 4 such stores completing 128 bytes per row
//       (the MFMA kernel with lanes along the SCALE rows: each lane's 4 accumulator registers are 4 consecutive kz),
//   (d) 16 bytes per lane fully contiguous (the fill floor).
// Build: hipcc -O3 --offload-arch=gfx950 tools/storefloor.hip -o tools/_build/storefloor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KZ = 100, KS = 100, KL = 100, M = 38;          // out[m][l][s][kz]

// one wave per (m, l, 64-wide kz block... ) -- (a): wave = 64 consecutive kz of row (m, l, s), loops s
__global__ __launch_bounds__(256) void pat_a(float *out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nvt = (M * KZ + 63) / 64;                       // value-row tiles of 64
    const int l = wave / nvt, vt = wave - l * nvt;
    if (l >= KL) return;
    const int v = vt * 64 + lane;
    if (v >= M * KZ) return;
    const int m = v / KZ, kz = v - m * KZ;
    float *p = out + ((size_t)m * KL + l) * KS * KZ + kz;
    for (int s = 0; s < KS; ++s) p[(size_t)s * KZ] = (float)s;
}
// (b): wave = 32 consecutive value rows x rows s, s + 4 per store
__global__ __launch_bounds__(256) void pat_b(float *out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nvt = (M * KZ + 31) / 32;
    const int l = wave / nvt, vt = wave - l * nvt;
    if (l >= KL) return;
    const int v = vt * 32 + (lane & 31), h = lane >> 5;
    if (v >= M * KZ) return;
    const int m = v / KZ, kz = v - m * KZ;
    float *p = out + ((size_t)m * KL + l) * KS * KZ + kz;
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = 32 * st + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (s < KS) p[(size_t)s * KZ] = (float)s;
        }
}
// (c): lane = scale row j (+ half h), 16 bytes = 4 consecutive value rows
__global__ __launch_bounds__(256) void pat_c(float *out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nvt = (M * KZ + 31) / 32;
    const int l = wave / nvt, vt = wave - l * nvt;
    if (l >= KL) return;
    const int j = lane & 31, h = lane >> 5;
    for (int st = 0; st < 4; ++st) {
        const int s = 32 * st + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = vt * 32 + 8 * q + 4 * h;            // 4 consecutive value rows v .. v + 3 (same m: KZ % 4 == 0)
            if (s < KS && v < M * KZ) {
                const int m = v / KZ, kz = v - m * KZ;
                *reinterpret_cast<f32x4 *>(out + (((size_t)m * KL + l) * KS + s) * KZ + kz) = f32x4{1.f, 2.f, 3.f, (float)s};
            }
        }
    }
}
__global__ __launch_bounds__(256) void pat_d(f32x4 *out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n4; i += (size_t)gridDim.x * 256) out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename L>
static float time_us(L launch, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(nullptr, nullptr);
    CK(hipStreamSynchronize(s));
    float tot = 0.f;
    for (int i = 0; i < 20; ++i) {
        launch(a, b);
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    return tot / 20 * 1e3f;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t n = (size_t)M * KL * KS * KZ;
    float *out; CK(hipMalloc(&out, n * 4 + 256));
    const double mb = n * 4 / 1e6;
    const int wa = KL * ((M * KZ + 63) / 64), wb = KL * ((M * KZ + 31) / 32);
    float u;
    u = time_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pat_a, dim3((wa + 3) / 4), dim3(256), 0, s, a, b, 0, out); }, s);
    printf("(a) dword, 256 B per wave-store      %7.2f us %6.0f GB/s\n", u, mb / u * 1e3);
    u = time_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pat_b, dim3((wb + 3) / 4), dim3(256), 0, s, a, b, 0, out); }, s);
    printf("(b) dword, 2 x 128 B per wave-store  %7.2f us %6.0f GB/s\n", u, mb / u * 1e3);
    u = time_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pat_c, dim3((wb + 3) / 4), dim3(256), 0, s, a, b, 0, out); }, s);
    printf("(c) dwordx4, 32 rows x 2 x 16 B      %7.2f us %6.0f GB/s\n", u, mb / u * 1e3);
    u = time_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pat_d, dim3(8192), dim3(256), 0, s, a, b, 0, (f32x4 *)out, n / 4); }, s);
    printf("(d) dwordx4 contiguous fill          %7.2f us %6.0f GB/s\n", u, mb / u * 1e3);
    return 0;
}
