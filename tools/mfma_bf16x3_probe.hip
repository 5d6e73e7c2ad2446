// Probe for VERDICT r2 item 8: the fused plate step's tile on bf16 matrix instructions with 3-way split operands.
//   D[k][s] = sum_e a[k][e] * b[e][s]   (a = (v - mu)^2, b = log2e / (2 sigma^2), f32)
// computed as six bf16 products per event -- ah bh + ah bm + am bh + ah bl + al bh + am bm, f32 accumulate -- packed
// ALONG the contraction dim: 6 x (E + 1) = 114 slots for E = 18, i.e. 8 v_mfma_f32_32x32x16_bf16 (256 cycles) where
// the f32 path needs 10 v_mfma_f32_32x32x2_f32 (640 cycles) that also block the vector unit.
// Questions: (1) cycles per tile with the A-operand build (9 VALU per element) and the log-sum-exp beside the MFMAs, at
// 1..4 waves per SIMD, A shared by NST = 1 / 2 / 4 scale tiles; (2) the error of D against fp64, next to the f32 fma chain.
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_bf16x3_probe.hip -o tools/_build/mfma_bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// the three packed registers of one f32: (h, h), (m, h), (l, m)  [low half, high half]
__device__ __forceinline__ void split_a(float x, unsigned &r1, unsigned &r2, unsigned &r3) {
    r1 = cvt_pk(x, x);
    const float hf = __uint_as_float(r1 & 0xffff0000u);
    const float e1 = x - hf;
    r2 = cvt_pk(e1, hf);
    const float mf = __uint_as_float(r2 << 16);
    const float e2 = e1 - mf;
    r3 = cvt_pk(e2, e1);
}
// the partner layout: (h, m), (h, l), (h, m)
__device__ __forceinline__ void split_b(float x, unsigned &r1, unsigned &r2, unsigned &r3) {
    const unsigned hh = cvt_pk(x, x);
    const float hf = __uint_as_float(hh & 0xffff0000u);
    const float e1 = x - hf;
    const unsigned mm = cvt_pk(e1, e1);
    const float mf = __uint_as_float(mm & 0xffff0000u);
    const float e2 = e1 - mf;
    const unsigned ll = cvt_pk(e2, e2);
    r1 = (hh & 0xffffu) | (mm & 0xffff0000u);
    r2 = (hh & 0xffffu) | (ll & 0xffff0000u);
    r3 = r1;
}

constexpr int EQ = 10;                     // events per lane half (E + 1 = 19 -> 10 / 9 + one pad)
constexpr int NV = 32;                     // operand registers per lane: 3 EQ = 30, padded to 8 steps x 4

// MODE 0: MFMAs only; 1: MFMAs + LSE; 2: A build + MFMAs + LSE; 3: A build + LSE, no MFMAs; 4: A build only
template <int MODE, int NST>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float zv[EQ], mu[EQ];
    for (int i = 0; i < EQ; ++i) zv[i] = seed + 0.001f * (lane + i), mu[i] = 0.3f + 0.01f * i;
    unsigned breg[NST][NV];
    for (int st = 0; st < NST; ++st)
        for (int i = 0; i < NV; ++i) breg[st][i] = 0x3f803f80u + (unsigned)(lane + i + st);
    unsigned areg[NV];
    for (int i = 0; i < NV; ++i) areg[i] = 0x3f803f80u + (unsigned)lane;
    float mn[NST], sm[NST];
    for (int st = 0; st < NST; ++st) mn[st] = 1e30f, sm[st] = 0.f;
    f32x16 keep = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
#pragma unroll
            for (int q = 0; q < EQ; ++q) {
                const float df = zv[q] - mu[q];
                split_a(df * df, areg[3 * q], areg[3 * q + 1], areg[3 * q + 2]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) areg[i] += 0x10001u;
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x16 acc = {0};
            if (MODE != 3 && MODE != 4) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const u32x4 av = {areg[4 * s], areg[4 * s + 1], areg[4 * s + 2], areg[4 * s + 3]};
                    const u32x4 bv = {breg[st][4 * s], breg[st][4 * s + 1], breg[st][4 * s + 2], breg[st][4 * s + 3]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __uint_as_float(areg[r + st]) + keep[r];
            }
            if (MODE == 0 || MODE == 4) {
                keep += acc;
            } else {
                float tmin = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmin = fminf(tmin, acc[r]);
                const float mnew = fminf(mn[st], tmin);
                float ssum = sm[st] * __builtin_amdgcn_exp2f(mnew - mn[st]);
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += __builtin_amdgcn_exp2f(mnew - acc[r]);
                mn[st] = mnew, sm[st] = ssum;
            }
        }
#pragma unroll
        for (int q = 0; q < EQ; ++q) zv[q] += 0.25f;
    }
    float accm = keep[0] + keep[5] + keep[11];
    for (int st = 0; st < NST; ++st) accm += mn[st] + sm[st];
    if (accm == 1234.5678f) out[blockIdx.x] = accm;
}

// ---- numerics: one 32 x 32 tile, E + 1 = 19 events, against fp64 and against the f32 fma chain
__global__ void tile_check(const float *a, const float *b, float *d_split, float *d_f32) {
    // a[32][20] (row k, event e), b[32][20] (column s, event e); lane half h takes events 2 q + h
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    unsigned areg[NV], breg[NV];
    for (int i = 0; i < NV; ++i) areg[i] = breg[i] = 0;
    for (int q = 0; q < EQ; ++q) {
        const int e = 2 * q + h;
        split_a(a[j * 20 + e], areg[3 * q], areg[3 * q + 1], areg[3 * q + 2]);
        split_b(b[j * 20 + e], breg[3 * q], breg[3 * q + 1], breg[3 * q + 2]);
    }
    f32x16 acc = {0}, acc32 = {0};
    for (int s = 0; s < 8; ++s) {
        const u32x4 av = {areg[4 * s], areg[4 * s + 1], areg[4 * s + 2], areg[4 * s + 3]};
        const u32x4 bv = {breg[4 * s], breg[4 * s + 1], breg[4 * s + 2], breg[4 * s + 3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
    }
    for (int q = 0; q < EQ; ++q) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j * 20 + 2 * q + h], b[j * 20 + 2 * q + h], acc32, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        d_split[row * 32 + j] = acc[r];
        d_f32[row * 32 + j] = acc32[r];
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int NST>
void run(const char *name, float *out) {
    const int iters = 2000;
    for (int wps : {1, 2, 3, 4}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((probe<MODE, NST>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE, NST>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double tiles_per_simd = (double)iters * wps * NST;
        printf("%-40s NST %d waves/SIMD %d: %8.1f us, %7.1f ns per tile per SIMD (f32 path: 442)\n",
               name, NST, wps, ms * 1e3, ms * 1e6 / tiles_per_simd);
    }
}

int main() {
    float *out; CK(hipMalloc(&out, 1 << 20));
    // numerics first
    {
        std::vector<float> a(32 * 20, 0.f), b(32 * 20, 0.f);
        srand(7);
        for (int r = 0; r < 32; ++r)
            for (int e = 0; e < 19; ++e) {
                const float u = (float)rand() / RAND_MAX, v = (float)rand() / RAND_MAX;
                a[r * 20 + e] = e == 18 ? 40.f * u : 30.f * u * u;                  // (v - mu)^2; slot: -sum small
                b[r * 20 + e] = e == 18 ? 1.44269504f : 1.44269504f * 0.5f / (0.05f + 2.f * v * v);
            }
        float *da, *db, *d1, *d2;
        CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&d1, 4096)); CK(hipMalloc(&d2, 4096));
        CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(tile_check, dim3(1), dim3(64), 0, 0, da, db, d1, d2);
        std::vector<float> r1(1024), r2(1024);
        CK(hipMemcpy(r1.data(), d1, 4096, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), d2, 4096, hipMemcpyDeviceToHost));
        double e_split = 0, e_f32 = 0, rel_split = 0, rel_f32 = 0;
        for (int k = 0; k < 32; ++k)
            for (int s = 0; s < 32; ++s) {
                double ref = 0, mag = 0;
                for (int e = 0; e < 19; ++e) ref += (double)a[k * 20 + e] * b[s * 20 + e], mag += fabs((double)a[k * 20 + e] * b[s * 20 + e]);
                e_split = fmax(e_split, fabs(r1[k * 32 + s] - ref)), e_f32 = fmax(e_f32, fabs(r2[k * 32 + s] - ref));
                rel_split = fmax(rel_split, fabs(r1[k * 32 + s] - ref) / mag), rel_f32 = fmax(rel_f32, fabs(r2[k * 32 + s] - ref) / mag);
            }
        printf("numerics, 32x32 tile, 19 events: max |err| vs fp64  bf16x3 %.3e (rel to sum|ab| %.3e)   f32 fma chain %.3e (rel %.3e)\n",
               e_split, rel_split, e_f32, rel_f32);
    }
    run<0, 1>("8 bf16 MFMAs only", out);
    run<1, 1>("MFMAs + LSE", out);
    run<2, 1>("A build + MFMAs + LSE", out);
    run<3, 1>("A build + LSE, no MFMAs", out);
    run<4, 1>("A build only", out);
    run<2, 2>("A build + 2 x (MFMAs + LSE)", out);
    run<2, 4>("A build + 4 x (MFMAs + LSE)", out);
    run<1, 4>("4 x (MFMAs + LSE)", out);
    return 0;
}
