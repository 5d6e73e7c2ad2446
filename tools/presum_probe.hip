// Probe: can ONE single-workgroup launch replace the fused plate step's second-stage sum (150 chunk partials of a
// [30, 30] result) AND the top-level log-sum-exp that follows it?  Times, inside replayed graphs of 20 launches:
//   (a) two launches: a 900-output sum over C chunks (many workgroups), then a 1-workgroup log-sum-exp over 900
//   (b) one launch of 1024 threads doing both
// Build: hipcc -O3 --offload-arch=gfx950 tools/presum_probe.hip -o tools/_build/presum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NE = 900;

__global__ __launch_bounds__(256) void sum_k(const float *part, int C, float *out) {
    const int gid = blockIdx.x * 256 + threadIdx.x, e = gid >> 4, l = gid & 15;      // 16 lanes per output
    if (e >= NE) return;
    float s = 0.f;
    for (int c = l; c < C; c += 16) s += part[(size_t)c * NE + e];
    for (int o = 8; o; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0) out[e] = s;
}
__global__ __launch_bounds__(256) void lse_k(const float *x, const float *a, const float *b, float *out) {
    __shared__ float sm[4], ss[4];
    float m = -1e30f, s = 0.f;
    for (int e = threadIdx.x; e < NE; e += 256) {
        const float v = x[e] + a[e / 30] + b[e % 30];
        const float mm = fmaxf(m, v);
        s = s * __expf(m - mm) + __expf(v - mm), m = mm;
    }
    for (int o = 32; o; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o), mm = fmaxf(m, m2);
        s = s * __expf(m - mm) + s2 * __expf(m2 - mm), m = mm;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m, ss[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { const float mm = fmaxf(m, sm[i]); s = s * __expf(m - mm) + ss[i] * __expf(sm[i] - mm), m = mm; }
        out[0] = logf(s) + m;
    }
}
__global__ __launch_bounds__(1024) void both_k(const float *part, int C, const float *a, const float *b, float *out) {
    __shared__ float S[NE];
    __shared__ float sm[16], ss[16];
    const int tid = threadIdx.x;
    // phase 1: thread (cp = tid / 256, r = tid % 256): a quarter of the chunks for elements r, r + 256, ...
    for (int e = tid; e < NE; e += 1024) S[e] = 0.f;
    __syncthreads();
    const int cp = tid >> 8, r = tid & 255;
    const int c0 = cp * ((C + 3) / 4), c1 = min(C, c0 + (C + 3) / 4);
    for (int e = r; e < NE; e += 256) {
        float acc[8] = {0.f};
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(c + u) * NE + e];
        }
        float s = 0.f;
        for (; c < c1; ++c) s += part[(size_t)c * NE + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += acc[u];
        atomicAdd(&S[e], s);
    }
    __syncthreads();
    float m = -1e30f, s = 0.f;
    for (int e = tid; e < NE; e += 1024) {
        const float v = S[e] + a[e / 30] + b[e % 30];
        m = v, s = 1.f;
    }
    for (int o = 32; o; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o), mm = fmaxf(m, m2);
        s = s * __expf(m - mm) + s2 * __expf(m2 - mm), m = mm;
    }
    if ((tid & 63) == 0) sm[tid >> 6] = m, ss[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) { const float mm = fmaxf(m, sm[i]); s = s * __expf(m - mm) + ss[i] * __expf(sm[i] - mm), m = mm; }
        out[0] = logf(s) + m;
    }
}
int main() {
    for (int C : {150, 60, 30}) {
        float *part, *x, *a, *b, *o1, *o2;
        hipMalloc(&part, sizeof(float) * C * NE); hipMalloc(&x, sizeof(float) * NE); hipMalloc(&a, 120); hipMalloc(&b, 120);
        hipMalloc(&o1, 4); hipMalloc(&o2, 4);
        std::vector<float> h(C * NE);
        for (size_t i = 0; i < h.size(); ++i) h[i] = -1.f - (float)((i * 2654435761u) % 1000) / 500.f;
        hipMemcpy(part, h.data(), sizeof(float) * C * NE, hipMemcpyHostToDevice);
        hipMemset(a, 0, 120); hipMemset(b, 0, 120);
        hipStream_t st; hipStreamCreate(&st);
        auto timeit = [&](auto body) {
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int i = 0; i < 20; ++i) body();
            hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st);
            for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms * 1e3f / 200.f;
        };
        const float two = timeit([&] { hipLaunchKernelGGL(sum_k, dim3((NE * 16 + 255) / 256), dim3(256), 0, st, part, C, x);
                                        hipLaunchKernelGGL(lse_k, dim3(1), dim3(256), 0, st, x, a, b, o1); });
        const float one = timeit([&] { hipLaunchKernelGGL(both_k, dim3(1), dim3(1024), 0, st, part, C, a, b, o2); });
        float r1, r2; hipMemcpy(&r1, o1, 4, hipMemcpyDeviceToHost); hipMemcpy(&r2, o2, 4, hipMemcpyDeviceToHost);
        printf("C=%3d chunks: two launches %.2f us, one launch %.2f us per evaluation  (results %.5f %.5f)\n", C, two, one, r1, r2);
    }
    return 0;
}
