// alan_adam_step: the optimiser step of the reference's training loop (basic_runner.py:108-110: opt.step() of a
// torch.optim.Adam over the problem's parameters) as ONE launch of this library for all parameter tensors, its step count on
// the device -- so that a whole training iteration (sample -> elbo_vi / elbo_rws -> backward -> step) consists of library
// launches alone and can be re-issued from a recorded launch list (calls.hip) like an evaluation.  Until round 3 the step
// was torch's: two multi_tensor_apply kernels (the step counters' increment and the fused update), which kept a training
// iteration a HIP-graph replay with its idle time between graph launches.
//
// Arithmetic: Adam (Kingma & Ba 2015) exactly as torch.optim.Adam(capturable=True, fused=True) evaluates it for fp32
// tensors -- the moment updates and the two divisions in double where torch's scalars are double, results rounded to fp32
// where torch stores fp32 -- so that switching optimisers does not change a training run:
//     m = beta1 m + (1 - beta1) g;   v = beta2 v + (1 - beta2) g g;   (g -> -g when maximising)
//     p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include <cmath>
#include <cstring>

#include "common.h"

namespace alan {

constexpr int ADAM_THREADS = 256, ADAM_ILP = 4, ADAM_CHUNK = ADAM_THREADS * ADAM_ILP;

struct AdamKern {
    float *p[ALAN_ADAM_MAX_TENSORS];
    const float *g[ALAN_ADAM_MAX_TENSORS];
    float *m[ALAN_ADAM_MAX_TENSORS], *v[ALAN_ADAM_MAX_TENSORS];
    int32_t n[ALAN_ADAM_MAX_TENSORS], first_block[ALAN_ADAM_MAX_TENSORS + 1];
    int32_t n_tensors, maximize;
    double lr, beta1, beta2, eps;
    float *step;
    int32_t *ticket;
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(const AdamKern a) {
    // (the library is built with -ffp-contract=off; torch's kernels are built with hipcc's default, which fuses a * b + c:
    // the same contraction here, so that the moment updates round as torch's do)
#pragma clang fp contract(fast)
    int ti = 0;
    while (ti + 1 < a.n_tensors && (int)blockIdx.x >= a.first_block[ti + 1]) ++ti;              // (workgroup-uniform)
    const int chunk = blockIdx.x - a.first_block[ti];
    // the step this launch takes: every workgroup reads the count BEFORE anyone advances it (the last one to finish does)
    const float t_new = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.f;
    const double bc1 = 1.0 - pow(a.beta1, (double)t_new), bc2 = 1.0 - pow(a.beta2, (double)t_new);
    const float bias_correction1 = (float)bc1, bias_correction2_sqrt = (float)sqrt(bc2);
    const float step_size = (float)(a.lr / (double)bias_correction1);
    float *p = a.p[ti], *m = a.m[ti], *v = a.v[ti];
    const float *g = a.g[ti];
    const int n = a.n[ti];
#pragma unroll
    for (int u = 0; u < ADAM_ILP; ++u) {
        const int i = chunk * ADAM_CHUNK + u * ADAM_THREADS + threadIdx.x;
        if (i < n) {
            float grad = g[i];
            if (a.maximize) grad = -grad;
            const float mo = (float)(a.beta1 * (double)m[i] + (1.0 - a.beta1) * (double)grad);
            const float vo = (float)(a.beta2 * (double)v[i] + (1.0 - a.beta2) * (double)grad * (double)grad);
            const float denom = (float)((double)(sqrtf(vo) / bias_correction2_sqrt) + a.eps);
            p[i] -= step_size * mo / denom;
            m[i] = mo, v[i] = vo;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (int)gridDim.x - 1) {
            __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
            __hip_atomic_store(a.step, t_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace alan

using namespace alan;

extern "C" int alan_adam_step(const alan_adam_desc_t *d, void *stream) {
    if (!d || d->n_tensors < 1 || d->n_tensors > ALAN_ADAM_MAX_TENSORS || !d->step || !d->ticket) return ALAN_ERR_BAD_DESC;
    if (!(d->lr >= 0) || !(d->beta1 >= 0 && d->beta1 < 1) || !(d->beta2 >= 0 && d->beta2 < 1) || !(d->eps >= 0)) return ALAN_ERR_BAD_DESC;
    AdamKern k;
    std::memset(&k, 0, sizeof(k));
    int64_t blocks = 0;
    for (int i = 0; i < d->n_tensors; ++i) {
        if (!d->param[i] || !d->grad[i] || !d->exp_avg[i] || !d->exp_avg_sq[i] || d->numel[i] < 1) return ALAN_ERR_BAD_DESC;
        if (d->numel[i] >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
        k.p[i] = (float *)d->param[i], k.g[i] = (const float *)d->grad[i];
        k.m[i] = (float *)d->exp_avg[i], k.v[i] = (float *)d->exp_avg_sq[i];
        k.n[i] = (int32_t)d->numel[i];
        k.first_block[i] = (int32_t)blocks;
        blocks += (d->numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        if (blocks >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    }
    k.first_block[d->n_tensors] = (int32_t)blocks;
    k.n_tensors = d->n_tensors, k.maximize = d->maximize;
    k.lr = d->lr, k.beta1 = d->beta1, k.beta2 = d->beta2, k.eps = d->eps;
    k.step = (float *)d->step, k.ticket = (int32_t *)d->ticket;
    ALAN_LAUNCH(adam_kernel, dim3((uint32_t)blocks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
}
