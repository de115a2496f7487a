// mnrf_optim.hip -- Adam over one flat parameter tensor (training.FlatAdam).
//
// torch's fused Adam is a multi-tensor kernel that deals 65 536-element chunks to 512-thread blocks: the 595 k parameters of
// a field model are TEN blocks on a 256-CU device, 46 us per model and step, for 9.5 MB of traffic.  Here: one thread per four
// elements, float4 accesses, the same arithmetic (torch/optim/adam.py _single_tensor_adam, non-amsgrad, L2 weight decay; the
// bias corrections in double like torch's fused kernel), the same found_inf / grad_scale contract as GradScaler's
// (torch/optim/_functional + fused_adam_utils.cuh: a step with found_inf != 0 changes nothing and does not count).
// Reference: train.py:101-109 builds torch.optim.Adam through utils/__init__.py get_optimizer; this is the same update.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"

namespace mnrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    float lr, eps, wd;
    double beta1, beta2;
    long long step;            // the host's count of step() calls, this one included
    int* skipped;              // device: how many of them found_inf has skipped so far
    const float* grad_scale;   // device scalars or null
    const float* found_inf;
    // device-resident hyper-parameters and step count (mnrf_adam_step_dev: a step captured in a hipGraph must not freeze them)
    const double* hyper;       // [lr, beta1, beta2, eps, weight_decay] or null
    const long long* step_dev; // the count of step() calls, this one included, or null
    // range-guard words (mnrf.h MNRF_GUARD_*) that veto the update like found_inf does: a non-zero word = a launch of this step
    // left the range of the split arithmetic (no torch ops needed to turn the words into a found_inf tensor inside a captured step)
    const unsigned* guard[4];
    int n_guard;
};

__global__ void adam_kernel(AdamArgs A) {
    // thread 0 reads every scalar of the step (device-resident ones included) once per block and leaves what the others need in LDS
    __shared__ float sh[6];      // veto, step_size, sqrt(bias correction 2), beta1, beta2, inv_scale
    __shared__ float sh_eps_wd[2];
    if (threadIdx.x == 0) {
        if (A.hyper) {
            A.lr = (float)A.hyper[0]; A.beta1 = A.hyper[1]; A.beta2 = A.hyper[2]; A.eps = (float)A.hyper[3]; A.wd = (float)A.hyper[4];
            A.step = *A.step_dev;
        }
        bool veto = A.found_inf && *A.found_inf != 0.f;
        for (int i = 0; i < A.n_guard; ++i) veto |= *A.guard[i] != 0u;
        const long long eff = A.step - (long long)*A.skipped;      // >= 1 when the caller counts as documented
        const double t = (double)(eff < 1 ? 1 : eff);
        sh[0] = veto ? 1.f : 0.f;
        sh[1] = A.lr / (float)(1.0 - pow(A.beta1, t));
        sh[2] = (float)sqrt(1.0 - pow(A.beta2, t));
        sh[3] = (float)A.beta1;
        sh[4] = (float)A.beta2;
        sh[5] = A.grad_scale ? 1.f / *A.grad_scale : 1.f;
        sh_eps_wd[0] = A.eps;
        sh_eps_wd[1] = A.wd;
    }
    __syncthreads();
    if (sh[0] != 0.f) {      // the whole grid leaves; one thread records that this call did not count
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(A.skipped, 1);
        return;
    }
    const float step_size = sh[1];
    const float bc2_sqrt = sh[2];
    const float b1 = sh[3], b2 = sh[4];
    const float inv_scale = sh[5];
    A.eps = sh_eps_wd[0];
    A.wd = sh_eps_wd[1];
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= A.n) return;
    const bool full = i4 + 4 <= A.n;
    float p[4], g[4], m[4], v[4];
    if (full) {
        const f32x4 P = *(const f32x4*)(A.p + i4), G = *(const f32x4*)(A.g + i4), M = *(const f32x4*)(A.m + i4), V = *(const f32x4*)(A.v + i4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { p[c] = P[c]; g[c] = G[c]; m[c] = M[c]; v[c] = V[c]; }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in = i4 + c < A.n;
            p[c] = in ? A.p[i4 + c] : 0.f; g[c] = in ? A.g[i4 + c] : 0.f; m[c] = in ? A.m[i4 + c] : 0.f; v[c] = in ? A.v[i4 + c] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float gr = A.grad_scale ? g[c] * inv_scale : g[c];
        if (A.wd != 0.f) gr = gr + A.wd * p[c];
        m[c] = m[c] + (1.f - b1) * (gr - m[c]);                    // lerp(exp_avg, grad, 1 - beta1)
        v[c] = b2 * v[c] + (1.f - b2) * gr * gr;                   // exp_avg_sq * beta2 + (1 - beta2) grad^2
        const float denom = sqrtf(v[c]) / bc2_sqrt + A.eps;
        p[c] = p[c] - step_size * (m[c] / denom);
    }
    if (full) {
        *(f32x4*)(A.p + i4) = f32x4{p[0], p[1], p[2], p[3]};
        *(f32x4*)(A.m + i4) = f32x4{m[0], m[1], m[2], m[3]};
        *(f32x4*)(A.v + i4) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i4 + c < A.n) { A.p[i4 + c] = p[c]; A.m[i4 + c] = m[c]; A.v[i4 + c] = v[c]; }
    }
}

}  // namespace mnrf

extern "C" int mnrf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, double beta1,
                              double beta2, float eps, float weight_decay, int64_t step, int32_t* skipped, const float* grad_scale,
                              const float* found_inf, void* stream) {
    using namespace mnrf;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !skipped) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step: null pointer");
    if (n < 0 || step < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step: n >= 0, step >= 1");
    if (n == 0) return MNRF_OK;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step: tensors must be 16-byte aligned");
    AdamArgs A{param, grad, exp_avg, exp_avg_sq, (long long)n, lr, eps, weight_decay, beta1, beta2, (long long)step, skipped, grad_scale, found_inf,
               nullptr, nullptr, {nullptr, nullptr, nullptr, nullptr}, 0};
    const long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_adam_step");
}

// counter += delta, one thread (the step count of mnrf_adam_step_dev inside a captured step)
namespace mnrf {
__global__ void add_i64_kernel(long long* c, long long d) { *c += d; }
}  // namespace mnrf
extern "C" int mnrf_add_i64(int64_t* counter, int64_t delta, void* stream) {
    if (!counter) return mnrf_fail(MNRF_ERR_ARG, "mnrf_add_i64: null pointer");
    hipLaunchKernelGGL(mnrf::add_i64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)counter, (long long)delta);
    return mnrf_check_launch("mnrf_add_i64");
}

extern "C" int mnrf_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const double* hyper,
                                  const int64_t* step, int32_t* skipped, const float* grad_scale, const float* found_inf,
                                  const uint32_t* const* guard_words, int n_guard_words, void* stream) {
    using namespace mnrf;
    if (n_guard_words < 0 || n_guard_words > 4 || (n_guard_words > 0 && !guard_words))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: 0..4 guard words");
    if (!param || !grad || !exp_avg || !exp_avg_sq || !skipped || !hyper || !step) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: null pointer");
    if (n < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: n >= 0");
    if (n == 0) return MNRF_OK;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: tensors must be 16-byte aligned");
    AdamArgs A{param, grad, exp_avg, exp_avg_sq, (long long)n, 0.f, 0.f, 0.f, 0.0, 0.0, 1, skipped, grad_scale, found_inf, hyper,
               (const long long*)step, {nullptr, nullptr, nullptr, nullptr}, n_guard_words};
    for (int i = 0; i < n_guard_words; ++i) {
        if (!guard_words[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: null guard word");
        A.guard[i] = guard_words[i];
    }
    const long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_adam_step_dev");
}
