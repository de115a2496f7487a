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
    // mnrf_adam_step_dev: every scalar of the step comes from a device block written by adam_prep_kernel (a step captured in a hipGraph
    // must not freeze them): [veto, lr / bias correction 1, sqrt(bias correction 2), beta1, beta2, 1 / grad_scale, eps, weight_decay]
    const float* state;
};

__device__ __forceinline__ void adam_body(AdamArgs A) {
    // thread 0 reads the scalars of the step once per block and leaves what the others need in LDS
    __shared__ float sh[8];
    if (threadIdx.x == 0) {
        if (A.state) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sh[i] = A.state[i];
        } else {
            const bool veto = A.found_inf && *A.found_inf != 0.f;
            const long long eff = A.step - (long long)*A.skipped;      // >= 1 when the caller counts as documented
            const double t = (double)(eff < 1 ? 1 : eff);
            sh[0] = veto ? 1.f : 0.f;
            sh[1] = A.lr / (float)(1.0 - pow(A.beta1, t));
            sh[2] = (float)sqrt(1.0 - pow(A.beta2, t));
            sh[3] = (float)A.beta1;
            sh[4] = (float)A.beta2;
            sh[5] = A.grad_scale ? 1.f / *A.grad_scale : 1.f;
            sh[6] = A.eps;
            sh[7] = A.wd;
        }
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
    const bool scaled = sh[5] != 1.f;
    A.eps = sh[6];
    A.wd = sh[7];
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
        float gr = scaled ? g[c] * inv_scale : g[c];
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


__global__ void adam_kernel(AdamArgs A) { adam_body(A); }

// the same for up to ADAM_BATCH flat tensors in one launch (blockIdx.y = tensor): a training step updates its coarse and its fine
// model behind ONE prep launch (mnrf_adam_step_dev_n)
constexpr int ADAM_BATCH = 4;
struct AdamBatch {
    AdamArgs a[ADAM_BATCH];
};
__global__ void adam_batch_kernel(AdamBatch B) { adam_body(B.a[blockIdx.y]); }

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
               nullptr};
    const long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_adam_step");
}

// ---- the same update with every scalar in device memory (a step captured in a hipGraph must not freeze the learning rate or the
// step count).  mnrf_adam_prep: one thread -- *step += 1, then the scalars of this step -> state[8]; mnrf_adam_step_dev: the update.
namespace mnrf {
struct AdamPrepArgs {
    const double* hyper; long long* step; const int* skipped; const float* grad_scale; const float* found_inf;
    const unsigned* guard[4]; int n_guard; float* state;
};
__global__ void adam_prep_kernel(AdamPrepArgs P) {
    const long long step = *P.step + 1;
    *P.step = step;
    bool veto = P.found_inf && *P.found_inf != 0.f;
    for (int i = 0; i < P.n_guard; ++i) veto |= *P.guard[i] != 0u;      // a range-guard word of this step's launches (mnrf.h MNRF_GUARD_*)
    const float lr = (float)P.hyper[0], eps = (float)P.hyper[3], wd = (float)P.hyper[4];
    const double beta1 = P.hyper[1], beta2 = P.hyper[2];
    const long long eff = step - (long long)*P.skipped;
    const double t = (double)(eff < 1 ? 1 : eff);
    P.state[0] = veto ? 1.f : 0.f;
    P.state[1] = lr / (float)(1.0 - pow(beta1, t));
    P.state[2] = (float)sqrt(1.0 - pow(beta2, t));
    P.state[3] = (float)beta1;
    P.state[4] = (float)beta2;
    P.state[5] = P.grad_scale ? 1.f / *P.grad_scale : 1.f;
    P.state[6] = eps;
    P.state[7] = wd;
}
}  // namespace mnrf

extern "C" int mnrf_adam_prep(const double* hyper, int64_t* step, const int32_t* skipped, const float* grad_scale, const float* found_inf,
                              const uint32_t* const* guard_words, int n_guard_words, float* state, void* stream) {
    using namespace mnrf;
    if (!hyper || !step || !skipped || !state) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_prep: null pointer");
    if (n_guard_words < 0 || n_guard_words > 4 || (n_guard_words > 0 && !guard_words))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_prep: 0..4 guard words");
    AdamPrepArgs P{hyper, (long long*)step, skipped, grad_scale, found_inf, {nullptr, nullptr, nullptr, nullptr}, n_guard_words, state};
    for (int i = 0; i < n_guard_words; ++i) {
        if (!guard_words[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_prep: null guard word");
        P.guard[i] = guard_words[i];
    }
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, P);
    return mnrf_check_launch("mnrf_adam_prep");
}

extern "C" int mnrf_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* state,
                                  int32_t* skipped, void* stream) {
    using namespace mnrf;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !skipped || !state) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: null pointer");
    if (n < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: n >= 0");
    if (n == 0) return MNRF_OK;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev: tensors must be 16-byte aligned");
    AdamArgs A{param, grad, exp_avg, exp_avg_sq, (long long)n, 0.f, 0.f, 0.f, 0.0, 0.0, 1, skipped, nullptr, nullptr, state};
    const long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_adam_step_dev");
}

extern "C" int mnrf_adam_step_dev_n(int n_tensors, float* const* param, const float* const* grad, float* const* exp_avg,
                                    float* const* exp_avg_sq, const int64_t* n, const float* state, int32_t* const* skipped, void* stream) {
    using namespace mnrf;
    if (n_tensors < 0 || n_tensors > ADAM_BATCH) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev_n: 0..4 tensors per call");
    if (n_tensors == 0) return MNRF_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !n || !skipped || !state) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev_n: null pointer");
    AdamBatch B{};
    long long most = 0;
    for (int t = 0; t < n_tensors; ++t) {
        if (!param[t] || !grad[t] || !exp_avg[t] || !exp_avg_sq[t] || !skipped[t]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev_n: null pointer");
        if (n[t] < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev_n: n >= 0");
        if (((uintptr_t)param[t] | (uintptr_t)grad[t] | (uintptr_t)exp_avg[t] | (uintptr_t)exp_avg_sq[t]) & 15)
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_adam_step_dev_n: tensors must be 16-byte aligned");
        B.a[t] = AdamArgs{param[t], grad[t], exp_avg[t], exp_avg_sq[t], (long long)n[t], 0.f, 0.f, 0.f, 0.0, 0.0, 1, skipped[t], nullptr, nullptr, state};
        most = n[t] > most ? n[t] : most;
    }
    if (most == 0) return MNRF_OK;
    const long long threads = (most + 3) / 4;
    hipLaunchKernelGGL(adam_batch_kernel, dim3((unsigned)((threads + 255) / 256), n_tensors), dim3(256), 0, (hipStream_t)stream, B);
    return mnrf_check_launch("mnrf_adam_step_dev_n");
}
