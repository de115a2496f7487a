// mnrf_loss.hip -- the loss reductions of the training step, value AND gradient in one pass.
//
// Replaces losses.py:7-255 of the reference (ColorLoss, MirrorMaskLoss, NormalLoss, NormalRegLoss,
// PlaneConsistentLoss, TotalLoss): ~40 torch reductions per step there, with boolean-mask gathers, and a Python
// loop of floor(M/4) iterations x 4 `.item()` calls in PlaneConsistentLoss (losses.py:96-110) that would dominate
// a 20 ms step.  Every term is a mean of per-ray (or per-sample) quantities, so the gradient with respect to every
// input of the loss is known in closed form once a few counts are: the kernels below write d(total)/d(input) next
// to the sums, and the autograd node of mirror_nerf_amd/losses.py only scales them by the incoming gradient.
//   count_kernel     one workgroup: #rays with gt < 0, #mirror rays by gt, #valid rays, #non-mirror rays by the
//                    thresholded prediction; list of the mirror rows in order (the plane loss indexes it)
//   ray_kernel       per ray: colour, mask BCE, normal_dif terms and their gradients; block partial sums
//   sample_kernel    per (ray, sample): NormalRegLoss terms and gradients (pred_normal, weights, normal_fine)
//   plane_kernel     per drawn quadruple: |(p1-p0) x (p2-p0) . (p3-p0)| and its gradient (atomics on x_surface)
//   finish_kernel    fixed-order sums of the block partials -> the five terms and their total
// HBM-bound streaming over 36 B/ray + 28 B/sample; deterministic except for the plane loss's atomics.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"
#include "mnrf_fill.h"

namespace {

constexpr int TPB = 256;
enum { C_INVALID = 0, C_MIRROR = 1, C_VALID = 2, C_PRED_NOT = 3, N_COUNTS = 8 };
enum { T_COLOR = 0, T_MASK = 1, T_PLANE = 2, T_NORMAL = 3, T_REG = 4, N_TERMS = 5 };

__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;   // valid in thread 0
}

// which prediction the train_geometry_stage / invalid-GT branch of ColorLoss thresholds (losses.py:24-29)
__device__ __forceinline__ float* thresholded_key(const MnrfLossArgs& A) {
    return A.mirror_mask[1] ? A.mirror_mask[1] : A.mirror_mask[0];
}

__global__ void count_kernel(MnrfLossArgs A, float* counts, int* mirror_rows) {
    __shared__ int sh[4][1024 / 64 + 1];
    __shared__ int base;
    const int tid = threadIdx.x;
    int c_inv = 0, c_val = 0, c_pn = 0;
    if (tid == 0) base = 0;
    __syncthreads();
    const float* pm = thresholded_key(A);
    for (long long i0 = 0; i0 < A.n_rays; i0 += blockDim.x) {
        const long long i = i0 + tid;
        const bool in = i < A.n_rays;
        const float g = (in && A.gt_mask) ? A.gt_mask[i] : 0.f;
        c_inv += in && g < 0.f;
        c_val += in && (A.valid_mask ? A.valid_mask[i] != 0 : true);
        c_pn += in && pm && pm[i] < 0.5f;       // after the in-place threshold: 0 <=> prediction below 0.5
        // ordered list of mirror rows (gt != 0): ballot prefix inside the wave, wave offsets through LDS
        const bool mir = in && A.gt_mask && g != 0.f;
        const unsigned long long b = __ballot(mir);
        const int w = tid >> 6, l = tid & 63;
        if (l == 0) sh[0][w] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int k = 0; k < w; ++k) off += sh[0][k];
        if (mir && mirror_rows) mirror_rows[off + __popcll(b & ((1ull << l) - 1ull))] = (int)i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[0][k];
            base += t;
        }
        __syncthreads();
    }
    // totals
    int v[3] = {c_inv, c_val, c_pn};
    for (int q = 0; q < 3; ++q) {
        int x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if ((tid & 63) == 0) sh[q + 1][tid >> 6] = x;
    }
    __syncthreads();
    if (tid == 0) {
        int t[3] = {0, 0, 0};
        for (int q = 0; q < 3; ++q)
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t[q] += sh[q + 1][k];
        counts[C_INVALID] = (float)t[0];
        counts[C_MIRROR] = (float)base;
        counts[C_VALID] = (float)t[1];
        counts[C_PRED_NOT] = (float)t[2];
    }
}

__device__ __forceinline__ float clamped_log(float x, bool clamp) {
    const float l = logf(x);
    return clamp ? fmaxf(l, -100.f) : l;   // nn.BCELoss clamps its logs at -100; utils/func.py:32-37 does not
}

__global__ void ray_kernel(MnrfLossArgs A, const float* counts, float* partials) {
    __shared__ float sh[TPB / 64];
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool in = i < A.n_rays;
    const float n = (float)A.n_rays;
    const bool any_invalid = counts[C_INVALID] > 0.f;
    const float n_mirror = counts[C_MIRROR];
    const bool stage = A.flags & MNRF_LOSS_GEOMETRY_STAGE;
    const float gt = (in && A.gt_mask) ? A.gt_mask[i] : 0.f;

    // ---- ColorLoss (losses.py:14-51): which rows, and how many
    float color = 0.f;
    {
        bool sel = in;
        float cnt = n;
        if (stage && A.gt_mask && any_invalid) {
            float* pm = thresholded_key(A);
            if (pm) {
                // the reference thresholds inputs["mirror_mask_*"].detach() IN PLACE (shared storage): the dict entry
                // the mask loss reads afterwards holds 0 / 1 (0.5 stays); restated as such
                if (in) {
                    float m = pm[i];
                    m = m > 0.5f ? 1.f : (m < 0.5f ? 0.f : m);
                    pm[i] = m;
                    sel = m == 0.f;
                }
                cnt = counts[C_PRED_NOT];
            } else {
                sel = false;      // "loss = 0"
                cnt = -1.f;
            }
        } else if (stage && A.gt_mask && (A.flags & MNRF_LOSS_WO_MASK_RGB_TO_BLACK)) {
            sel = in && gt == 0.f;
            cnt = n - n_mirror;
        }
        for (int typ = 0; typ < 2; ++typ) {
            if (!A.rgb[typ]) continue;
            float g[3] = {0.f, 0.f, 0.f};
            if (sel && cnt >= 0.f) {
                for (int c = 0; c < 3; ++c) {
                    const float d = A.rgb[typ][i * 3 + c] - A.targets[i * 3 + c];
                    color += d * d;
                    g[c] = A.w_color * 2.f * d / (3.f * cnt);
                }
            }
            if (in && A.g_rgb[typ])
                for (int c = 0; c < 3; ++c) A.g_rgb[typ][i * 3 + c] = g[c];
        }
        // every typ shares the denominator: divide the block sums later; an empty selection gives 0/0 = nan like torch
        color = cnt >= 0.f ? color / (3.f * cnt) : 0.f;
        if (cnt == 0.f && in && threadIdx.x == 0 && blockIdx.x == 0 && (A.rgb[0] || A.rgb[1])) color = __builtin_nanf("");
    }

    // ---- MirrorMaskLoss (losses.py:186-198)
    float mask_l = 0.f;
    for (int typ = 0; typ < 2; ++typ) {
        if (!A.mirror_mask[typ]) continue;
        float g = 0.f;
        if (in && A.gt_mask && (A.flags & MNRF_LOSS_USE_MASK)) {
            const float m = A.mirror_mask[typ][i];           // possibly thresholded above (same thread wrote it)
            const float p = fminf(fmaxf(m, 1e-7f), 1.f - 1e-7f);
            const bool clampl = !(A.flags & MNRF_LOSS_TCNN_BCE);
            const float valid = gt >= 0.f ? 1.f : 0.f;
            const float l = -(gt * clamped_log(p, clampl) + (1.f - gt) * clamped_log(1.f - p, clampl));
            mask_l += l * valid / n;
            if (m >= 1e-7f && m <= 1.f - 1e-7f)               // clamp passes the gradient inside [min, max]
                g = A.w_mask * valid * (-(gt / p) + (1.f - gt) / (1.f - p)) / n;
        }
        if (in && A.g_mirror_mask[typ]) A.g_mirror_mask[typ][i] = g;
    }

    // ---- NormalLoss (losses.py:60-78)
    float normal_l = 0.f;
    for (int typ = 0; typ < 2; ++typ) {
        if (!A.normal_dif[typ]) continue;
        float g = 0.f;
        if (in && (A.flags & MNRF_LOSS_USE_NORMAL)) {
            const float v = A.normal_dif[typ][i];
            if (A.gt_mask && !any_invalid) {
                if (gt != 0.f) { g = 100.f / n_mirror; }
                else if (!(A.flags & MNRF_LOSS_NORMAL_ONLY_INSIDE_MIRROR)) { g = 1.f / (n - n_mirror); }
            } else {
                g = 1.f / n;
            }
            normal_l += v * g;
            g *= A.w_normal;
        }
        if (in && A.g_normal_dif[typ]) A.g_normal_dif[typ][i] = g;
    }
    // an empty side of the mask is mean([]) = nan in the reference
    if ((A.flags & MNRF_LOSS_USE_NORMAL) && A.gt_mask && !any_invalid && threadIdx.x == 0 && blockIdx.x == 0 &&
        (A.normal_dif[0] || A.normal_dif[1]) &&
        (n_mirror == 0.f || (n_mirror == n && !(A.flags & MNRF_LOSS_NORMAL_ONLY_INSIDE_MIRROR))))
        normal_l = __builtin_nanf("");

    // the plane loss scatters into g_x_surface with atomics: clear it here (same stream, earlier kernel)
    for (int typ = 0; typ < 2; ++typ)
        if (in && A.g_x_surface[typ])
            for (int c = 0; c < 3; ++c) A.g_x_surface[typ][i * 3 + c] = 0.f;

    const float s0 = block_sum(color, sh), s1 = block_sum(mask_l, sh), s2 = block_sum(normal_l, sh);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = s0;
        partials[blockIdx.x * 3 + 1] = s1;
        partials[blockIdx.x * 3 + 2] = s2;
    }
}

// NormalRegLoss (losses.py:143-172): relu(n * d).sum(-1) * w, mean over valid rays x samples
__global__ void sample_kernel(MnrfLossArgs A, int typ, const float* counts, float* partials) {
    __shared__ float sh[TPB / 64];
    const int S = A.n_samples[typ];
    const long long j = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool in = j < A.n_rays * S;
    const long long ray = in ? j / S : 0;
    const bool valid = in && (A.valid_mask ? A.valid_mask[ray] != 0 : true);
    const float inv = 1.f / (counts[C_VALID] * (float)S);
    const float d[3] = {A.rays[ray * 8 + 3], A.rays[ray * 8 + 4], A.rays[ray * 8 + 5]};
    const float w = in ? A.weights[typ][j] : 0.f;
    float total = 0.f, gw = 0.f;
    const float* srcs[2] = {A.pred_normal[typ], (typ == 1 && (A.flags & MNRF_LOSS_EXT_GRAD_NORMAL)) ? A.normal_fine : nullptr};
    float* dsts[2] = {A.g_pred_normal[typ], typ == 1 ? A.g_normal_fine : nullptr};
    for (int k = 0; k < 2; ++k) {
        if (!srcs[k]) {
            if (in && dsts[k])
                for (int c = 0; c < 3; ++c) dsts[k][j * 3 + c] = 0.f;
            continue;
        }
        float s = 0.f, g[3] = {0.f, 0.f, 0.f};
        if (valid) {
            for (int c = 0; c < 3; ++c) {
                const float p = srcs[k][j * 3 + c] * d[c];
                if (p > 0.f) {
                    s += p;
                    g[c] = A.w_normal_reg * w * d[c] * inv;
                }
            }
            total += s * w * inv;
            gw += s;
        }
        if (in && dsts[k])
            for (int c = 0; c < 3; ++c) dsts[k][j * 3 + c] = g[c];
    }
    if (in && A.g_weights[typ]) A.g_weights[typ][j] = A.w_normal_reg * gw * inv;
    if (counts[C_VALID] == 0.f && threadIdx.x == 0 && blockIdx.x == 0) total = __builtin_nanf("");   // mean of an empty selection
    const float s = block_sum(total, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// PlaneConsistentLoss (losses.py:88-110): one thread per drawn quadruple of mirror-mask rows
// DEVICE: MNRF_LOSS_PLANE_ON_DEVICE -- the launch is sized for plane_cap quadruples, the live ones (M // 4) and M itself come from
// the count launch's words; a draw is floor(u * M) (u < 1 in fp32; the product can still round up to M: clamped).
template <bool DEVICE>
__global__ void plane_kernel(MnrfLossArgs A, int typ, const float* counts, const int* mirror_rows, float* partials) {
    __shared__ float sh[TPB / 64];
    const long long q = (long long)blockIdx.x * TPB + threadIdx.x;
    const int m = (int)counts[C_MIRROR];
    const long long times = DEVICE ? (counts[C_INVALID] > 0.f ? 0 : (long long)(m / 4)) : A.plane_times[typ];
    float v = 0.f;
    if (q < times) {
        int r[4];
        float p[4][3];
        for (int k = 0; k < 4; ++k) {
            long long pick;
            if (DEVICE) {
                const int d = (int)(A.plane_u[typ][q * 4 + k] * (float)m);
                pick = d < m - 1 ? d : m - 1;
            } else {
                pick = A.plane_idx[typ][q * 4 + k];
            }
            r[k] = mirror_rows[pick];
            for (int c = 0; c < 3; ++c) p[k][c] = A.x_surface[typ][(long long)r[k] * 3 + c];
        }
        float a[3], b[3], c3[3];
        for (int c = 0; c < 3; ++c) { a[c] = p[1][c] - p[0][c]; b[c] = p[2][c] - p[0][c]; c3[c] = p[3][c] - p[0][c]; }
        const float ab[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const float t = ab[0] * c3[0] + ab[1] * c3[1] + ab[2] * c3[2];
        v = fabsf(t) / (float)times;
        if (A.g_x_surface[typ] && t != 0.f) {
            const float sgn = (t > 0.f ? 1.f : -1.f) * A.w_plane / (float)times;
            const float bc[3] = {b[1] * c3[2] - b[2] * c3[1], b[2] * c3[0] - b[0] * c3[2], b[0] * c3[1] - b[1] * c3[0]};
            const float ca[3] = {c3[1] * a[2] - c3[2] * a[1], c3[2] * a[0] - c3[0] * a[2], c3[0] * a[1] - c3[1] * a[0]};
            float* g = A.g_x_surface[typ];
            for (int c = 0; c < 3; ++c) {
                atomicAdd(&g[(long long)r[1] * 3 + c], sgn * bc[c]);     // d t / d a
                atomicAdd(&g[(long long)r[2] * 3 + c], sgn * ca[c]);     // d t / d b
                atomicAdd(&g[(long long)r[3] * 3 + c], sgn * ab[c]);     // d t / d c
                atomicAdd(&g[(long long)r[0] * 3 + c], -sgn * (bc[c] + ca[c] + ab[c]));
            }
        }
    }
    const float s = block_sum(v, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

struct FinishArgs {
    const float* ray_partials; int ray_blocks;
    const float* reg_partials[2]; int reg_blocks[2];
    const float* plane_partials[2]; int plane_blocks[2];
    float w[N_TERMS];
    unsigned flags;
    float* out;
};

// One workgroup, a FIXED summation order (thread t sums entries t, t + 256, ...; then a fixed tree over the 256 partial sums): the
// same bits run to run.  (It was one thread walking ~800 partials one dependent load after the other: 37 us per TotalLoss step.)
__device__ __forceinline__ float fixed_sum(const float* p, int n, int stride, int off, float* sh) {
    float v = 0.f;
    for (int b = threadIdx.x; b < n; b += TPB) v += p[b * stride + off];
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(TPB) void finish_kernel(FinishArgs F) {
    __shared__ float sh[TPB];
    float t[N_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
    t[T_COLOR] = fixed_sum(F.ray_partials, F.ray_blocks, 3, 0, sh);
    t[T_MASK] = fixed_sum(F.ray_partials, F.ray_blocks, 3, 1, sh);
    t[T_NORMAL] = fixed_sum(F.ray_partials, F.ray_blocks, 3, 2, sh);
    for (int typ = 0; typ < 2; ++typ) {
        if (F.reg_blocks[typ] > 0) t[T_REG] += fixed_sum(F.reg_partials[typ], F.reg_blocks[typ], 1, 0, sh);
        if (F.plane_blocks[typ] > 0) t[T_PLANE] += fixed_sum(F.plane_partials[typ], F.plane_blocks[typ], 1, 0, sh);
    }
    if (threadIdx.x != 0) return;
    // TotalLoss.forward: sum(list(loss_dict.values())) in insertion order, absent terms skipped (losses.py:226-253)
    const bool use[N_TERMS] = {true, (F.flags & MNRF_LOSS_USE_MASK) != 0, (F.flags & MNRF_LOSS_USE_PLANE) != 0,
                               (F.flags & MNRF_LOSS_USE_NORMAL) != 0, (F.flags & MNRF_LOSS_USE_NORMAL) != 0};
    float total = 0.f;
    for (int k = 0; k < N_TERMS; ++k) {
        t[k] *= F.w[k];
        F.out[k] = use[k] ? t[k] : 0.f;
        if (use[k]) total += t[k];
    }
    F.out[N_TERMS] = total;
}

// metrics.py:5-15: mse = mean((pred - gt)^2 [mask]), psnr = -10 log10(mse)
__global__ void mse_partial_kernel(const float* a, const float* b, const unsigned char* mask, long long n, int per_mask,
                                   float* partials) {
    __shared__ float sh[TPB / 64];
    float s = 0.f, c = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        if (!mask || mask[i / per_mask]) {
            const float d = a[i] - b[i];
            s += d * d;
            c += 1.f;
        }
    }
    const float bs = block_sum(s, sh), bc = block_sum(c, sh);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = bs;
        partials[2 * blockIdx.x + 1] = bc;
    }
}

__global__ void mse_finish_kernel(const float* partials, int blocks, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f, c = 0.f;
    for (int b = 0; b < blocks; ++b) { s += partials[2 * b]; c += partials[2 * b + 1]; }
    const float m = s / c;                 // an empty selection is 0/0 = nan, like torch.mean
    out[0] = m;
    out[1] = -10.f * log10f(m);
    out[2] = c;
}

}  // namespace

extern "C" int mnrf_mse_blocks(void) { return 1024; }

extern "C" int mnrf_mse_psnr(const float* pred, const float* gt, const unsigned char* mask, int64_t n, int per_mask,
                             float* partials, float* out, void* stream) {
    if (!pred || !gt || !partials || !out) return mnrf_fail(MNRF_ERR_ARG, "mnrf_mse_psnr: null pointer");
    if (n < 0 || per_mask < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_mse_psnr: bad size");
    int64_t blocks = (n + TPB - 1) / TPB;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, pred, gt, mask,
                       (long long)n, per_mask, partials);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, (int)blocks, out);
    return mnrf_check_launch("mnrf_mse_psnr");
}

extern "C" int64_t mnrf_loss_workspace_floats(int64_t n_rays, int n_samples_coarse, int n_samples_fine, int64_t plane_times) {
    const int64_t rb = (n_rays + TPB - 1) / TPB;
    const int64_t sb = (n_rays * (int64_t)(n_samples_coarse > n_samples_fine ? n_samples_coarse : n_samples_fine) + TPB - 1) / TPB;
    const int64_t pb = (plane_times + TPB - 1) / TPB;
    return N_COUNTS + n_rays /* mirror rows (int32) */ + 3 * rb + 2 * sb + 2 * pb + 16;
}

extern "C" int mnrf_total_loss(const MnrfLossArgs* args, float* workspace, void* stream) {
    if (!args || !workspace) return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: null pointer");
    MnrfLossArgs A = *args;
    if (A.n_rays <= 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: n_rays must be positive");
    if (!A.out || !A.targets || !A.rays) return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: out / targets / rays are required");
    for (int typ = 0; typ < 2; ++typ) {
        if ((A.pred_normal[typ] || (typ == 1 && A.normal_fine)) && (!A.weights[typ] || A.n_samples[typ] < 1))
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: per-sample normals need weights and n_samples");
        if (A.plane_times[typ] > 0 && (!A.plane_idx[typ] || !A.x_surface[typ] || !A.gt_mask))
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: plane loss needs indices, x_surface and the GT mask");
        if ((A.flags & MNRF_LOSS_PLANE_ON_DEVICE) && A.plane_u[typ] &&
            (!A.x_surface[typ] || !A.gt_mask || A.plane_cap[typ] < A.n_rays / 4))
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: plane loss on the device needs x_surface, the GT mask and n_rays / 4 quadruples of draws");
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = A.n_rays;
    const int rb = (int)((n + TPB - 1) / TPB);
    float* counts = workspace;
    int* mirror_rows = (int*)(workspace + N_COUNTS);
    float* ray_partials = workspace + N_COUNTS + n;
    float* cursor = ray_partials + 3 * (int64_t)rb;
    hipLaunchKernelGGL(count_kernel, dim3(1), dim3(1024), 0, s, A, counts, mirror_rows);
    hipLaunchKernelGGL(ray_kernel, dim3(rb), dim3(TPB), 0, s, A, counts, ray_partials);
    FinishArgs F{};
    F.ray_partials = ray_partials;
    F.ray_blocks = rb;
    for (int typ = 0; typ < 2; ++typ) {
        const bool reg = (A.flags & MNRF_LOSS_USE_NORMAL) && A.weights[typ] &&
                         (A.pred_normal[typ] || (typ == 1 && A.normal_fine && (A.flags & MNRF_LOSS_EXT_GRAD_NORMAL)));
        if (reg) {
            const int64_t blocks = (n * A.n_samples[typ] + TPB - 1) / TPB;
            if (blocks > 0x7fffffff) return mnrf_fail(MNRF_ERR_ARG, "mnrf_total_loss: too many samples");
            hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(TPB), 0, s, A, typ, counts, cursor);
            F.reg_partials[typ] = cursor;
            F.reg_blocks[typ] = (int)blocks;
            cursor += blocks;
        } else {
            // term switched off (or no per-sample inputs): the gradients of the per-sample tensors are zero
            const size_t per = (size_t)n * (size_t)(A.n_samples[typ] > 0 ? A.n_samples[typ] : 0) * sizeof(float);
            if (per) mnrf::zero_fill(s, A.g_pred_normal[typ], per * 3, A.g_weights[typ], per, typ == 1 ? A.g_normal_fine : nullptr, per * 3);
        }
    }
    // losses.py:124: "fine" first
    for (int typ = 1; typ >= 0; --typ) {
        const bool on_device = (A.flags & MNRF_LOSS_PLANE_ON_DEVICE) && A.plane_u[typ] && n / 4 > 0;
        if ((A.flags & MNRF_LOSS_USE_PLANE) && (on_device || A.plane_times[typ] > 0)) {
            const int64_t blocks = ((on_device ? n / 4 : A.plane_times[typ]) + TPB - 1) / TPB;
            if (on_device)
                hipLaunchKernelGGL(plane_kernel<true>, dim3((unsigned)blocks), dim3(TPB), 0, s, A, typ, counts, mirror_rows, cursor);
            else
                hipLaunchKernelGGL(plane_kernel<false>, dim3((unsigned)blocks), dim3(TPB), 0, s, A, typ, counts, mirror_rows, cursor);
            F.plane_partials[typ] = cursor;
            F.plane_blocks[typ] = (int)blocks;
            cursor += blocks;
        }
    }
    F.w[T_COLOR] = A.w_color; F.w[T_MASK] = A.w_mask; F.w[T_PLANE] = A.w_plane; F.w[T_NORMAL] = A.w_normal; F.w[T_REG] = A.w_normal_reg;
    F.flags = A.flags;
    F.out = A.out;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(TPB), 0, s, F);
    return mnrf_check_launch("mnrf_total_loss");
}

// number of mirror rows by the GT mask (what PlaneConsistentLoss draws its indices from); needs the counts of a
// previous mnrf_loss_count on the same workspace
extern "C" int mnrf_loss_count(const MnrfLossArgs* args, float* workspace, void* stream) {
    if (!args || !workspace) return mnrf_fail(MNRF_ERR_ARG, "mnrf_loss_count: null pointer");
    MnrfLossArgs A = *args;
    if (A.n_rays <= 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_loss_count: n_rays must be positive");
    hipLaunchKernelGGL(count_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A, workspace, (int*)nullptr);
    return mnrf_check_launch("mnrf_loss_count");
}
