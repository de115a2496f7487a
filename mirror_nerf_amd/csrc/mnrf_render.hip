// mnrf_render.hip -- the ray-side kernels of the rendering hot path for gfx950:
// view/position embedding, coarse depth sampling, alpha compositing with a wavefront
// prefix product, hierarchical resampling (inverse CDF + sort), mirror-mask thresholding,
// reflected-ray construction with order-preserving compaction, blend/scatter, pin-hole
// ray generation.  All of them are HBM-bound streaming kernels (a few dozen bytes per
// sample); they exist so that nothing on the path runs as a chain of framework ops.
// Compiled with -ffp-contract=off (separate multiply and add as in the reference).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"
#include "mnrf_fill.h"

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";

int mnrf_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int mnrf_check_launch(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
        return MNRF_ERR_LAUNCH;
    }
    return MNRF_OK;
}

extern "C" const char* mnrf_last_error(void) { return g_err; }
extern "C" int mnrf_version(void) { return 1; }

namespace {

constexpr float EPS32 = 1.1920928955078125e-07f;  // torch.finfo(float32).eps, utils/func.py:5

// Live row count of the `_n` entry points (include/mnrf.h "live row counts on the device"): the launch is sized for the capacity
// `cap`, *n_live (device, written by an earlier launch on the stream: mnrf_reflect_compact's count) says how many rows exist.
__device__ __forceinline__ long long live_rows(long long cap, const int* __restrict__ n_live) {
    if (!n_live) return cap;
    const long long v = *n_live;
    return v < 0 ? 0 : (v < cap ? v : cap);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
#include "mnrf_composite.inc"

// ------------------------------------------------------------------ Embedding.forward
// models/mirror_nerf.py:20-38: out = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(N-1) x), cos(2^(N-1) x)]
__global__ void embed_kernel(const float* __restrict__ x, long long n, int c, int n_freqs, float* __restrict__ out,
                             const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    const long long row = i / c;
    const int ch = (int)(i % c);
    const int ld = c * (2 * n_freqs + 1);
    const float v = x[i];
    float* o = out + row * ld;
    o[ch] = v;
    for (int f = 0; f < n_freqs; ++f) {
        float s, co;
        sincosf(ldexpf(v, f), &s, &co);
        o[c * (1 + 2 * f) + ch] = s;
        o[c * (2 + 2 * f) + ch] = co;
    }
}

// ------------------------------------------------------------------ coarse depths
// models/rendering.py:283-300
__global__ void sample_coarse_kernel(const float* __restrict__ rays, long long n_rays, const float* __restrict__ z_steps,
                                     int ns, int use_disp, float perturb, const float* __restrict__ prand,
                                     float* __restrict__ z_out, const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n_rays, n_live) * ns) return;
    const long long r = i / ns;
    const int s = (int)(i % ns);
    const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
    auto zat = [&](int k) -> float {
        const float t = z_steps[k];
        if (!use_disp) return near * (1.f - t) + far * t;
        return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
    };
    float z = zat(s);
    if (perturb > 0.f) {
        const float lower = s == 0 ? z : 0.5f * (zat(s - 1) + z);
        const float upper = s == ns - 1 ? z : 0.5f * (z + zat(s + 1));
        z = lower + (upper - lower) * (perturb * prand[i]);
    }
    z_out[i] = z;
}

// ------------------------------------------------------------------ alpha compositing
// models/rendering.py:181-264, 362-367.  One wavefront per ray; the transmittance
// T_i = prod_{j<i} (1 - alpha_j + 1e-10) is an exclusive prefix product over the lanes
// (Hillis-Steele over __shfl_up), carried across 64-sample blocks.
struct CompArgs {
    const float* rays; long long n_rays; int S;
    const float* sigma; const float* z; const float* noise; const float* rgb; const float* is_mirror;
    const float* pred_normal; const float* normal; int white_back;
    float* weights; float* opacity; float* rgb_map; float* depth; float* mirror_mask;
    float* surf_normal; float* surf_normal_grad; float* normal_dif; float* x_surface;
    const int* n_live;
};

// per-sample inputs of one ray straight from the flat (n_rays * S) tensors in HBM
struct CompGlobalSrc {
    const CompArgs& A;
    long long ray;
    __device__ __forceinline__ bool has_noise() const { return A.noise != nullptr; }
    __device__ __forceinline__ bool has_rgb() const { return A.rgb != nullptr; }
    __device__ __forceinline__ bool has_mirror() const { return A.is_mirror != nullptr; }
    __device__ __forceinline__ bool has_pn() const { return A.pred_normal != nullptr; }
    __device__ __forceinline__ bool has_gn() const { return A.normal != nullptr; }
    __device__ __forceinline__ float z(int s) const { return A.z[ray * A.S + s]; }
    __device__ __forceinline__ float sigma(int s) const { return A.sigma[ray * A.S + s]; }
    __device__ __forceinline__ float noise(int s) const { return A.noise[ray * A.S + s]; }
    __device__ __forceinline__ float rgb(int s, int k) const { return A.rgb[(ray * A.S + s) * 3 + k]; }
    __device__ __forceinline__ float mirror(int s) const { return A.is_mirror[ray * A.S + s]; }
    __device__ __forceinline__ float pn(int s, int k) const { return A.pred_normal[(ray * A.S + s) * 3 + k]; }
    __device__ __forceinline__ float gn(int s, int k) const { return A.normal[(ray * A.S + s) * 3 + k]; }
};

__global__ __launch_bounds__(256) void composite_kernel(CompArgs A) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= live_rows(A.n_rays, A.n_live)) return;
    const CompGlobalSrc src{A, ray};
    const CompMaps out{A.weights ? A.weights + ray * A.S : nullptr, A.opacity, A.rgb_map, A.depth, A.mirror_mask, A.surf_normal,
                       A.surf_normal_grad, A.normal_dif, A.x_surface, A.rays, A.white_back};
    composite_ray(src, A.S, lane, ray, out);
}


// ------------------------------------------------------------------ compositing backward
// w_i = alpha_i T_i, T_i = prod_{j<i} t_j, t_j = 1 - alpha_j + 1e-10.  With G_i = dL/dw_i (direct +
// through every composited output):  dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / t_i, and
// dalpha/dsigma = delta (1 - alpha) [sigma + noise > 0].  One wavefront per ray: T by the forward
// prefix product of the forward kernel, the suffix sum by a reverse scan over the lanes; both are
// carried across 64-sample blocks (S <= 256).
struct CompBwdArgs {
    const float* rays; long long n_rays; int S;
    const float* sigma; const float* z; const float* noise; const float* rgb; const float* is_mirror;
    const float* pred_normal; const float* normal; int white_back;
    const float* depth;
    const float* g_w; const float* g_op; const float* g_rgb; const float* g_depth; const float* g_mask;
    const float* g_sn; const float* g_sng; const float* g_nd; const float* g_xs;
    float* d_sigma; float* d_rgb; float* d_mirror; float* d_pn; float* d_n; float* d_rays;
    // gradient steering (models/rendering.py:222-264): which composited outputs see the weights as constants
    int detach;                 // MNRF_DETACH_W_MASK: mirror mask of every ray; MNRF_DETACH_W_NORMAL: the three normal outputs
    const float* keep_mirror;   // (n_rays) or null: rays whose entry is 0 see detached weights in the mirror mask
    const int* n_live;
};
constexpr int CB_MAXB = 4;

__global__ __launch_bounds__(256) void composite_backward_kernel(CompBwdArgs A) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= live_rows(A.n_rays, A.n_live)) return;
    const int S = A.S;
    const float* r8 = A.rays ? A.rays + ray * 8 : nullptr;
    float gop = A.g_op ? A.g_op[ray] : 0.f;
    float gc[3] = {0.f, 0.f, 0.f}, gsn[3] = {0.f, 0.f, 0.f}, gsg[3] = {0.f, 0.f, 0.f}, gxs[3] = {0.f, 0.f, 0.f};
    if (A.g_rgb) { for (int k = 0; k < 3; ++k) gc[k] = A.g_rgb[ray * 3 + k]; }
    if (A.g_sn) { for (int k = 0; k < 3; ++k) gsn[k] = A.g_sn[ray * 3 + k]; }
    if (A.g_sng) { for (int k = 0; k < 3; ++k) gsg[k] = A.g_sng[ray * 3 + k]; }
    if (A.g_xs) { for (int k = 0; k < 3; ++k) gxs[k] = A.g_xs[ray * 3 + k]; }
    float gd = A.g_depth ? A.g_depth[ray] : 0.f;
    const float gm = A.g_mask ? A.g_mask[ray] : 0.f;
    const float gnd = A.g_nd ? A.g_nd[ray] : 0.f;
    // the same cotangents as seen by the WEIGHTS (zero where the reference multiplies weights.detach())
    const bool w_mask = !(A.detach & MNRF_DETACH_W_MASK) && !(A.keep_mirror && A.keep_mirror[ray] == 0.f);
    const bool w_nrm = !(A.detach & MNRF_DETACH_W_NORMAL);
    const float gm_w = w_mask ? gm : 0.f;
    const float gnd_w = w_nrm ? gnd : 0.f;
    float gsn_w[3], gsg_w[3];
    for (int k = 0; k < 3; ++k) { gsn_w[k] = w_nrm ? gsn[k] : 0.f; gsg_w[k] = w_nrm ? gsg[k] : 0.f; }
    if (A.g_xs && r8) gd += gxs[0] * r8[3] + gxs[1] * r8[4] + gxs[2] * r8[5];   // x_surface = o + d * depth
    if (A.white_back) gop -= gc[0] + gc[1] + gc[2];                            // rgb_map += 1 - opacity
    if (A.d_rays && lane == 0) {
        float* o = A.d_rays + ray * 8;
        const float dep = A.depth ? A.depth[ray] : 0.f;
        for (int k = 0; k < 3; ++k) { o[k] = gxs[k]; o[3 + k] = gxs[k] * dep; }
        o[6] = 0.f; o[7] = 0.f;
    }
    // forward sweep: alpha, T, w, G per sample (kept in registers: <= 4 blocks per lane)
    float Tn[CB_MAXB], Gw[CB_MAXB], Gv[CB_MAXB], tv[CB_MAXB], dads[CB_MAXB];
    float carry = 1.f;
#pragma unroll
    for (int blk = 0; blk < CB_MAXB; ++blk) {
        Tn[blk] = 0.f; Gw[blk] = 0.f; Gv[blk] = 0.f; tv[blk] = 1.f; dads[blk] = 0.f;
        if (blk * 64 >= S) continue;
        const int s = blk * 64 + lane;
        const bool in = s < S;
        const long long i1 = ray * S + s, i3 = i1 * 3;
        float alpha = 0.f, G = 0.f, delta = 0.f, sv = 0.f;
        float pn[3] = {0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};
        if (in) {
            const float zv = A.z[i1];
            delta = s + 1 < S ? A.z[i1 + 1] - zv : 1e10f;
            sv = A.sigma[i1];
            if (A.noise) sv = sv + A.noise[i1];
            alpha = 1.f - expf(-delta * fmaxf(sv, 0.f));
            G = (A.g_w ? A.g_w[i1] : 0.f) + gop + gd * zv;
            if (A.rgb) G += gc[0] * A.rgb[i3] + gc[1] * A.rgb[i3 + 1] + gc[2] * A.rgb[i3 + 2];
            if (A.is_mirror) G += gm_w * A.is_mirror[i1];
            if (A.pred_normal) { for (int k = 0; k < 3; ++k) { pn[k] = A.pred_normal[i3 + k]; G += gsn_w[k] * pn[k]; } }
            if (A.normal) { for (int k = 0; k < 3; ++k) { gn[k] = A.normal[i3 + k]; G += gsg_w[k] * gn[k]; } }
            if (A.pred_normal && A.normal) {
                const float d0 = gn[0] - pn[0], d1 = gn[1] - pn[1], d2 = gn[2] - pn[2];
                G += gnd_w * (d0 * d0 + d1 * d1 + d2 * d2);
            }
        }
        const float t = in ? (1.f - alpha) + 1e-10f : 1.f;
        float incl = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o);
            if (lane >= o) incl *= up;
        }
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63);
        const float w = alpha * T;
        Tn[blk] = T; Gv[blk] = G; Gw[blk] = in ? G * w : 0.f; tv[blk] = t;
        dads[blk] = (in && sv > 0.f) ? delta * (1.f - alpha) : 0.f;   // relu'(0) = 0 as in torch
        if (in) {   // gradients of the per-sample inputs that enter linearly
            if (A.d_rgb) { for (int k = 0; k < 3; ++k) A.d_rgb[i3 + k] = w * gc[k]; }
            if (A.d_mirror) A.d_mirror[i1] = w * gm;
            if (A.d_pn) { for (int k = 0; k < 3; ++k) A.d_pn[i3 + k] = w * (gsn[k] - 2.f * gnd * (gn[k] - pn[k])); }
            if (A.d_n) { for (int k = 0; k < 3; ++k) A.d_n[i3 + k] = w * (gsg[k] + 2.f * gnd * (gn[k] - pn[k])); }
        }
    }
    // backward sweep: suffix sums of G*w
    float tail = 0.f;
#pragma unroll
    for (int blk = CB_MAXB - 1; blk >= 0; --blk) {
        if (blk * 64 >= S) continue;
        float suf = Gw[blk];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float dn = __shfl_down(suf, o);
            if (lane + o < 64) suf += dn;
        }
        const float total = __shfl(suf, 0);
        float after = __shfl_down(suf, 1);
        if (lane == 63) after = 0.f;
        after += tail;
        tail += total;
        const int s = blk * 64 + lane;
        if (s < S && A.d_sigma) A.d_sigma[ray * S + s] = (Gv[blk] * Tn[blk] - after / tv[blk]) * dads[blk];
    }
}

// ------------------------------------------------------------------ hierarchical resampling
// models/rendering.py:7-51 (sample_pdf) + 312-326 (merge and sort).  One wavefront per ray.
// cdf is accumulated in double and rounded per entry, as ATen's CPU cumsum does.
constexpr int SF_MAX = 512;   // S + n_importance <= 512, S <= 256

// ascending bitonic sort of 64*NE values held as v[e] = element (lane + 64 e); compare-exchange semantics of the LDS version
// (lower index gets the smaller value in an ascending block)
template <int NE>
__device__ __forceinline__ void sort_store(const float* srt, int lane, float* out, int T) {
    float v[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = srt[lane + 64 * e];
#pragma unroll
    for (int k = 2; k <= 64 * NE; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            if (jj >= 64) {
                const int je = jj / 64;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if ((e ^ je) > e) {
                        const bool up = ((lane + 64 * e) & k) == 0;
                        const float a = v[e], b = v[e ^ je];
                        const bool sw = (a > b) == up;
                        v[e] = sw ? b : a;
                        v[e ^ je] = sw ? a : b;
                    }
                }
            } else {
                const bool lower = (lane & jj) == 0;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const bool up = ((lane + 64 * e) & k) == 0;
                    const float mine = v[e];
                    const float other = __shfl_xor(mine, jj);
                    // the pair (lo, hi): lo keeps min when ascending; "a > b swaps" leaves equal values in place
                    const float lo = lower ? mine : other, hi = lower ? other : mine;
                    const bool sw = (lo > hi) == up;
                    v[e] = lower ? (sw ? hi : lo) : (sw ? lo : hi);
                }
            }
        }
    }
    if (out) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (lane + 64 * e < T) out[lane + 64 * e] = v[e];
    }
}

// body of the resampling for the ray of this wavefront (`ray` < n_rays; live = false: a wave past the last ray, which runs along on
// the last ray for the barriers and stores nothing).  All four waves of the workgroup must call it.
__device__ __forceinline__ void sample_fine_ray(const float* __restrict__ z_coarse, const float* __restrict__ weights, long long ray,
                                                bool live, int S, const float* __restrict__ u, int u_per_ray, int n_imp,
                                                float* __restrict__ z_fine) {
    __shared__ float s_cdf[4][256];
    __shared__ float s_bin[4][256];
    __shared__ float s_sort[4][SF_MAX];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const float* zc = z_coarse + ray * S;
    const float* wc = weights + ray * S;
    const int nw = S - 2;          // weights[:, 1:-1]
    const int nbins = S - 1;       // mid-points
    float* cdf = s_cdf[wv];
    float* bin = s_bin[wv];
    float* srt = s_sort[wv];
    const float eps = 1e-5f;
    // mid-points and the weight sum
    float part = 0.f;
    for (int i = lane; i < nbins; i += 64) bin[i] = 0.5f * (zc[i] + zc[i + 1]);
    for (int i = lane; i < nw; i += 64) part += wc[1 + i] + eps;
    const float total = wave_sum(part);
    // cdf[0] = 0, cdf[i+1] = cumsum(pdf)[i]
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.f;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + lane;
        double v = i < nw ? (double)((wc[1 + i] + eps) / total) : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(v, o);
            if (lane >= o) v += up;
        }
        if (i < nw) cdf[i + 1] = (float)(carry + v);
        carry += __shfl(v, 63);
    }
    // coarse depths go straight into the sort buffer
    const int T = S + n_imp;
    int P2 = 1;
    while (P2 < T) P2 <<= 1;
    for (int i = lane; i < S; i += 64) srt[i] = zc[i];
    for (int i = T + lane; i < (P2 < 64 ? 64 : P2); i += 64) srt[i] = __builtin_inff();      // the sort works on >= 64 values
    __syncthreads();
    // inverse CDF
    for (int j = lane; j < n_imp; j += 64) {
        const float uj = u_per_ray ? u[ray * n_imp + j] : u[j];
        // searchsorted(cdf, u, right=True): number of entries <= u
        int lo = 0, hi = nbins;    // cdf has nbins entries
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nw ? lo : nw;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bin[below], b1 = bin[above];
        float denom = c1 - c0;
        if (denom < eps) denom = 1.f;
        srt[S + j] = b0 + (uj - c0) / denom * (b1 - b0);
    }
    __syncthreads();
    // bitonic sort of P2 values, in REGISTERS: lane l holds elements l, l + 64, ...; a partner at distance < 64 is another
    // lane's register of the same slot (one __shfl_xor), a partner at distance >= 64 another slot of the same lane.  (In
    // LDS, with a barrier per step, the 36 steps of a 256-value sort were a chain of dependent LDS round trips.)
    if (P2 <= 64) sort_store<1>(srt, lane, live ? z_fine + ray * T : nullptr, T);
    else if (P2 == 128) sort_store<2>(srt, lane, live ? z_fine + ray * T : nullptr, T);
    else if (P2 == 256) sort_store<4>(srt, lane, live ? z_fine + ray * T : nullptr, T);
    else sort_store<8>(srt, lane, live ? z_fine + ray * T : nullptr, T);
}

__global__ __launch_bounds__(256) void sample_fine_kernel(const float* __restrict__ z_coarse, const float* __restrict__ weights,
                                                          long long n_rays, int S, const float* __restrict__ u, int u_per_ray,
                                                          int n_imp, float* __restrict__ z_fine, const int* __restrict__ n_live) {
    long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    n_rays = live_rows(n_rays, n_live);
    if ((long long)blockIdx.x * 4 >= n_rays) return;      // (the whole workgroup: the barriers of the body stay uniform)
    const bool live = ray < n_rays;
    if (!live) ray = n_rays - 1;
    sample_fine_ray(z_coarse, weights, ray, live, S, u, u_per_ray, n_imp, z_fine);
}

// Compositing of a pass and the resampling that follows it (models/rendering.py:181-264, then 312-326) in ONE launch: both are
// "one wavefront per ray, four rays per workgroup", and the resampling reads nothing but the weights the same wavefront has just
// written (made visible to the workgroup by the fence + barrier).  Same bodies: the same maps, weights and depths bit for bit.
__global__ __launch_bounds__(256) void composite_sample_kernel(CompArgs A, const float* __restrict__ u, int u_per_ray, int n_imp,
                                                               float* __restrict__ z_fine) {
    const int lane = threadIdx.x & 63;
    long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_rays = live_rows(A.n_rays, A.n_live);
    if ((long long)blockIdx.x * 4 >= n_rays) return;
    const bool live = ray < n_rays;
    if (live) {
        const CompGlobalSrc src{A, ray};
        const CompMaps out{A.weights + ray * A.S, A.opacity, A.rgb_map, A.depth, A.mirror_mask, A.surf_normal,
                           A.surf_normal_grad, A.normal_dif, A.x_surface, A.rays, A.white_back};
        composite_ray(src, A.S, lane, ray, out);
    }
    __threadfence_block();
    __syncthreads();
    if (!live) ray = n_rays - 1;
    sample_fine_ray(A.z, A.weights, ray, live, A.S, u, u_per_ray, n_imp, z_fine);
}

// ------------------------------------------------------------------ mask threshold
// train.py:165-166 / eval.py:305-306: m[m>0.5]=1; m[m<0.5]=0 (exactly 0.5 stays), any(m != 0)
__global__ void threshold_kernel(float* __restrict__ m, long long n, int* __restrict__ any, const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool nz = false;
    if (i < live_rows(n, n_live)) {
        float v = m[i];
        if (v > 0.5f) v = 1.f;
        else if (v < 0.5f) v = 0.f;
        m[i] = v;
        nz = v != 0.f;
    }
    if (__ballot(nz) != 0ull && (threadIdx.x & 63) == 0 && any) atomicOr(any, 1);
}

// ------------------------------------------------------------------ reflected rays + compaction
// train.py:192-252, eval.py:336-360, 506-548.  One workgroup walks the chunk in order so
// that the compacted rows keep the order of `secondary_rays[mask]`.
struct ReflArgs {
    const float* rays; const float* x_surface; const float* normal; const float* normal_noise; float noise_std;
    const float* mask; long long n; int compact; float near2;
    float* sec; int* index; int* count; float* reflect_dir;
    const int* n_live;
    int* slot;      // (n) or null: row of sec that ray i went to, -1 for a ray that was not selected (the inverse of `index`)
};

__global__ __launch_bounds__(1024) void reflect_compact_kernel(ReflArgs A) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    A.n = live_rows(A.n, A.n_live);
    for (long long start = 0; start < A.n; start += 1024) {
        const long long i = start + tid;
        const bool in = i < A.n;
        float r[3] = {0.f, 0.f, 0.f};
        bool sel = false;
        if (in) {
            float nv[3], wvv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                nv[k] = A.normal[i * 3 + k];
                if (A.normal_noise) nv[k] = nv[k] + A.normal_noise[i * 3 + k] * A.noise_std;   // eval.py:506-511
                wvv[k] = -A.rays[i * 8 + 3 + k];
            }
            // l2_normalize (utils/func.py:5-7) of both, then r = 2 (w.n) n - w
            const float ninv = 1.f / sqrtf(fmaxf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2], EPS32));
            const float winv = 1.f / sqrtf(fmaxf(wvv[0] * wvv[0] + wvv[1] * wvv[1] + wvv[2] * wvv[2], EPS32));
#pragma unroll
            for (int k = 0; k < 3; ++k) { nv[k] = nv[k] * ninv; wvv[k] = wvv[k] * winv; }
            const float c = wvv[0] * nv[0] + wvv[1] * nv[1] + wvv[2] * nv[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] = 2.f * c * nv[k] - wvv[k];
            if (A.reflect_dir) { for (int k = 0; k < 3; ++k) A.reflect_dir[i * 3 + k] = r[k]; }
            sel = A.compact ? (A.mask[i] != 0.f) : true;
        }
        const unsigned long long bal = __ballot(sel);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { const int c = s_wave[k]; if (k < wv) woff += c; tot += c; }
        const int base = s_base;
        if (sel) {
            const long long p = base + woff + before;
            float* o = A.sec + p * 8;
            o[0] = A.x_surface[i * 3]; o[1] = A.x_surface[i * 3 + 1]; o[2] = A.x_surface[i * 3 + 2];
            o[3] = r[0]; o[4] = r[1]; o[5] = r[2];
            o[6] = A.near2;                    // ray_forward_offset, absolute (train.py:232, eval.py:529)
            o[7] = A.rays[i * 8 + 7];
            A.index[p] = (int)i;
            if (A.slot) A.slot[i] = (int)p;
        } else if (in && A.slot) {
            A.slot[i] = -1;
        }
        __syncthreads();
        if (tid == 0) s_base = base + tot;
        __syncthreads();
    }
    if (tid == 0) *A.count = s_base;
}

// Without compaction (eval level 0 traces every ray, eval.py:159) ray i lands in row i: no ordering to preserve, one thread
// per ray over the whole chip instead of one workgroup walking the chunk (158 us per 32768-ray chunk, 3 ms per frame).
__global__ __launch_bounds__(256) void reflect_all_kernel(ReflArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    A.n = live_rows(A.n, A.n_live);
    if (i == 0) *A.count = (int)A.n;
    if (i >= A.n) return;
    float nv[3], wvv[3], r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        nv[k] = A.normal[i * 3 + k];
        if (A.normal_noise) nv[k] = nv[k] + A.normal_noise[i * 3 + k] * A.noise_std;   // eval.py:506-511
        wvv[k] = -A.rays[i * 8 + 3 + k];
    }
    // the same expressions, in the same order, as reflect_compact_kernel
    const float ninv = 1.f / sqrtf(fmaxf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2], EPS32));
    const float winv = 1.f / sqrtf(fmaxf(wvv[0] * wvv[0] + wvv[1] * wvv[1] + wvv[2] * wvv[2], EPS32));
#pragma unroll
    for (int k = 0; k < 3; ++k) { nv[k] = nv[k] * ninv; wvv[k] = wvv[k] * winv; }
    const float c = wvv[0] * nv[0] + wvv[1] * nv[1] + wvv[2] * nv[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = 2.f * c * nv[k] - wvv[k];
    if (A.reflect_dir) { for (int k = 0; k < 3; ++k) A.reflect_dir[i * 3 + k] = r[k]; }
    float* o = A.sec + i * 8;
    o[0] = A.x_surface[i * 3]; o[1] = A.x_surface[i * 3 + 1]; o[2] = A.x_surface[i * 3 + 2];
    o[3] = r[0]; o[4] = r[1]; o[5] = r[2];
    o[6] = A.near2;
    o[7] = A.rays[i * 8 + 7];
    A.index[i] = (int)i;
    if (A.slot) A.slot[i] = (int)i;
}

// ------------------------------------------------------------------ blend / scatter
// train.py:261-296, eval.py:676-697
__global__ void blend_all_kernel(const float* __restrict__ base, const float* __restrict__ sec, const float* __restrict__ mask,
                                 long long n, int c, int direct, float* __restrict__ out, float* __restrict__ refl,
                                 const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    const float m = mask[i / c];
    const float b = base[i];
    const float part = direct ? sec[i] : b;   // compacted case: rows without a source keep base
    out[i] = m * part + (1.f - m) * b;
    if (refl) refl[i] = direct ? sec[i] : 0.f;
}

__global__ void blend_scatter_kernel(const float* __restrict__ base, const float* __restrict__ sec, const int* __restrict__ index,
                                     long long n_sec, const float* __restrict__ mask, int c, float* __restrict__ out,
                                     float* __restrict__ refl, const int* __restrict__ n_sec_live) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= live_rows(n_sec, n_sec_live) * c) return;
    const long long row = index[j / c];
    const int ch = (int)(j % c);
    const float m = mask[row];
    const float v = sec[j];
    out[row * c + ch] = m * v + (1.f - m) * base[row * c + ch];
    if (refl) refl[row * c + ch] = v;
}

// Both blends of a level (rgb_coarse and rgb_fine, train.py:263-296) in ONE launch, in gather form through `slot` (the inverse of the
// compaction's index): out = m * part + (1 - m) * base with part = sec[slot] where the ray was reflected, base where it was not -- the
// very expressions of blend_all_kernel / blend_scatter_kernel, so the values are identical bit for bit.  (Round 5: the scatter form
// took two launches per tensor and two more backward.)
__global__ void blend2_kernel(const float* __restrict__ base_a, const float* __restrict__ sec_a, const float* __restrict__ base_b,
                              const float* __restrict__ sec_b, const int* __restrict__ slot, const float* __restrict__ mask, long long n,
                              int c, float* __restrict__ out_a, float* __restrict__ out_b, const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    const long long row = i / c;
    const int ch = (int)(i % c);
    const float m = mask[row];
    const int sl = slot[row];
    if (base_a) {
        const float b = base_a[i];
        const float part = sl >= 0 ? sec_a[(long long)sl * c + ch] : b;
        out_a[i] = m * part + (1.f - m) * b;
    }
    if (base_b) {
        const float b = base_b[i];
        const float part = sl >= 0 ? sec_b[(long long)sl * c + ch] : b;
        out_b[i] = m * part + (1.f - m) * b;
    }
}
__global__ void blend2_backward_kernel(const float* __restrict__ g_out_a, const float* __restrict__ g_out_b, const int* __restrict__ slot,
                                       const float* __restrict__ mask, long long n, int c, float* __restrict__ g_base_a,
                                       float* __restrict__ g_sec_a, float* __restrict__ g_base_b, float* __restrict__ g_sec_b,
                                       const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    const long long row = i / c;
    const int ch = (int)(i % c);
    const float m = mask[row];
    const int sl = slot[row];
    if (g_out_a) {
        const float g = g_out_a[i];
        if (g_base_a) g_base_a[i] = (1.f - m) * g;
        if (g_sec_a && sl >= 0) g_sec_a[(long long)sl * c + ch] = m * g;
    }
    if (g_out_b) {
        const float g = g_out_b[i];
        if (g_base_b) g_base_b[i] = (1.f - m) * g;
        if (g_sec_b && sl >= 0) g_sec_b[(long long)sl * c + ch] = m * g;
    }
}

// ------------------------------------------------------------------ pin-hole rays
// datasets/ray_utils.py:6-53
struct Pose { float m[12]; };

__global__ void generate_rays_kernel(int H, int W, float focal, Pose c2w, float near, float far, float* __restrict__ rays) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)H * W) return;
    const int i = (int)(p % W), j = (int)(p / W);
    const float dx = ((float)i - (float)W / 2.f) / focal;   // no +0.5 (ray_utils.py:19-24)
    const float dy = -((float)j - (float)H / 2.f) / focal;
    const float dz = -1.f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = dx * c2w.m[r * 4] + dy * c2w.m[r * 4 + 1] + dz * c2w.m[r * 4 + 2];
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float* o = rays + p * 8;
    o[0] = c2w.m[3]; o[1] = c2w.m[7]; o[2] = c2w.m[11];
    o[3] = d[0] / nrm; o[4] = d[1] / nrm; o[5] = d[2] / nrm;
    o[6] = near; o[7] = far;
}

// ------------------------------------------------------------------ backward of the per-ray glue (training)
// reflect: r = 2 (w.n) n - w with n = l2n(normal), w = l2n(-d); secondary = [x_surface, r, near2, far]
__global__ void reflect_backward_kernel(const float* __restrict__ rays, const float* __restrict__ normal,
                                        const int* __restrict__ index, long long n_sec, const float* __restrict__ g_sec,
                                        float* __restrict__ g_xs, float* __restrict__ g_normal, float* __restrict__ g_rays,
                                        const int* __restrict__ n_sec_live) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= live_rows(n_sec, n_sec_live)) return;
    const long long i = index ? index[j] : j;
    float nv[3], wv[3], gr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { nv[k] = normal[i * 3 + k]; wv[k] = -rays[i * 8 + 3 + k]; gr[k] = g_sec[j * 8 + 3 + k]; }
    const float nsq = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
    const float wsq = wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2];
    const float ninv = 1.f / sqrtf(fmaxf(nsq, EPS32)), winv = 1.f / sqrtf(fmaxf(wsq, EPS32));
    float nh[3], wh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { nh[k] = nv[k] * ninv; wh[k] = wv[k] * winv; }
    const float c = wh[0] * nh[0] + wh[1] * nh[1] + wh[2] * nh[2];
    const float grn = gr[0] * nh[0] + gr[1] * nh[1] + gr[2] * nh[2];
    float gnh[3], gwh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gnh[k] = 2.f * grn * wh[k] + 2.f * c * gr[k]; gwh[k] = 2.f * grn * nh[k] - gr[k]; }
    const float pn = nsq > EPS32 ? nh[0] * gnh[0] + nh[1] * gnh[1] + nh[2] * gnh[2] : 0.f;   // clamped norm: constant denominator
    const float pw = wsq > EPS32 ? wh[0] * gwh[0] + wh[1] * gwh[1] + wh[2] * gwh[2] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g_xs[i * 3 + k] = g_sec[j * 8 + k];
        g_normal[i * 3 + k] = (gnh[k] - nh[k] * pn) * ninv;
        g_rays[i * 8 + 3 + k] = -(gwh[k] - wh[k] * pw) * winv;     // w = -d / |d|
    }
    g_rays[i * 8 + 7] = g_sec[j * 8 + 7];                           // far is passed through
}

// The same in GATHER form through the compaction's inverse index (slot[i] = row of sec_rays ray i went to, -1: not reflected): one
// thread per ray writes its row of all three outputs -- zeros where nothing was reflected -- so nothing has to be zero-filled in front
// (the scatter form above: a zeroing launch + this one).  Same expressions: bit-identical values.
__global__ void reflect_backward_gather_kernel(const float* __restrict__ rays, const float* __restrict__ normal,
                                               const int* __restrict__ slot, const float* __restrict__ g_sec, long long n_rays,
                                               float* __restrict__ g_xs, float* __restrict__ g_normal, float* __restrict__ g_rays,
                                               const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n_rays, n_live)) return;
    const long long j = slot[i];
    float* gx = g_xs + i * 3;
    float* gn = g_normal + i * 3;
    float* gy = g_rays + i * 8;
    if (j < 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { gx[k] = 0.f; gn[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) gy[k] = 0.f;
        return;
    }
    float nv[3], wv[3], gr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { nv[k] = normal[i * 3 + k]; wv[k] = -rays[i * 8 + 3 + k]; gr[k] = g_sec[j * 8 + 3 + k]; }
    const float nsq = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
    const float wsq = wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2];
    const float ninv = 1.f / sqrtf(fmaxf(nsq, EPS32)), winv = 1.f / sqrtf(fmaxf(wsq, EPS32));
    float nh[3], wh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { nh[k] = nv[k] * ninv; wh[k] = wv[k] * winv; }
    const float c = wh[0] * nh[0] + wh[1] * nh[1] + wh[2] * nh[2];
    const float grn = gr[0] * nh[0] + gr[1] * nh[1] + gr[2] * nh[2];
    float gnh[3], gwh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gnh[k] = 2.f * grn * wh[k] + 2.f * c * gr[k]; gwh[k] = 2.f * grn * nh[k] - gr[k]; }
    const float pn = nsq > EPS32 ? nh[0] * gnh[0] + nh[1] * gnh[1] + nh[2] * gnh[2] : 0.f;
    const float pw = wsq > EPS32 ? wh[0] * gwh[0] + wh[1] * gwh[1] + wh[2] * gwh[2] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gx[k] = g_sec[j * 8 + k];
        gn[k] = (gnh[k] - nh[k] * pn) * ninv;
        gy[k] = 0.f;
        gy[3 + k] = -(gwh[k] - wh[k] * pw) * winv;
    }
    gy[6] = 0.f;
    gy[7] = g_sec[j * 8 + 7];
}

// blend: out = m*part + (1-m)*base, part = sec scattered by index (rows without a source: base.detach())
__global__ void blend_backward_kernel(const float* __restrict__ g_out, const float* __restrict__ mask, long long n, int c,
                                      float* __restrict__ g_base, const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    g_base[i] = (1.f - mask[i / c]) * g_out[i];
}
__global__ void blend_backward_sec_kernel(const float* __restrict__ g_out, const float* __restrict__ mask,
                                          const int* __restrict__ index, long long n_sec, int c, float* __restrict__ g_sec,
                                          const int* __restrict__ n_sec_live) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= live_rows(n_sec, n_sec_live) * c) return;
    const long long row = index ? index[j / c] : j / c;
    g_sec[j] = mask[row] * g_out[row * c + (j % c)];
}

// Embedding backward: dx = g_x + sum_k 2^k (g_sin_k cos(2^k x) - g_cos_k sin(2^k x))
__global__ void embed_backward_kernel(const float* __restrict__ x, const float* __restrict__ g, long long n, int c,
                                      int n_freqs, float* __restrict__ gx, const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n, n_live) * c) return;
    const long long row = i / c;
    const int ch = (int)(i % c);
    const int ld = c * (2 * n_freqs + 1);
    const float v = x[i];
    const float* gr = g + row * ld;
    float a = gr[ch];
    for (int f = 0; f < n_freqs; ++f) {
        float s, co;
        sincosf(ldexpf(v, f), &s, &co);
        a += ldexpf(gr[c * (1 + 2 * f) + ch] * co - gr[c * (2 + 2 * f) + ch] * s, f);
    }
    gx[i] = a;
}

// ------------------------------------------------------------------ ray prologue / ray gradient fan-in (training)
// What render_rays does with a ray before the first field evaluation (models/rendering.py:275-300) as ONE launch: the view
// encoding of the direction (columns 3..5 of the ray, read in place: no contiguous copy) and the coarse depths.  The arithmetic
// is embed_kernel's and sample_coarse_kernel's: the outputs are theirs bit for bit.  Needs ns >= 3 (thread s < 3 of a ray also
// writes encoding channel s).
__global__ void ray_prologue_kernel(const float* __restrict__ rays, long long n_rays, int n_freqs,
                                    const float* __restrict__ z_steps, int ns, int use_disp, float perturb,
                                    const float* __restrict__ prand, float* __restrict__ dir_emb, float* __restrict__ z_out,
                                    const int* __restrict__ n_live) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(n_rays, n_live) * ns) return;
    const long long r = i / ns;
    const int s = (int)(i % ns);
    const float near = rays[r * 8 + 6], far = rays[r * 8 + 7];
    auto zat = [&](int k) -> float {
        const float t = z_steps[k];
        if (!use_disp) return near * (1.f - t) + far * t;
        return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
    };
    float z = zat(s);
    if (perturb > 0.f) {
        const float lower = s == 0 ? z : 0.5f * (zat(s - 1) + z);
        const float upper = s == ns - 1 ? z : 0.5f * (z + zat(s + 1));
        z = lower + (upper - lower) * (perturb * prand[i]);
    }
    z_out[i] = z;
    if (s < 3) {
        const int ld = 3 * (2 * n_freqs + 1);
        const float v = rays[r * 8 + 3 + s];
        float* o = dir_emb + r * ld;
        o[s] = v;
        for (int f = 0; f < n_freqs; ++f) {
            float sn, co;
            sincosf(ldexpf(v, f), &sn, &co);
            o[3 * (1 + 2 * f) + s] = sn;
            o[3 * (2 + 2 * f) + s] = co;
        }
    }
}

// The gradient of a ray that went into several consumers (two field evaluations, two compositing passes, the view encoding):
// autograd would add the (N,8) pieces pairwise (four launches), run the encoding's backward on the sum of its two gradients
// (one add, one launch), pad that to eight columns (a fill and a strided copy) and add it (one more).  Here: one launch,
// out = ((((g0 + g1) + g2) + g3) + [0 0 0 dview 0 0]) with dview = embed_backward(rays[:, 3:6], ga + gb) -- the same sums in
// the same order.  Any g may be null (a consumer that sent nothing).
struct RayFanArgs {
    const float* g[4]; const float* rays; const float* ga; const float* gb; long long n; int n_freqs; float* out; const int* n_live;
};
__global__ void ray_fan_backward_kernel(RayFanArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= live_rows(A.n, A.n_live) * 8) return;
    const long long row = i >> 3;
    const int col = (int)(i & 7);
    float acc = 0.f;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (A.g[k]) {
            const float t = A.g[k][i];
            acc = any ? acc + t : t;
            any = true;
        }
    if (col >= 3 && col < 6 && (A.ga || A.gb)) {
        const int ch = col - 3;
        const int ld = 3 * (2 * A.n_freqs + 1);
        const float v = A.rays[i];
        const float* ra = A.ga ? A.ga + row * ld : nullptr;
        const float* rb = A.gb ? A.gb + row * ld : nullptr;
        auto gat = [&](int k) -> float { return ra ? (rb ? ra[k] + rb[k] : ra[k]) : rb[k]; };
        float a = gat(ch);
        for (int f = 0; f < A.n_freqs; ++f) {
            float sn, co;
            sincosf(ldexpf(v, f), &sn, &co);
            a += ldexpf(gat(3 * (1 + 2 * f) + ch) * co - gat(3 * (2 + 2 * f) + ch) * sn, f);
        }
        acc = any ? acc + a : a;
    }
    A.out[i] = acc;
}

inline unsigned blocks_for(long long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

// ====================================================================== C ABI
// Every entry point with a row count exists twice: `mnrf_x(..., stream)` and `mnrf_x_n(..., n_live, stream)` (include/mnrf.h
// "live row counts on the device"); both are the `_impl` below, the first with n_live = null.
static int embed_impl(const float* x, int64_t n, int c, int n_freqs, float* out, const int32_t* n_live, void* stream) {
    if (n < 0 || c < 1 || n_freqs < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_embed: bad size");
    if (n == 0) return MNRF_OK;
    if (!x || !out) return mnrf_fail(MNRF_ERR_ARG, "mnrf_embed: null pointer");
    hipLaunchKernelGGL(embed_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, c,
                       n_freqs, out, n_live);
    return mnrf_check_launch("mnrf_embed");
}
extern "C" int mnrf_embed(const float* x, int64_t n, int c, int n_freqs, float* out, void* stream) {
    return embed_impl(x, n, c, n_freqs, out, nullptr, stream);
}
extern "C" int mnrf_embed_n(const float* x, int64_t n, int c, int n_freqs, float* out, const int32_t* n_live, void* stream) {
    return embed_impl(x, n, c, n_freqs, out, n_live, stream);
}

static int sample_coarse_impl(const float* rays, int64_t n_rays, const float* z_steps, int n_samples, int use_disp,
                              float perturb, const float* perturb_rand, float* z_vals, const int32_t* n_live, void* stream) {
    if (n_rays < 0 || n_samples < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_sample_coarse: bad size");
    if (n_rays == 0) return MNRF_OK;
    if (!rays || !z_steps || !z_vals) return mnrf_fail(MNRF_ERR_ARG, "mnrf_sample_coarse: null pointer");
    if (perturb > 0.f && !perturb_rand) return mnrf_fail(MNRF_ERR_ARG, "mnrf_sample_coarse: perturb > 0 needs perturb_rand");
    hipLaunchKernelGGL(sample_coarse_kernel, dim3(blocks_for(n_rays * n_samples, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, (long long)n_rays, z_steps, n_samples, use_disp, perturb, perturb_rand, z_vals, n_live);
    return mnrf_check_launch("mnrf_sample_coarse");
}
extern "C" int mnrf_sample_coarse(const float* rays, int64_t n_rays, const float* z_steps, int n_samples, int use_disp,
                                  float perturb, const float* perturb_rand, float* z_vals, void* stream) {
    return sample_coarse_impl(rays, n_rays, z_steps, n_samples, use_disp, perturb, perturb_rand, z_vals, nullptr, stream);
}
extern "C" int mnrf_sample_coarse_n(const float* rays, int64_t n_rays, const float* z_steps, int n_samples, int use_disp,
                                    float perturb, const float* perturb_rand, float* z_vals, const int32_t* n_live, void* stream) {
    return sample_coarse_impl(rays, n_rays, z_steps, n_samples, use_disp, perturb, perturb_rand, z_vals, n_live, stream);
}

static int composite_impl(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                          const float* noise, const float* rgb, const float* is_mirror, const float* pred_normal,
                          const float* normal, int white_back, float* weights, float* opacity, float* rgb_map,
                          float* depth, float* mirror_mask, float* surf_normal, float* surf_normal_grad,
                          float* normal_dif, float* x_surface, const int32_t* n_live, void* stream,
                          const float* u = nullptr, int u_per_ray = 0, int n_importance = 0, float* z_fine = nullptr) {
    if (n_rays < 0 || S < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: bad size");
    if (n_rays == 0) return MNRF_OK;
    if (!sigma || !z_vals) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: sigma and z_vals are required");
    if (rgb_map && !rgb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: rgb_map needs rgb");
    if (mirror_mask && !is_mirror) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: mirror_mask needs is_mirror");
    if (surf_normal && !pred_normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: surf_normal needs pred_normal");
    if (surf_normal_grad && !normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: surf_normal_grad needs normal");
    if (normal_dif && !(normal && pred_normal)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: normal_dif needs both normals");
    if (x_surface && !rays) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite: x_surface needs rays");
    CompArgs A{rays, (long long)n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back,
               weights, opacity, rgb_map, depth, mirror_mask, surf_normal, surf_normal_grad, normal_dif, x_surface, n_live};
    if (z_fine) {      // mnrf_composite_sample_n: the resampling of mnrf_sample_fine behind the compositing, same launch
        if (S < 3 || n_importance < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_sample_n: bad size");
        if (S > 256 || S + n_importance > SF_MAX)
            return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_composite_sample_n: needs S <= 256 and S + n_importance <= 512");
        if (!u || !weights) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_sample_n: null pointer (u, weights)");
        hipLaunchKernelGGL(composite_sample_kernel, dim3(blocks_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, A, u, u_per_ray,
                           n_importance, z_fine);
        return mnrf_check_launch("mnrf_composite_sample_n");
    }
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_composite");
}
extern "C" int mnrf_composite(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                              const float* noise, const float* rgb, const float* is_mirror, const float* pred_normal,
                              const float* normal, int white_back, float* weights, float* opacity, float* rgb_map,
                              float* depth, float* mirror_mask, float* surf_normal, float* surf_normal_grad,
                              float* normal_dif, float* x_surface, void* stream) {
    return composite_impl(rays, n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back, weights, opacity,
                          rgb_map, depth, mirror_mask, surf_normal, surf_normal_grad, normal_dif, x_surface, nullptr, stream);
}
extern "C" int mnrf_composite_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                                const float* noise, const float* rgb, const float* is_mirror, const float* pred_normal,
                                const float* normal, int white_back, float* weights, float* opacity, float* rgb_map,
                                float* depth, float* mirror_mask, float* surf_normal, float* surf_normal_grad,
                                float* normal_dif, float* x_surface, const int32_t* n_live, void* stream) {
    return composite_impl(rays, n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back, weights, opacity,
                          rgb_map, depth, mirror_mask, surf_normal, surf_normal_grad, normal_dif, x_surface, n_live, stream);
}

extern "C" int mnrf_composite_sample_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                                       const float* noise, const float* rgb, const float* is_mirror, const float* pred_normal,
                                       const float* normal, int white_back, float* weights, float* opacity, float* rgb_map,
                                       float* depth, float* mirror_mask, float* surf_normal, float* surf_normal_grad,
                                       float* normal_dif, float* x_surface, const float* u, int u_per_ray, int n_importance,
                                       float* z_fine, const int32_t* n_live, void* stream) {
    if (!z_fine) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_sample_n: z_fine is null");
    return composite_impl(rays, n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back, weights, opacity,
                          rgb_map, depth, mirror_mask, surf_normal, surf_normal_grad, normal_dif, x_surface, n_live, stream, u, u_per_ray,
                          n_importance, z_fine);
}

static int composite_backward_impl(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                                   const float* noise, const float* rgb, const float* is_mirror,
                                   const float* pred_normal, const float* normal, int white_back,
                                   const float* weights, const float* depth, const float* g_weights,
                                   const float* g_opacity, const float* g_rgb_map, const float* g_depth,
                                   const float* g_mirror_mask, const float* g_surf_normal,
                                   const float* g_surf_normal_grad, const float* g_normal_dif,
                                   const float* g_x_surface, float* d_sigma, float* d_rgb, float* d_is_mirror,
                                   float* d_pred_normal, float* d_normal, float* d_rays, int detach,
                                   const float* keep_mirror, const int32_t* n_live, void* stream) {
    (void)weights;   // recomputed in-kernel from sigma and z (cheaper than reading them back)
    if (n_rays < 0 || S < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: bad size");
    if (S > 64 * CB_MAXB) return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_composite_backward: needs S <= 256");
    if (n_rays == 0) return MNRF_OK;
    if (!sigma || !z_vals) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: sigma and z_vals are required");
    if ((g_rgb_map || d_rgb) && !rgb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: rgb missing");
    if ((g_mirror_mask || d_is_mirror) && !is_mirror) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: is_mirror missing");
    if ((g_surf_normal || d_pred_normal) && !pred_normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: pred_normal missing");
    if ((g_surf_normal_grad || d_normal) && !normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: normal missing");
    if (g_normal_dif && !(normal && pred_normal)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: normal_dif needs both normals");
    if ((g_x_surface || d_rays) && !rays) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: rays missing");
    if (d_rays && !depth) return mnrf_fail(MNRF_ERR_ARG, "mnrf_composite_backward: d_rays needs the forward depth");
    CompBwdArgs A{rays, (long long)n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back,
                  depth, g_weights, g_opacity, g_rgb_map, g_depth, g_mirror_mask, g_surf_normal, g_surf_normal_grad,
                  g_normal_dif, g_x_surface, d_sigma, d_rgb, d_is_mirror, d_pred_normal, d_normal, d_rays, detach, keep_mirror, n_live};
    hipLaunchKernelGGL(composite_backward_kernel, dim3(blocks_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_composite_backward");
}
extern "C" int mnrf_composite_backward(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                                       const float* noise, const float* rgb, const float* is_mirror,
                                       const float* pred_normal, const float* normal, int white_back,
                                       const float* weights, const float* depth, const float* g_weights,
                                       const float* g_opacity, const float* g_rgb_map, const float* g_depth,
                                       const float* g_mirror_mask, const float* g_surf_normal,
                                       const float* g_surf_normal_grad, const float* g_normal_dif,
                                       const float* g_x_surface, float* d_sigma, float* d_rgb, float* d_is_mirror,
                                       float* d_pred_normal, float* d_normal, float* d_rays, int detach,
                                       const float* keep_mirror, void* stream) {
    return composite_backward_impl(rays, n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back, weights, depth,
                                   g_weights, g_opacity, g_rgb_map, g_depth, g_mirror_mask, g_surf_normal, g_surf_normal_grad,
                                   g_normal_dif, g_x_surface, d_sigma, d_rgb, d_is_mirror, d_pred_normal, d_normal, d_rays, detach,
                                   keep_mirror, nullptr, stream);
}
extern "C" int mnrf_composite_backward_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                                         const float* noise, const float* rgb, const float* is_mirror,
                                         const float* pred_normal, const float* normal, int white_back,
                                         const float* weights, const float* depth, const float* g_weights,
                                         const float* g_opacity, const float* g_rgb_map, const float* g_depth,
                                         const float* g_mirror_mask, const float* g_surf_normal,
                                         const float* g_surf_normal_grad, const float* g_normal_dif,
                                         const float* g_x_surface, float* d_sigma, float* d_rgb, float* d_is_mirror,
                                         float* d_pred_normal, float* d_normal, float* d_rays, int detach,
                                         const float* keep_mirror, const int32_t* n_live, void* stream) {
    return composite_backward_impl(rays, n_rays, S, sigma, z_vals, noise, rgb, is_mirror, pred_normal, normal, white_back, weights, depth,
                                   g_weights, g_opacity, g_rgb_map, g_depth, g_mirror_mask, g_surf_normal, g_surf_normal_grad,
                                   g_normal_dif, g_x_surface, d_sigma, d_rgb, d_is_mirror, d_pred_normal, d_normal, d_rays, detach,
                                   keep_mirror, n_live, stream);
}

static int sample_fine_impl(const float* z_coarse, const float* weights, int64_t n_rays, int S, const float* u,
                            int u_per_ray, int n_importance, float* z_fine, const int32_t* n_live, void* stream) {
    if (n_rays < 0 || S < 3 || n_importance < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_sample_fine: bad size");
    if (S > 256 || S + n_importance > SF_MAX)
        return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_sample_fine: needs S <= 256 and S + n_importance <= 512");
    if (n_rays == 0) return MNRF_OK;
    if (!z_coarse || !weights || !u || !z_fine) return mnrf_fail(MNRF_ERR_ARG, "mnrf_sample_fine: null pointer");
    hipLaunchKernelGGL(sample_fine_kernel, dim3(blocks_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, z_coarse, weights,
                       (long long)n_rays, S, u, u_per_ray, n_importance, z_fine, n_live);
    return mnrf_check_launch("mnrf_sample_fine");
}
extern "C" int mnrf_sample_fine(const float* z_coarse, const float* weights, int64_t n_rays, int S, const float* u,
                                int u_per_ray, int n_importance, float* z_fine, void* stream) {
    return sample_fine_impl(z_coarse, weights, n_rays, S, u, u_per_ray, n_importance, z_fine, nullptr, stream);
}
extern "C" int mnrf_sample_fine_n(const float* z_coarse, const float* weights, int64_t n_rays, int S, const float* u,
                                  int u_per_ray, int n_importance, float* z_fine, const int32_t* n_live, void* stream) {
    return sample_fine_impl(z_coarse, weights, n_rays, S, u, u_per_ray, n_importance, z_fine, n_live, stream);
}

static int threshold_impl(float* mask, int64_t n, int32_t* any, const int32_t* n_live, void* stream) {
    if (n < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_threshold_mask: bad size");
    if (n == 0) return MNRF_OK;
    if (!mask) return mnrf_fail(MNRF_ERR_ARG, "mnrf_threshold_mask: null pointer");
    hipLaunchKernelGGL(threshold_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, mask, (long long)n, any, n_live);
    return mnrf_check_launch("mnrf_threshold_mask");
}
extern "C" int mnrf_threshold_mask(float* mask, int64_t n, int32_t* any, void* stream) { return threshold_impl(mask, n, any, nullptr, stream); }
extern "C" int mnrf_threshold_mask_n(float* mask, int64_t n, int32_t* any, const int32_t* n_live, void* stream) {
    return threshold_impl(mask, n, any, n_live, stream);
}

static int reflect_compact_impl(const float* rays, const float* x_surface, const float* normal,
                                const float* normal_noise, float noise_std, const float* mask, int64_t n_rays,
                                int compact, float near2, float* sec_rays, int32_t* index, int32_t* count,
                                float* reflect_dir, const int32_t* n_live, int32_t* slot, void* stream) {
    if (n_rays < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_compact: bad size");
    if (!count) return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_compact: count is null");
    if (n_rays > 0 && (!rays || !x_surface || !normal || !sec_rays || !index))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_compact: null pointer");
    if (compact && n_rays > 0 && !mask) return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_compact: compaction needs the mask");
    ReflArgs A{rays, x_surface, normal, normal_noise, noise_std, mask, (long long)n_rays, compact, near2,
               sec_rays, index, count, reflect_dir, n_live, slot};
    if (!compact && n_rays > 0)
        hipLaunchKernelGGL(reflect_all_kernel, dim3(blocks_for(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL(reflect_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_reflect_compact");
}
extern "C" int mnrf_reflect_compact(const float* rays, const float* x_surface, const float* normal,
                                    const float* normal_noise, float noise_std, const float* mask, int64_t n_rays,
                                    int compact, float near2, float* sec_rays, int32_t* index, int32_t* count,
                                    float* reflect_dir, void* stream) {
    return reflect_compact_impl(rays, x_surface, normal, normal_noise, noise_std, mask, n_rays, compact, near2, sec_rays, index, count,
                                reflect_dir, nullptr, nullptr, stream);
}
extern "C" int mnrf_reflect_compact_n(const float* rays, const float* x_surface, const float* normal,
                                      const float* normal_noise, float noise_std, const float* mask, int64_t n_rays,
                                      int compact, float near2, float* sec_rays, int32_t* index, int32_t* count,
                                      float* reflect_dir, const int32_t* n_live, int32_t* slot, void* stream) {
    return reflect_compact_impl(rays, x_surface, normal, normal_noise, noise_std, mask, n_rays, compact, near2, sec_rays, index, count,
                                reflect_dir, n_live, slot, stream);
}

extern "C" int mnrf_blend2_n(const float* base_a, const float* sec_a, const float* base_b, const float* sec_b, const int32_t* slot,
                             const float* mask, int64_t n, int c, float* out_a, float* out_b, const int32_t* n_live, void* stream) {
    if (n < 0 || c < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend2_n: bad size");
    if (n == 0) return MNRF_OK;
    if (!slot || !mask || (!base_a && !base_b)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend2_n: null pointer");
    if ((base_a && (!sec_a || !out_a)) || (base_b && (!sec_b || !out_b))) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend2_n: a tensor needs base, sec and out");
    hipLaunchKernelGGL(blend2_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, (hipStream_t)stream, base_a, sec_a, base_b, sec_b, slot, mask,
                       (long long)n, c, out_a, out_b, n_live);
    return mnrf_check_launch("mnrf_blend2_n");
}
extern "C" int mnrf_blend2_backward_n(const float* g_out_a, const float* g_out_b, const int32_t* slot, const float* mask, int64_t n, int c,
                                      float* g_base_a, float* g_sec_a, float* g_base_b, float* g_sec_b, const int32_t* n_live, void* stream) {
    if (n < 0 || c < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend2_backward_n: bad size");
    if (n == 0) return MNRF_OK;
    if (!slot || !mask) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend2_backward_n: null pointer");
    hipLaunchKernelGGL(blend2_backward_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, (hipStream_t)stream, g_out_a, g_out_b, slot, mask,
                       (long long)n, c, g_base_a, g_sec_a, g_base_b, g_sec_b, n_live);
    return mnrf_check_launch("mnrf_blend2_backward_n");
}

// n_live: live rows of base / mask / out (null: n); n_sec_live: live rows of sec / index (null: n_sec)
static int blend_scatter_impl(const float* base, const float* sec, const int32_t* index, int64_t n_sec,
                              const float* mask, int64_t n, int c, float* out, float* reflect_out, const int32_t* n_sec_live,
                              const int32_t* n_live, void* stream) {
    if (n < 0 || n_sec < 0 || c < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_scatter: bad size");
    if (n == 0) return MNRF_OK;
    if (!base || !mask || !out) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_scatter: null pointer");
    if (n_sec > 0 && !sec) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_scatter: sec is null");
    if (!index && n_sec != n) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_scatter: without an index n_sec must equal n");
    hipStream_t s = (hipStream_t)stream;
    const int direct = index == nullptr;
    hipLaunchKernelGGL(blend_all_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, s, base, sec, mask, (long long)n, c,
                       direct, out, reflect_out, n_live);
    if (!direct && n_sec > 0)
        hipLaunchKernelGGL(blend_scatter_kernel, dim3(blocks_for(n_sec * c, 256)), dim3(256), 0, s, base, sec, index,
                           (long long)n_sec, mask, c, out, reflect_out, n_sec_live);
    return mnrf_check_launch("mnrf_blend_scatter");
}
extern "C" int mnrf_blend_scatter(const float* base, const float* sec, const int32_t* index, int64_t n_sec,
                                  const float* mask, int64_t n, int c, float* out, float* reflect_out, void* stream) {
    return blend_scatter_impl(base, sec, index, n_sec, mask, n, c, out, reflect_out, nullptr, nullptr, stream);
}
extern "C" int mnrf_blend_scatter_n(const float* base, const float* sec, const int32_t* index, int64_t n_sec,
                                    const float* mask, int64_t n, int c, float* out, float* reflect_out,
                                    const int32_t* n_sec_live, const int32_t* n_live, void* stream) {
    return blend_scatter_impl(base, sec, index, n_sec, mask, n, c, out, reflect_out, n_sec_live, n_live, stream);
}

static int reflect_backward_impl(const float* rays, const float* normal, const int32_t* index, int64_t n_sec,
                                 const float* g_sec, int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays,
                                 const int32_t* n_sec_live, void* stream) {
    if (n_rays < 0 || n_sec < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_backward: bad size");
    if (n_rays == 0) return MNRF_OK;
    if (!rays || !normal || !g_x_surface || !g_normal || !g_rays || (n_sec > 0 && !g_sec))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    mnrf::zero_fill(s, g_x_surface, n_rays * 3 * sizeof(float), g_normal, n_rays * 3 * sizeof(float), g_rays, n_rays * 8 * sizeof(float));
    if (n_sec > 0)
        hipLaunchKernelGGL(reflect_backward_kernel, dim3(blocks_for(n_sec, 256)), dim3(256), 0, s, rays, normal, index, (long long)n_sec,
                           g_sec, g_x_surface, g_normal, g_rays, n_sec_live);
    return mnrf_check_launch("mnrf_reflect_backward");
}
extern "C" int mnrf_reflect_backward(const float* rays, const float* normal, const int32_t* index, int64_t n_sec,
                                     const float* g_sec, int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays,
                                     void* stream) {
    return reflect_backward_impl(rays, normal, index, n_sec, g_sec, n_rays, g_x_surface, g_normal, g_rays, nullptr, stream);
}
extern "C" int mnrf_reflect_backward_n(const float* rays, const float* normal, const int32_t* index, int64_t n_sec,
                                       const float* g_sec, int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays,
                                       const int32_t* n_sec_live, void* stream) {
    return reflect_backward_impl(rays, normal, index, n_sec, g_sec, n_rays, g_x_surface, g_normal, g_rays, n_sec_live, stream);
}

extern "C" int mnrf_reflect_backward_gather_n(const float* rays, const float* normal, const int32_t* slot, const float* g_sec,
                                              int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays, const int32_t* n_live,
                                              void* stream) {
    if (n_rays < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_backward_gather_n: bad size");
    if (n_rays == 0) return MNRF_OK;
    if (!rays || !normal || !slot || !g_sec || !g_x_surface || !g_normal || !g_rays)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_reflect_backward_gather_n: null pointer");
    hipLaunchKernelGGL(reflect_backward_gather_kernel, dim3(blocks_for(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rays, normal, slot,
                       g_sec, (long long)n_rays, g_x_surface, g_normal, g_rays, n_live);
    return mnrf_check_launch("mnrf_reflect_backward_gather_n");
}

static int blend_backward_impl(const float* g_out, const float* mask, const int32_t* index, int64_t n_sec, int64_t n, int c,
                               float* g_base, float* g_sec, const int32_t* n_sec_live, const int32_t* n_live, void* stream) {
    if (n < 0 || n_sec < 0 || c < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_backward: bad size");
    if (n == 0) return MNRF_OK;
    if (!g_out || !mask) return mnrf_fail(MNRF_ERR_ARG, "mnrf_blend_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (g_base) hipLaunchKernelGGL(blend_backward_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, s, g_out, mask, (long long)n, c, g_base, n_live);
    if (g_sec && n_sec > 0)
        hipLaunchKernelGGL(blend_backward_sec_kernel, dim3(blocks_for(n_sec * c, 256)), dim3(256), 0, s, g_out, mask, index,
                           (long long)n_sec, c, g_sec, n_sec_live);
    return mnrf_check_launch("mnrf_blend_backward");
}
extern "C" int mnrf_blend_backward(const float* g_out, const float* mask, const int32_t* index, int64_t n_sec, int64_t n, int c,
                                   float* g_base, float* g_sec, void* stream) {
    return blend_backward_impl(g_out, mask, index, n_sec, n, c, g_base, g_sec, nullptr, nullptr, stream);
}
extern "C" int mnrf_blend_backward_n(const float* g_out, const float* mask, const int32_t* index, int64_t n_sec, int64_t n, int c,
                                     float* g_base, float* g_sec, const int32_t* n_sec_live, const int32_t* n_live, void* stream) {
    return blend_backward_impl(g_out, mask, index, n_sec, n, c, g_base, g_sec, n_sec_live, n_live, stream);
}

// Ray gradients of one field evaluation from its per-sample position / view-encoding gradients: x = o + d z
// (models/rendering.py:302) gives dL/do = sum_s dL/dx_s and dL/dd = sum_s z_s dL/dx_s; the view encoding is per ray, so its
// gradient is the sum over the ray's samples.  One wavefront per ray, fixed summation order.  (This was seven torch kernels
// per evaluation -- zeros, two slices, a product, three reductions -- 28 launches of a training step.)
__global__ __launch_bounds__(256) void ray_grads_kernel(const float* __restrict__ d_xyz, const float* __restrict__ z,
                                                        const float* __restrict__ d_dir, long long n_rays, int spr,
                                                        float* __restrict__ g_rays, float* __restrict__ g_de,
                                                        const int* __restrict__ n_live) {
    // one WORKGROUP per ray (round 5; it was one wavefront per ray, 64 dependent 256-byte loads deep for the view-encoding sum:
    // 23 us per launch for 257 live rays): wave 0 sums the position gradients, all four waves the 32-float rows of d_dir
    __shared__ float part[8][32];
    const int lane = threadIdx.x & 63;
    const long long ray = blockIdx.x;
    if (ray >= live_rows(n_rays, n_live)) return;
    if (g_rays && threadIdx.x < 64) {
        float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = lane; s < spr; s += 64) {
            const float* p = d_xyz + (ray * spr + s) * 3;
            const float zz = z[ray * spr + s];
            const float p0 = p[0], p1 = p[1], p2 = p[2];
            a[0] += p0; a[1] += p1; a[2] += p2;
            a[3] += zz * p0; a[4] += zz * p1; a[5] += zz * p2;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a[k] += __shfl_xor(a[k], d);
        if (lane < 8) g_rays[ray * 8 + lane] = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : lane == 3 ? a[3]
                                               : lane == 4 ? a[4] : lane == 5 ? a[5] : 0.f;      // (near, far: no gradient)
    }
    if (g_de) {      // d_dir: (B, 32) rows, the 27 channels of Embedding(4) used; fixed summation order
        const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;      // 8 groups of 32 threads, samples grp, grp + 8, ...
        float acc = 0.f;
        for (int s = grp; s < spr; s += 8) acc += d_dir[(ray * spr + s) * 32 + c];
        part[grp][c] = acc;
        __syncthreads();
        if (threadIdx.x < 27) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
            g_de[ray * 27 + threadIdx.x] = t;
        }
    }
}

static int ray_grads_impl(const float* d_xyz, const float* z_vals, const float* d_dir, int64_t n_rays, int spr, float* g_rays,
                          float* g_de, const int32_t* n_live, void* stream) {
    if (n_rays < 0 || spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_grads: bad size");
    if (n_rays == 0 || (!g_rays && !g_de)) return MNRF_OK;
    if ((g_rays && (!d_xyz || !z_vals)) || (g_de && !d_dir)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_grads: null pointer");
    if (n_rays > 0x7fffffff) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_grads: too many rays for one launch");
    hipLaunchKernelGGL(ray_grads_kernel, dim3((unsigned)n_rays), dim3(256), 0, (hipStream_t)stream, d_xyz, z_vals, d_dir,
                       (long long)n_rays, spr, g_rays, g_de, n_live);
    return mnrf_check_launch("mnrf_ray_grads");
}
extern "C" int mnrf_ray_grads(const float* d_xyz, const float* z_vals, const float* d_dir, int64_t n_rays, int spr, float* g_rays,
                              float* g_de, void* stream) {
    return ray_grads_impl(d_xyz, z_vals, d_dir, n_rays, spr, g_rays, g_de, nullptr, stream);
}
extern "C" int mnrf_ray_grads_n(const float* d_xyz, const float* z_vals, const float* d_dir, int64_t n_rays, int spr, float* g_rays,
                                float* g_de, const int32_t* n_live, void* stream) {
    return ray_grads_impl(d_xyz, z_vals, d_dir, n_rays, spr, g_rays, g_de, n_live, stream);
}

static int embed_backward_impl(const float* x, const float* g_out, int64_t n, int c, int n_freqs, float* g_x, const int32_t* n_live,
                               void* stream) {
    if (n < 0 || c < 1 || n_freqs < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_embed_backward: bad size");
    if (n == 0) return MNRF_OK;
    if (!x || !g_out || !g_x) return mnrf_fail(MNRF_ERR_ARG, "mnrf_embed_backward: null pointer");
    hipLaunchKernelGGL(embed_backward_kernel, dim3(blocks_for(n * c, 256)), dim3(256), 0, (hipStream_t)stream, x, g_out, (long long)n, c,
                       n_freqs, g_x, n_live);
    return mnrf_check_launch("mnrf_embed_backward");
}
extern "C" int mnrf_embed_backward(const float* x, const float* g_out, int64_t n, int c, int n_freqs, float* g_x, void* stream) {
    return embed_backward_impl(x, g_out, n, c, n_freqs, g_x, nullptr, stream);
}
extern "C" int mnrf_embed_backward_n(const float* x, const float* g_out, int64_t n, int c, int n_freqs, float* g_x,
                                     const int32_t* n_live, void* stream) {
    return embed_backward_impl(x, g_out, n, c, n_freqs, g_x, n_live, stream);
}

extern "C" int mnrf_ray_prologue_n(const float* rays, int64_t n_rays, int n_freqs_dir, const float* z_steps, int n_samples, int use_disp,
                                   float perturb, const float* perturb_rand, float* dir_emb, float* z_vals, const int32_t* n_live,
                                   void* stream) {
    if (n_rays < 0 || n_samples < 3 || n_freqs_dir < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_prologue_n: bad size (n_samples >= 3)");
    if (n_rays == 0) return MNRF_OK;
    if (!rays || !z_steps || !z_vals || !dir_emb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_prologue_n: null pointer");
    if (perturb > 0.f && !perturb_rand) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_prologue_n: perturb > 0 needs perturb_rand");
    hipLaunchKernelGGL(ray_prologue_kernel, dim3(blocks_for(n_rays * n_samples, 256)), dim3(256), 0, (hipStream_t)stream, rays,
                       (long long)n_rays, n_freqs_dir, z_steps, n_samples, use_disp, perturb, perturb_rand, dir_emb, z_vals, n_live);
    return mnrf_check_launch("mnrf_ray_prologue_n");
}

extern "C" int mnrf_ray_fan_backward_n(const float* g0, const float* g1, const float* g2, const float* g3, const float* rays,
                                       const float* g_dir_a, const float* g_dir_b, int64_t n_rays, int n_freqs_dir, float* g_rays,
                                       const int32_t* n_live, void* stream) {
    if (n_rays < 0 || n_freqs_dir < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_fan_backward_n: bad size");
    if (n_rays == 0) return MNRF_OK;
    if (!g_rays || ((g_dir_a || g_dir_b) && !rays)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_fan_backward_n: null pointer");
    if (!g0 && !g1 && !g2 && !g3 && !g_dir_a && !g_dir_b)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_ray_fan_backward_n: no gradient at all (the caller returns None instead)");
    RayFanArgs A{{g0, g1, g2, g3}, rays, g_dir_a, g_dir_b, (long long)n_rays, n_freqs_dir, g_rays, n_live};
    hipLaunchKernelGGL(ray_fan_backward_kernel, dim3(blocks_for(n_rays * 8, 256)), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_ray_fan_backward_n");
}

extern "C" int mnrf_generate_rays(int H, int W, float focal, const float* c2w_host12, float near, float far, float* rays,
                                  void* stream) {
    if (H < 1 || W < 1 || !c2w_host12 || !rays) return mnrf_fail(MNRF_ERR_ARG, "mnrf_generate_rays: bad argument");
    Pose p;
    memcpy(p.m, c2w_host12, sizeof(p.m));
    hipLaunchKernelGGL(generate_rays_kernel, dim3(blocks_for((long long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, H, W,
                       focal, p, near, far, rays);
    return mnrf_check_launch("mnrf_generate_rays");
}
