// mnrf_tcnn.hip -- the hash-grid field of BASELINE config 5 (MirrorNeRFTcnn) for gfx950.
//
// Replaces, per sample, models/mirror_nerf_tcnn.py:151-259: multiresolution hash encoding (16 levels
// x 2 features, 2^19 entries per hashed level; arithmetic of models/gridencoder/src/gridencoder.cu:
// 51-89, 91-272 with tinycudann's per_level_scale, mirror_nerf_tcnn.py:38), degree-4 spherical
// harmonics of the view direction (models/shencoder/src/shencoder.cu:49-79), the bias-free sigma /
// colour / normal MLPs and the mirror head, plus the density-gradient normal through the encoding's
// analytic d/dx.  PARITY UNPINNED against tinycudann (absent from the image): checked
// against this repository's own CPU restatement (oracle `tcnn_field_forward`).
//
// Regime: 22 kFLOP and 128 random 8-byte gathers per sample -> gather/L2 bound, not MFMA bound.
// One thread per sample; the 11 k weights sit in LDS and are read as wave-uniform (broadcast)
// ds_read_b128; the table (53 MB fp32 at bound 6) lives in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NL = 16;            // levels
constexpr float EPS32 = 1.1920928955078125e-07f;
// weight blob (floats; rows padded to a multiple of 4 columns so that every row is float4-readable)
constexpr int W_S0 = 0;           // 64 x 32
constexpr int W_S1 = 2048;        // 16 x 64
constexpr int W_C0 = 3072;        // 64 x 32 (31 used: 16 SH + 15 geo)
constexpr int W_C1 = 5120;        // 64 x 64
constexpr int W_C2 = 9216;        // 3 x 64
constexpr int W_N0 = 9408;        // 64 x 16 (15 used)
constexpr int W_N1 = 10432;       // 3 x 64
constexpr int W_M0 = 10624;       // 32 x 16 (15 used)
constexpr int B_M0 = 11136;       // 32
constexpr int W_M1 = 11168;       // 1 x 32
constexpr int B_M1 = 11200;       // 1
constexpr int W_TOTAL = 11204;

struct TcnnArgs {
    const float* table;           // (entries, 2) fp32
    const float* weights;         // W_TOTAL floats
    long long B;
    const float* xyz; long long xyz_stride;          // positions (and, at +3, raw directions) or null
    const float* rays; const float* z_vals; int spr;  // ray mode
    const float* dirs; long long dir_stride;          // per-ray raw directions (ray mode) or null
    float bound;
    float scale[NL];              // exp2(level*S)*H - 1
    unsigned res[NL];             // ceil(scale) + 1
    unsigned off[NL + 1];         // level offsets in entries
    float* sigma; float* rgb; float* pred_normal; float* is_mirror; float* normal; float* geo_feat;
};

extern __shared__ __attribute__((aligned(16))) float wlds[];

constexpr int TPB = 256;                        // threads (= samples) per workgroup
constexpr int VEC_OFF = (W_TOTAL + 3) / 4 * 4;  // per-thread vector buffer vec[64][TPB] behind the weights
#define VEC(k) wlds[VEC_OFF + (k) * TPB + threadIdx.x]

// Layers run as "inputs in registers, loop over outputs": the output loop is NOT unrolled (a fully
// unrolled 11 k-FMA body makes hipcc hoist thousands of LDS reads and spill 2-4 KB per lane); each
// output goes to the thread's column of an LDS vector buffer and is re-loaded as the next input.
template <int NI>
__device__ __forceinline__ float dot_row(const float (&in)[NI], int woff) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i += 4) {
        const f32x4 w = *(const f32x4*)(wlds + woff + i);
        a = fmaf(w[0], in[i], a); a = fmaf(w[1], in[i + 1], a);
        a = fmaf(w[2], in[i + 2], a); a = fmaf(w[3], in[i + 3], a);
    }
    return a;
}

template <int N>
__device__ __forceinline__ void load_vec(float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = VEC(k);
}

__device__ __forceinline__ unsigned grid_index(unsigned x, unsigned y, unsigned z, unsigned hsize, unsigned res) {
    // get_grid_index (gridencoder.cu:68-89): dense while the running stride fits, else the spatial hash
    unsigned stride = 1, index = 0;
    const unsigned p[3] = {x, y, z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (stride <= hsize) {
            index += p[d] * stride;
            stride *= res + 1;
        }
    }
    if (stride > hsize) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % hsize;
}

// one level of the encoding: the two features and (GRAD) their derivatives w.r.t. the [0,1] coordinates
template <bool GRAD>
__device__ __forceinline__ void encode_level(const TcnnArgs& A, int lv, const float (&u)[3], bool oob, float& a0, float& a1,
                                             float (&g0)[3], float (&g1)[3]) {
    const float scale = A.scale[lv];
    const unsigned res = A.res[lv];
    const unsigned hsize = A.off[lv + 1] - A.off[lv];
    const float2* tab = (const float2*)A.table + A.off[lv];
    unsigned pg[3];
    float fr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = u[a] * scale + 0.5f;
        const float fl = floorf(pos);
        pg[a] = (unsigned)fl;
        fr[a] = pos - fl;
    }
    a0 = 0.f; a1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g0[a] = 0.f; g1[a] = 0.f; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wx = (c & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (c & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (c & 4) ? fr[2] : 1.f - fr[2];
        float2 v = make_float2(0.f, 0.f);
        if (!oob) v = tab[grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res)];
        const float w = wx * wy * wz;
        a0 += w * v.x; a1 += w * v.y;
        if (GRAD) {
            const float sx = ((c & 1) ? scale : -scale) * wy * wz;
            const float sy = ((c & 2) ? scale : -scale) * wx * wz;
            const float sz = ((c & 4) ? scale : -scale) * wx * wy;
            g0[0] += sx * v.x; g1[0] += sx * v.y;
            g0[1] += sy * v.x; g1[1] += sy * v.y;
            g0[2] += sz * v.x; g1[2] += sz * v.y;
        }
    }
}

template <bool SIGMA_ONLY, bool GRAD>
__global__ __launch_bounds__(TPB) void tcnn_kernel(TcnnArgs A) {
    for (int k = threadIdx.x; k < W_TOTAL; k += TPB) wlds[k] = A.weights[k];
    __syncthreads();
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    const bool live = i < A.B;
    if (!live) i = A.B - 1;
    float x[3], d[3] = {0.f, 0.f, 0.f};
    if (A.xyz) {
        const float* p = A.xyz + i * A.xyz_stride;
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
        if (!SIGMA_ONLY) { d[0] = p[3]; d[1] = p[4]; d[2] = p[5]; }
    } else {
        const long long ray = i / A.spr;
        const float* r = A.rays + ray * 8;
        const float z = A.z_vals[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
        if (!SIGMA_ONLY) {
            const float* dp = A.dirs ? A.dirs + ray * A.dir_stride : r + 3;
            d[0] = dp[0]; d[1] = dp[1]; d[2] = dp[2];
        }
    }
    // ---- x -> [0,1] (mirror_nerf_tcnn.py:224), multiresolution hash encoding into the vector buffer
    float u[3];
    bool oob = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        u[a] = (x[a] + A.bound) / (2.f * A.bound);
        oob |= u[a] < 0.f || u[a] > 1.f;
    }
#pragma unroll 1
    for (int lv = 0; lv < NL; ++lv) {
        float a0, a1, g0[3], g1[3];
        encode_level<false>(A, lv, u, oob, a0, a1, g0, g1);
        VEC(2 * lv) = a0;
        VEC(2 * lv + 1) = a1;
    }
    // ---- sigma net: 32 -> 64 (ReLU) -> 16; sigma = h[0] raw, geo_feat = h[1:16]  (228-236)
    unsigned long long relu_bits = 0;
    {
        float in[32];
        load_vec(in);
#pragma unroll 1
        for (int o = 0; o < 64; ++o) {
            const float pre = dot_row(in, W_S0 + o * 32);
            relu_bits |= (unsigned long long)(pre > 0.f) << o;
            VEC(o) = fmaxf(pre, 0.f);
        }
    }
    float geo[16];
    {
        float in[64];
        load_vec(in);
#pragma unroll 1
        for (int o = 0; o < 16; ++o) VEC(o) = dot_row(in, W_S1 + o * 64);
        if (A.sigma && live) A.sigma[i] = VEC(0);
#pragma unroll
        for (int k = 0; k < 15; ++k) geo[k] = VEC(1 + k);
        geo[15] = 0.f;
        if (A.geo_feat && live) {
#pragma unroll
            for (int k = 0; k < 15; ++k) A.geo_feat[i * 15 + k] = geo[k];
        }
    }
    if (GRAD && A.normal) {
        // d sigma/dx = (1/(2 bound)) * dydx^T W_s0^T (relu' * W_s1[0,:]); the encoding derivatives are
        // recomputed level by level (second gather pass, L2-resident) instead of being kept (96 values)
        float genc[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) genc[e] = 0.f;
#pragma unroll 1
        for (int k = 0; k < 64; ++k) {
            const float s = ((relu_bits >> k) & 1ull) ? wlds[W_S1 + k] : 0.f;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                const f32x4 w = *(const f32x4*)(wlds + W_S0 + k * 32 + e);
                genc[e] = fmaf(s, w[0], genc[e]); genc[e + 1] = fmaf(s, w[1], genc[e + 1]);
                genc[e + 2] = fmaf(s, w[2], genc[e + 2]); genc[e + 3] = fmaf(s, w[3], genc[e + 3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) VEC(e) = genc[e];
        float gd[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int lv = 0; lv < NL; ++lv) {
            float a0, a1, g0[3], g1[3];
            encode_level<true>(A, lv, u, oob, a0, a1, g0, g1);
            const float e0 = VEC(2 * lv), e1 = VEC(2 * lv + 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) gd[a] += e0 * g0[a] + e1 * g1[a];
        }
        const float s = 1.f / (2.f * A.bound);
        const float n0 = -gd[0] * s, n1 = -gd[1] * s, n2 = -gd[2] * s;
        const float inv = 1.f / sqrtf(fmaxf(n0 * n0 + n1 * n1 + n2 * n2, EPS32));
        if (live) { A.normal[i * 3] = n0 * inv; A.normal[i * 3 + 1] = n1 * inv; A.normal[i * 3 + 2] = n2 * inv; }
    }
    // ---- predicted normal: 15 -> 64 (ReLU) -> 3, l2-normalised (249-255, 185-192)
    if (A.pred_normal) {
#pragma unroll 1
        for (int o = 0; o < 64; ++o) VEC(o) = fmaxf(dot_row(geo, W_N0 + o * 16), 0.f);
        float hn[64];
        load_vec(hn);
        float v[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) v[o] = dot_row(hn, W_N1 + o * 64);
        const float inv = 1.f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], EPS32));
        if (live) { A.pred_normal[i * 3] = v[0] * inv; A.pred_normal[i * 3 + 1] = v[1] * inv; A.pred_normal[i * 3 + 2] = v[2] * inv; }
    }
    if (SIGMA_ONLY) return;
    // ---- colour: cat[SH4(d), geo_feat] -> 64 -> 64 -> 3 sigmoid (238-247)
    {
        float in[32];
        const float X = d[0], Y = d[1], Z = d[2];
        const float xy = X * Y, xz = X * Z, yz = Y * Z, x2 = X * X, y2 = Y * Y, z2 = Z * Z;
        in[0] = 0.28209479177387814f;
        in[1] = -0.48860251190291987f * Y;
        in[2] = 0.48860251190291987f * Z;
        in[3] = -0.48860251190291987f * X;
        in[4] = 1.0925484305920792f * xy;
        in[5] = -1.0925484305920792f * yz;
        in[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        in[7] = -1.0925484305920792f * xz;
        in[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        in[9] = 0.59004358992664352f * Y * (-3.0f * x2 + y2);
        in[10] = 2.8906114426405538f * xy * Z;
        in[11] = 0.45704579946446572f * Y * (1.0f - 5.0f * z2);
        in[12] = 0.3731763325901154f * Z * (5.0f * z2 - 3.0f);
        in[13] = 0.45704579946446572f * X * (1.0f - 5.0f * z2);
        in[14] = 1.4453057213202769f * Z * (x2 - y2);
        in[15] = 0.59004358992664352f * X * (-x2 + 3.0f * y2);
#pragma unroll
        for (int k = 0; k < 15; ++k) in[16 + k] = geo[k];
        in[31] = 0.f;
#pragma unroll 1
        for (int o = 0; o < 64; ++o) VEC(o) = fmaxf(dot_row(in, W_C0 + o * 32), 0.f);
        float c1[64];
        load_vec(c1);
#pragma unroll 1
        for (int o = 0; o < 64; ++o) VEC(o) = fmaxf(dot_row(c1, W_C1 + o * 64), 0.f);
        load_vec(c1);
        if (A.rgb && live) {
#pragma unroll
            for (int k = 0; k < 3; ++k) A.rgb[i * 3 + k] = 1.f / (1.f + expf(-dot_row(c1, W_C2 + k * 64)));
        }
    }
    // ---- mirror probability: 15 -> 32 LeakyReLU(0.01) -> 1 sigmoid, with biases (141-149)
    if (A.is_mirror) {
#pragma unroll 1
        for (int o = 0; o < 32; ++o) {
            const float v = dot_row(geo, W_M0 + o * 16) + wlds[B_M0 + o];
            VEC(o) = v > 0.f ? v : 0.01f * v;
        }
        float hm[32];
        load_vec(hm);
        if (live) A.is_mirror[i] = 1.f / (1.f + expf(-(dot_row(hm, W_M1) + wlds[B_M1])));
    }
}
#undef VEC

}  // namespace

extern "C" int mnrf_tcnn_weight_floats(void) { return W_TOTAL; }

extern "C" int mnrf_tcnn_forward(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                 int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                                 const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                                 const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                 float* is_mirror, float* normal, float* geo_feat, void* stream) {
    if (!table || !offsets17_host || !weights) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: null pointer");
    if (B < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: negative sample count");
    if (B == 0) return MNRF_OK;
    const bool sigma_only = flags & MNRF_SIGMA_ONLY, grad = flags & MNRF_GRAD_NORMAL;
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: need xyz or rays+z_vals");
    if (xyz && xyz_stride < (sigma_only ? 3 : 6)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: xyz_stride too small");
    if (spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: samples per ray must be >= 1");
    if (grad && !normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: GRAD_NORMAL needs the normal output");
    if (bound <= 0.f) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: bound must be positive");
    TcnnArgs A;
    A.table = table; A.weights = weights; A.B = B; A.xyz = xyz; A.xyz_stride = xyz_stride; A.rays = rays;
    A.z_vals = z_vals; A.spr = spr; A.dirs = dirs; A.dir_stride = dir_stride; A.bound = bound;
    for (int l = 0; l < NL; ++l) {
        // gridencoder.cu:150 evaluates exp2f(level*S)*H - 1 on the device; here the per-level scale is fixed on the
        // host in double precision so that every implementation (kernel, oracle) sees bit-identical scales
        A.scale[l] = (float)(exp2((double)l * log2_per_level_scale) * (double)base_resolution - 1.0);
        A.res[l] = (unsigned)ceilf(A.scale[l]) + 1u;
        A.off[l] = (unsigned)offsets17_host[l];
    }
    A.off[NL] = (unsigned)offsets17_host[NL];
    A.sigma = sigma; A.rgb = rgb; A.pred_normal = pred_normal; A.is_mirror = is_mirror; A.normal = normal; A.geo_feat = geo_feat;
    const dim3 grid((unsigned)((B + TPB - 1) / TPB)), block(TPB);
    const size_t lds = (VEC_OFF + 64 * TPB) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (sigma_only && !grad) hipLaunchKernelGGL((tcnn_kernel<true, false>), grid, block, lds, s, A);
    else if (sigma_only) hipLaunchKernelGGL((tcnn_kernel<true, true>), grid, block, lds, s, A);
    else if (!grad) hipLaunchKernelGGL((tcnn_kernel<false, false>), grid, block, lds, s, A);
    else hipLaunchKernelGGL((tcnn_kernel<false, true>), grid, block, lds, s, A);
    return mnrf_check_launch("mnrf_tcnn_forward");
}
