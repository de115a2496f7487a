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
// One thread per sample; the 11 k weights are wave-uniform and arrive as scalar loads through the
// constant address space (mlp_layer below); the table (53 MB fp32 at bound 6) lives in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"

// per-sample outputs of the matrix-pipe kernel: -DMNRF_EXP_TCNN_OUT_NT stores them non-temporally (experiment: 201 MB of outputs
// per launch compete with the 53 MB table for the 256 MB Infinity Cache)
#ifdef MNRF_EXP_TCNN_OUT_NT
#define TOUT(p, v) __builtin_nontemporal_store((float)(v), (float*)(p))
#elif defined(MNRF_EXP_TCNN_NO_STORES)      // experiment: the MLP launch without its output traffic (the compiler must keep the arithmetic)
#define TOUT(p, v) do { const float tout_v = (v); if (tout_v == 1234.56787109375f) *(p) = tout_v; } while (0)
#else
#define TOUT(p, v) (*(p) = (v))
#endif

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
    unsigned mode[NL];            // 0: dense level, 1: hashed, power-of-two size (mask), 2: hashed, any size (modulo)
    float* sigma; float* rgb; float* pred_normal; float* is_mirror; float* normal; float* geo_feat;
    float* enc;                   // level-major encoding planes [NL][B] float2 (caller's workspace) or null
    unsigned table_f16;           // MNRF_TCNN_TABLE_F16: `table` holds half2 entries (4 B: tinycudann's storage) instead of float2
    // live row count (round 6; include/mnrf.h "live row counts on the device"): B is the CAPACITY the buffers and the plane strides
    // (Bs) are sized for, the kernels evaluate the first *n_live * spr samples (ray mode); null: all B exist
    const int* n_live;
    long long Bs;                 // = the capacity B (set by the launcher): stride of the [NL][Bs] planes
};
// first thing in every kernel that walks the samples: B becomes what exists
__device__ __forceinline__ void live_rows(TcnnArgs& A) {
    if (A.n_live) {
        long long bl = (long long)*A.n_live * A.spr;
        bl = bl < 0 ? 0 : bl;
        if (bl < A.B) A.B = bl;
    }
}

// one table entry (two features) by storage type.  The flag is a kernel argument: a wave-uniform branch next to a gather.
__device__ __forceinline__ float2 tab_fetch(const float* table, unsigned f16, unsigned long long idx) {
    if (f16) return __half22float2(((const __half2*)table)[idx]);
    return ((const float2*)table)[idx];
}

extern __shared__ __attribute__((aligned(16))) float wlds[];

constexpr int TPB = 512;                        // threads (= samples) per workgroup: 128 KB of LDS for the vector buffer, 8 waves
                                                // per CU (with the weights in LDS too: 256 threads 8.98 ms per 32768-ray chunk,
                                                // 384 threads 8.50 ms; now 6.7 ms)
constexpr int VEC_OFF = 0;                      // per-thread vector buffer vec[64][TPB] (the forward keeps no weights in LDS)
constexpr int BWD_W_FLOATS = (W_TOTAL + 3) / 4 * 4;   // the backward kernel keeps them at wlds[0 ..)
#define VEC(k) wlds[VEC_OFF + (k) * TPB + threadIdx.x]

// Layers run as "inputs in registers, loop over outputs": the output loop is NOT unrolled (a fully
// unrolled 11 k-FMA body makes hipcc hoist thousands of LDS reads and spill 2-4 KB per lane); each
// output goes to the thread's column of an LDS vector buffer and is re-loaded as the next input.
// Weight rows are wave-uniform: read through the CONSTANT address space they become s_load_dwordx4/8/16 into SGPRs and the
// FMAs take them as scalar operands -- no LDS traffic (a broadcast ds_read_b128 still moves 64 x 16 B through the LDS port
// for four FMAs per lane, which bounded this kernel).  The weights are not written while a kernel runs.
typedef const __attribute__((address_space(4))) float* cptr;
typedef const __attribute__((address_space(4))) f32x4* cptr4;
__device__ __forceinline__ cptr as_const(const float* p) { return (cptr)(unsigned long long)p; }

template <int NI>
__device__ __forceinline__ float dot_row(cptr W, const float (&in)[NI], int woff) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;       // four independent chains: one chain of NI dependent v_fmac runs at the
                                                    // VALU latency, not at its rate (measured: 6.97 -> see DESIGN 4.3)
#pragma unroll
    for (int i = 0; i < NI; i += 4) {
        const f32x4 w = *(cptr4)(W + woff + i);
        a = fmaf(w[0], in[i], a); b = fmaf(w[1], in[i + 1], b);
        c = fmaf(w[2], in[i + 2], c); d = fmaf(w[3], in[i + 3], d);
    }
    return (a + b) + (c + d);
}

// One Linear of the forward kernel: out[o] = <W[o, :], in>, o < n_out, handed to `store(o, value)`.  The weights are
// consumed in chunks of 32 floats (= half a 64-wide row, one 32-wide row, two 16-wide rows) and chunk j+1 is requested
// BEFORE chunk j is used: an s_load issued when its data are needed costs its whole latency per row (scalar loads return
// out of order, so the wait is always lgkmcnt(0)), and one or two waves per SIMD do not hide it.  The body handles two
// chunks per iteration so that both buffers and every register-array index are static.
__device__ __forceinline__ void load_chunk(cptr W, int off, f32x4 (&dst)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = *(cptr4)(W + off + 4 * q);
}
// Explicit 2-wide FMAs: (w[q].xy, w[q].zw) are even-aligned SGPR pairs of the s_load destination and `in` is kept as
// float2s, so every v_pk_fma_f32 takes its operands where they are.  (Left to the SLP vectoriser, the scalar FMAs were
// paired across odd SGPR boundaries and each pair cost three s_mov_b32: the CU's one scalar unit ran 93 % busy
// -- SQ_ACTIVE_INST_SCA, profiles/r01s_pmc_tcnn -- and bounded the kernel.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N4>
__device__ __forceinline__ float dot_chunk(const f32x4* w, const f32x2* in2) {      // N4 float4s of weights
    f32x2 a = {0.f, 0.f}, b = {0.f, 0.f}, c = {0.f, 0.f}, d = {0.f, 0.f};     // four independent chains of packed FMAs
#pragma unroll
    for (int q = 0; q < N4; q += 2) {
        a = __builtin_elementwise_fma(w[q].xy, in2[2 * q], a);
        b = __builtin_elementwise_fma(w[q].zw, in2[2 * q + 1], b);
        c = __builtin_elementwise_fma(w[q + 1].xy, in2[2 * q + 2], c);
        d = __builtin_elementwise_fma(w[q + 1].zw, in2[2 * q + 3], d);
    }
    a = (a + b) + (c + d);
    return a.x + a.y;
}
template <int NI, class Store>
__device__ __forceinline__ void mlp_layer(cptr W, int woff, int n_out, const float (&in)[NI], Store store) {
    static_assert(NI == 16 || NI == 32 || NI == 64, "row widths of the weight blob");
    const int nchunks = n_out * NI / 32;           // even for every layer routed here
    f32x2 in2[NI / 2];
#pragma unroll
    for (int k = 0; k < NI / 2; ++k) in2[k] = f32x2{in[2 * k], in[2 * k + 1]};
    f32x4 A[8], B[8];
    load_chunk(W, woff, A);
    float half = 0.f;
#pragma unroll 1
    for (int j = 0; j < nchunks; j += 2) {
        load_chunk(W, woff + 32 * (j + 1), B);
        if constexpr (NI == 64) half = dot_chunk<8>(A, in2);
        else if constexpr (NI == 32) store(j, dot_chunk<8>(A, in2));
        else { store(2 * j, dot_chunk<4>(A, in2)); store(2 * j + 1, dot_chunk<4>(A + 4, in2)); }
        load_chunk(W, woff + 32 * (j + 2 < nchunks ? j + 2 : j), A);
        if constexpr (NI == 64) store(j >> 1, half + dot_chunk<8>(B, in2 + 16));
        else if constexpr (NI == 32) store(j + 1, dot_chunk<8>(B, in2));
        else { store(2 * j + 2, dot_chunk<4>(B, in2)); store(2 * j + 3, dot_chunk<4>(B + 4, in2)); }
    }
}

// the same with the weights in LDS at wlds[0 ..) (backward kernel: its transposed passes read them from LDS anyway, and the
// scalar variant measured slower there, 4.68 vs 4.38 ms per training step)
template <int NI>
__device__ __forceinline__ float dot_row_lds(const float (&in)[NI], int woff) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i += 4) {
        const f32x4 w = *(const f32x4*)(wlds + woff + i);
        a = fmaf(w[0], in[i], a); b = fmaf(w[1], in[i + 1], b);
        c = fmaf(w[2], in[i + 2], c); d = fmaf(w[3], in[i + 3], d);
    }
    return (a + b) + (c + d);
}

template <int N>
__device__ __forceinline__ void load_vec(float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = VEC(k);
}

__device__ __forceinline__ unsigned grid_index(unsigned x, unsigned y, unsigned z, unsigned hsize, unsigned res, unsigned mode) {
    // get_grid_index (gridencoder.cu:68-89): dense while the running stride fits, else the spatial hash, then `% hsize`.
    // Which of the two a LEVEL takes is fixed on the host (level_mode below: dense iff (res+1)^3 <= hsize, and then the
    // index is already < hsize), so the per-corner code has no branch and no integer division: hashed levels of the
    // reference configuration hold 2^19 entries (a mask); any other size keeps the modulo.
    if (mode == 0) return x + (res + 1) * (y + (res + 1) * z);
    const unsigned h = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return mode == 1 ? (h & (hsize - 1)) : (h % hsize);
}

// one level of the encoding: the two features and (GRAD) their derivatives w.r.t. the [0,1] coordinates
template <bool GRAD>
__device__ __forceinline__ void encode_level_p(const float* table, float scale, unsigned res, unsigned off0, unsigned hsize, unsigned mode,
                                               const float (&u)[3], bool oob, float& a0, float& a1, float (&g0)[3], float (&g1)[3],
                                               unsigned f16 = 0u) {
    unsigned pg[3];
    float fr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = (oob ? 0.5f : u[a]) * scale + 0.5f;     // (outside the box: a valid cell, weight 0 below)
        const float fl = floorf(pos);
        pg[a] = (unsigned)fl;
        fr[a] = pos - fl;
    }
    const float in_box = oob ? 0.f : 1.f;
    a0 = 0.f; a1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g0[a] = 0.f; g1[a] = 0.f; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wx = (c & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (c & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (c & 4) ? fr[2] : 1.f - fr[2];
        // unconditional loads: the eight gathers of a level (and of the next levels, the loop is unrolled) go out together
        const float2 v = tab_fetch(table, f16, off0 + grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, mode));
        const float w = wx * wy * wz * in_box;
        a0 += w * v.x; a1 += w * v.y;
        if (GRAD) {
            const float sx = ((c & 1) ? scale : -scale) * wy * wz * in_box;
            const float sy = ((c & 2) ? scale : -scale) * wx * wz * in_box;
            const float sz = ((c & 4) ? scale : -scale) * wx * wy * in_box;
            g0[0] += sx * v.x; g1[0] += sx * v.y;
            g0[1] += sy * v.x; g1[1] += sy * v.y;
            g0[2] += sz * v.x; g1[2] += sz * v.y;
        }
    }
}

template <bool GRAD>
__device__ __forceinline__ void encode_level(const TcnnArgs& A, int lv, const float (&u)[3], bool oob, float& a0, float& a1,
                                             float (&g0)[3], float (&g1)[3]) {
    encode_level_p<GRAD>(A.table, A.scale[lv], A.res[lv], A.off[lv], A.off[lv + 1] - A.off[lv], A.mode[lv], u, oob, a0, a1, g0, g1, A.table_f16);
}

template <bool SIGMA_ONLY, bool GRAD>
__global__ __launch_bounds__(TPB) void tcnn_kernel(TcnnArgs A) {
    live_rows(A);
    if ((long long)blockIdx.x * TPB >= A.B) return;
    const cptr WC = as_const(A.weights);
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
#pragma unroll 4
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
        mlp_layer(WC, W_S0, 64, in, [&](int o, float pre) {
            relu_bits |= (unsigned long long)(pre > 0.f) << o;
            VEC(o) = fmaxf(pre, 0.f);
        });
    }
    float geo[16];
    {
        float in[64];
        load_vec(in);
        mlp_layer(WC, W_S1, 16, in, [&](int o, float v) { VEC(o) = v; });
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
            const float s = ((relu_bits >> k) & 1ull) ? WC[W_S1 + k] : 0.f;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                const f32x4 w = *(cptr4)(WC + W_S0 + k * 32 + e);
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
        mlp_layer(WC, W_N0, 64, geo, [&](int o, float v) { VEC(o) = fmaxf(v, 0.f); });
        float hn[64];
        load_vec(hn);
        float v[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) v[o] = dot_row(WC, hn, W_N1 + o * 64);
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
        mlp_layer(WC, W_C0, 64, in, [&](int o, float v) { VEC(o) = fmaxf(v, 0.f); });
        float c1[64];
        load_vec(c1);
        mlp_layer(WC, W_C1, 64, c1, [&](int o, float v) { VEC(o) = fmaxf(v, 0.f); });
        load_vec(c1);
        if (A.rgb && live) {
#pragma unroll
            for (int k = 0; k < 3; ++k) A.rgb[i * 3 + k] = 1.f / (1.f + expf(-dot_row(WC, c1, W_C2 + k * 64)));
        }
    }
    // ---- mirror probability: 15 -> 32 LeakyReLU(0.01) -> 1 sigmoid, with biases (141-149)
    if (A.is_mirror) {
        mlp_layer(WC, W_M0, 32, geo, [&](int o, float dotv) {
            const float v = dotv + WC[B_M0 + o];
            VEC(o) = v > 0.f ? v : 0.01f * v;
        });
        float hm[32];
        load_vec(hm);
        if (live) A.is_mirror[i] = 1.f / (1.f + expf(-(dot_row(WC, hm, W_M1) + WC[B_M1])));
    }
}
#undef VEC

// ------------------------------------------------------------------------------------------------------------
// The same field with its MLPs on the matrix pipe (inference / training forward).
//
// tcnn_kernel above spends most of its time on 11 k scalar-operand FMAs per sample (VALU 31 % busy, waiting on the
// weight stream through the scalar cache, profiles/r01s_pmc_tcnn).  Here a wavefront owns NG GROUPS of 16 samples
// and evaluates every Linear transposed, Out^T[rows x 16 samples] = W . In^T, on v_mfma_f32_16x16x32_f16 with the hi/lo
// f16 split of the 8x256 field kernel (three products per tile, fp32 accumulation: ~2^-20 relative, so the results stay
// within fp32 noise of the VALU kernel; tinycudann itself runs these MLPs in plain fp16).
//     (Here: 8 waves x 2 groups per workgroup iteration, one workgroup per CU: with <= 128 registers for a second one the
//     64 gathers a lane keeps in flight spill.)
//   * The gathers run with one lane per sample (phase 1: every gather instruction of a wave addresses one level for 64
//     consecutive samples; a first version in which the four lanes of a sample shared its levels -- no exchange needed --
//     made 16 samples x 4 levels out of every instruction and gathered 15 % slower) and hand the 32 features to the MLP
//     lanes through a per-wave LDS area.  Lane (j = l & 15, g = l >> 4) then works for sample j of each group: features
//     8g .. 8g+7 are its slots of the B operand of sigma_net.0 and, as in mnrf_layout.h, accumulator register r of row
//     block nb (row 16 nb + 4 g + r) is its slot of the next layer's B operand: activations stay in their lane.
//   * The weights (11 204 floats) become 33 hi/lo tile pairs of 2 KiB in LDS, built by the workgroup itself from the fp32
//     blob in the column order those slots imply (tile_value below); workgroups are persistent (one per CU), so this is
//     paid once per ~24 000 samples.  A tile pair is read once per four groups: 66 KiB of LDS reads per 64 samples.
//   * 29 tile pairs x 3 products x 4 groups = 348 MFMAs per 64 samples: 0.25 ms of matrix-pipe time per 6.29 M-sample
//     launch -- the kernel is left with its 128 gathers per sample.
//   * Launches that also want the density-gradient normal (a second gather pass) or sigma only stay on tcnn_kernel: measured
//     per 6.29 M samples, full + normal 7.8 (this structure, spilling at 256 registers) vs 7.1 ms, sigma only 1.11 vs 0.98 ms
//     per 2.1 M; the full evaluation without the normal -- the fine pass of every eval render -- 2.83 vs 5.46 ms.
namespace mf {

// Waves per workgroup.  Round 6: EIGHT (was six).  A CU has four SIMDs and a workgroup's waves are dealt to them in turn: six waves
// leave two SIMDs with two waves and two with one, and the kernel -- bound by its own VALU / MFMA instruction stream, not by memory
//  -- then runs at the pace of the loaded pair.  One 32768-ray chunk's fine pass, encoding + MLP launch, alternating
// libraries on one box (scripts/exp_tcnn_mlp_parts.py, profiles/r06_tcnn_waves.txt): 6 waves 2.24 ms (hi/lo) / 1.76 (f16 MLPs),
// 4 waves 2.10 / 1.72, **8 waves 2.05 / 1.63**, 10 waves 2.12 / 1.73, 12 waves 2.02 / 1.66, 16 waves 2.03 / 1.62: every multiple
// of four beats its neighbours; 8 is the smallest workgroup of the fast group for both arithmetics.
#ifndef MNRF_EXP_TCNN_WAVES
#define MNRF_EXP_TCNN_WAVES 8
#endif
#ifndef MNRF_EXP_TCNN_MINWG
#define MNRF_EXP_TCNN_MINWG 2
#endif
constexpr int WAVES = MNRF_EXP_TCNN_WAVES;
#ifndef MNRF_EXP_TCNN_NG
#define MNRF_EXP_TCNN_NG 2
#endif
constexpr int NG = MNRF_EXP_TCNN_NG;                // groups of 16 samples per wave iteration (4: 800 B/lane of spills at 256 registers)
constexpr int TILE = WAVES * NG * 16;               // samples per workgroup iteration
constexpr int NT_FWD = 29;                          // tile pairs
constexpr int PAIR_B = 2048;
constexpr int LDS_TILES = NT_FWD * PAIR_B;          // 58 KiB
constexpr int LDS_WS1 = LDS_TILES;                  // sigma_net.1 row 0 (64 floats) + mirror biases (33 floats)
constexpr int LDS_BYTES = LDS_WS1 + 128 * 4;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum Kind : int { K_ENC = 0, K_H = 1, K_GEO = 2, K_SHGEO = 3 };
struct TileDesc { short w_off, n_true, ld, nb, T, kind; };
// consumption order: S0 (4), S1 (2), N0 (4), N1 (2), M0 (2), M1 (1), C0 (4), C1 (8: T-major), C2 (2)
__device__ __forceinline__ TileDesc tile_desc(int id) {
    if (id < 4) return TileDesc{W_S0, 64, 32, (short)id, 0, K_ENC};
    if (id < 6) return TileDesc{W_S1, 16, 64, 0, (short)(id - 4), K_H};
    if (id < 10) return TileDesc{W_N0, 64, 16, (short)(id - 6), 0, K_GEO};
    if (id < 12) return TileDesc{W_N1, 3, 64, 0, (short)(id - 10), K_H};
    if (id < 14) return TileDesc{W_M0, 32, 16, (short)(id - 12), 0, K_GEO};
    if (id < 15) return TileDesc{W_M1, 1, 32, 0, 0, K_H};
    if (id < 19) return TileDesc{W_C0, 64, 32, (short)(id - 15), 0, K_SHGEO};
    if (id < 27) return TileDesc{W_C1, 64, 64, (short)((id - 19) & 3), (short)((id - 19) >> 2), K_H};
    return TileDesc{W_C2, 3, 64, 0, (short)(id - 27), K_H};
}
// element (lane, e) of the A operand of a tile: W[row 16 nb + (lane & 15)][column of slot (T, lane >> 4, e)]
__device__ __forceinline__ float tile_value(const float* W, int id, int lane, int e) {
    const TileDesc d = tile_desc(id);
    const int i = lane & 15, g = lane >> 4;
    const int kh = 32 * d.T + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));      // slot -> previous layer's row (K_H)
    const int row = 16 * d.nb + i;
    if (row >= d.n_true) return 0.f;
    int col = -1;
    if (d.kind == K_ENC) col = 8 * g + e;
    else if (d.kind == K_H) col = kh;
    else {      // K_GEO / K_SHGEO: slots 0-3 = the lane's four rows of sigma_net.1's output (row 0 is sigma: no weight)
        if (e < 4) { const int rr = 4 * g + e; col = rr >= 1 ? (d.kind == K_SHGEO ? 16 : 0) + rr - 1 : -1; }
        else if (d.kind == K_SHGEO) col = 4 * g + (e - 4);                  // spherical-harmonics components 4g .. 4g+3
    }
    if (col < 0 || col >= d.ld) return 0.f;
    return W[d.w_off + row * d.ld + col];
}

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    // (the two subtractions as one v_pk_add_f32 -- 1811 -> 1729 VALU instructions in the kernel -- measured no faster, round 6:
    //  the launch is bound by the dependent MFMA -> convert -> MFMA chain of a 16-sample group, not by VALU issue)
    const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// MODE of the matrix-pipe kernel: 0 = every fp32 operand as a hi/lo f16 pair, three MFMAs per product (~fp32 accuracy);
// 1 = single-pass f16 (MNRF_TCNN_F16): operands rounded to nearest f16, ONE MFMA per product, fp32 accumulation -- what
// tinycudann's FullyFusedMLP / the reference's precision=16 trainer compute (models/mirror_nerf_tcnn.py:36-49, train.py:586)
template <int MODE>
__device__ __forceinline__ void to_b(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (MODE == 1) {
            typedef _Float16 h2t __attribute__((ext_vector_type(2)));
            const h2t h = {(_Float16)v[2 * w], (_Float16)v[2 * w + 1]};      // round to nearest even
            hi[w] = __builtin_bit_cast(unsigned, h);
            lo[w] = 0u;
        } else {
            unsigned h, l; split_pair(v[2 * w], v[2 * w + 1], h, l); hi[w] = h; lo[w] = l;
        }
    }
}
__device__ __forceinline__ f32x4 mfma3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, al), __builtin_bit_cast(h8, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bl), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bh), c, 0, 0, 0);
}

extern __shared__ __attribute__((aligned(16))) char smem_mf[];

__device__ __forceinline__ void read_pair(int id, int lane, u32x4& ah, u32x4& al) {
    const char* p = smem_mf + id * PAIR_B + lane * 16;
    ah = *(const u32x4*)p;
    al = *(const u32x4*)(p + 1024);
}

// One Linear for the NG groups of a wave: NB row blocks, NTK k-steps, tiles id0 .. in T-major order (id0 + T*NB + nb)
template <int MODE, int NB, int NTK>
__device__ __forceinline__ void layer(int id0, int lane, const u32x4 (&bh)[NG][NTK], const u32x4 (&bl)[NG][NTK], f32x4 (&acc)[NG][NB]) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[gi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NTK; ++T)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            u32x4 ah, al;
            if (MODE == 1) {
                ah = *(const u32x4*)(smem_mf + (id0 + T * NB + nb) * PAIR_B + lane * 16);
#pragma unroll
                for (int gi = 0; gi < NG; ++gi)
                    acc[gi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), __builtin_bit_cast(h8, bh[gi][T]), acc[gi][nb], 0, 0, 0);
            } else {
                read_pair(id0 + T * NB + nb, lane, ah, al);
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) acc[gi][nb] = mfma3(ah, al, bh[gi][T], bl[gi][T], acc[gi][nb]);
            }
        }
}
// accumulators of 2*NTK row blocks -> the next layer's NTK B operands (ACT: 0 none, 1 relu, 2 leaky relu 0.01 with bias)
template <int MODE, int NTK, int ACT>
__device__ __forceinline__ void next_b(const f32x4 (&acc)[NG][2 * NTK], u32x4 (&bh)[NG][NTK], u32x4 (&bl)[NG][NTK], const float* bias4) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int T = 0; T < NTK; ++T) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = acc[gi][2 * T + (e >> 2)][e & 3];
                if (ACT == 2) { a += bias4[4 * (2 * T + (e >> 2)) + (e & 3)]; a = a > 0.f ? a : 0.01f * a; }
                v[e] = ACT == 1 ? fmaxf(a, 0.f) : a;
            }
            to_b<MODE>(v, bh[gi][T], bl[gi][T]);
        }
}

// A lane's levels depend on its lane group: their parameters live in registers (kernel arguments cannot be indexed per
// lane).  Level kinds here: dense, or hashed with a power-of-two size (a mask); a hashed level of any other size needs an
// integer modulo per corner -- mnrf_tcnn_forward sends such configurations to the VALU kernel.
struct LevelP { float scale; unsigned res, off0, hmask; bool dense; };
template <bool GRAD>
__device__ __forceinline__ void encode_level_p(const float* table, const LevelP& L, const float (&u)[3], bool oob, float& a0, float& a1,
                                               float (&g0)[3], float (&g1)[3], unsigned f16 = 0u) {
    unsigned pg[3];
    float fr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = (oob ? 0.5f : u[a]) * L.scale + 0.5f;
        const float fl = floorf(pos);
        pg[a] = (unsigned)fl;
        fr[a] = pos - fl;
    }
    const float in_box = oob ? 0.f : 1.f;
    a0 = 0.f; a1 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g0[a] = 0.f; g1[a] = 0.f; }
    const unsigned r1 = L.res + 1;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wx = (c & 1) ? fr[0] : 1.f - fr[0];
        const float wy = (c & 2) ? fr[1] : 1.f - fr[1];
        const float wz = (c & 4) ? fr[2] : 1.f - fr[2];
        const unsigned x = pg[0] + (c & 1), y = pg[1] + ((c >> 1) & 1), z = pg[2] + ((c >> 2) & 1);
        const unsigned lin = x + r1 * (y + r1 * z);
        const unsigned hsh = ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & L.hmask;
        const float2 v = tab_fetch(table, f16, L.off0 + (L.dense ? lin : hsh));
        const float w = wx * wy * wz * in_box;
        a0 += w * v.x; a1 += w * v.y;
        if (GRAD) {
            const float sx = ((c & 1) ? L.scale : -L.scale) * wy * wz * in_box;
            const float sy = ((c & 2) ? L.scale : -L.scale) * wx * wz * in_box;
            const float sz = ((c & 4) ? L.scale : -L.scale) * wx * wy * in_box;
            g0[0] += sx * v.x; g1[0] += sx * v.y;
            g0[1] += sy * v.x; g1[1] += sy * v.y;
            g0[2] += sz * v.x; g1[2] += sz * v.y;
        }
    }
}

// Level-major encoding (the default of the matrix-pipe path when the caller hands a workspace): one thread per (sample, level),
// the grid's y dimension is the level, so that the whole device works on ONE level at a time.  Why: the 12 hashed levels are
// 4 MB each and a sample's corners land on random lines of them; with every wave walking all 16 levels (the in-kernel
// encoding) the 4 MB L2 of an XCD sees 48 MB of table at once, misses on 54 % of the lines and the kernel streams 20.7 GB per
// 6.29 M samples over the fabric (3.2x the 1 KiB of table entries a sample needs, at 7.5 TB/s: profiles/r04b_pmc_tcnn).
// Level by level the working set is one level.  Costs 128 B per sample written and read once (the planes).
// Thread -> sample map of a workgroup (ray mode).  ENC_PATCH_RAYS = 1: 256 consecutive samples of the flat (ray-major) order -- one or
// two rays, 256 depths.  ENC_PATCH_RAYS = R > 1: a PATCH of R consecutive rays x 256 / R consecutive depths: neighbouring rays of a
// chunk are neighbouring pixels, whose samples at equal depth share grid cells up to the middle levels, so the lanes of a wave
// instruction fall on fewer distinct cache lines (the launch is bound by the vector L1's requests to the L2, not by bytes: 4.3).
// Measured on one 32768-ray chunk of the bench frame, fine pass (scripts/exp_tcnn_encode.py, alternating libraries, identical
// outputs): 1.68 ms per launch flat, 1.55 with 4 rays x 64 depths, 1.32 with 8 x 32, 1.28 with 16 x 16, **1.27 with 32 x 8** (a wave = 8
// rays x 8 depths), 1.45 with 64 x 4.  Falls back to the flat map when the shape does not tile (spr % 8, rays % 32) or with xyz input.
#ifndef MNRF_EXP_ENC_PATCH_RAYS
#define MNRF_EXP_ENC_PATCH_RAYS 32
#endif
constexpr int ENC_PATCH_RAYS = MNRF_EXP_ENC_PATCH_RAYS;
__global__ __launch_bounds__(256) void tcnn_encode_kernel(TcnnArgs A) {
    live_rows(A);
    const int lv = blockIdx.y;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (ENC_PATCH_RAYS > 1 && !A.xyz && A.spr % (256 / ENC_PATCH_RAYS) == 0 && (A.Bs / A.spr) % ENC_PATCH_RAYS == 0) {
        constexpr int PS = 256 / ENC_PATCH_RAYS;                  // depths per patch
        const int groups = A.spr / PS;                            // patches along a ray
        long long patch = blockIdx.x;
#ifndef MNRF_EXP_ENC_NO_XCD
        // workgroups go round-robin over the 8 XCDs, each with its own L2: XCD k takes the k-th contiguous eighth of the patches (a
        // contiguous range of rays with all their depths) instead of every eighth patch.  Same chunk as above, alternating libraries:
        // 1.27 -> 1.24 ms per launch with the float2 table, 1.23 -> 1.13 ms with the half2 table.  (Patches ordered depth slab by
        // depth slab instead: 1.26 / 1.16 ms without this map, 1.37 / 1.30 with it.)
        if (gridDim.x % 8 == 0) patch = (patch % 8) * (gridDim.x / 8) + patch / 8;
#endif
        const long long ray0 = patch / groups * ENC_PATCH_RAYS;
        const int s0 = (int)(patch % groups) * PS;
        // lanes: depth fastest within PS, then ray -- a wave of 64 covers 64 / PS rays x PS depths (PS < 64) or one ray (PS >= 64)
        const int r = threadIdx.x / PS, sidx = threadIdx.x % PS;
        i = (ray0 + r) * A.spr + s0 + sidx;
    }
    if (i >= A.B) return;
    LevelP L{0.f, 0u, 0u, 0u, true};
#pragma unroll
    for (int l = 0; l < NL; ++l)
        if (l == lv) L = LevelP{A.scale[l], A.res[l], A.off[l], A.off[l + 1] - A.off[l] - 1u, A.mode[l] == 0u};
    float x[3];
    if (A.xyz) {
        const float* p = A.xyz + i * A.xyz_stride;
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
    } else {
        const float* r = A.rays + (i / A.spr) * 8;
        const float z = A.z_vals[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
    }
    float u[3];
    bool oob = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        u[a] = (x[a] + A.bound) / (2.f * A.bound);
        oob |= u[a] < 0.f || u[a] > 1.f;
    }
    float a0, a1, g0[3], g1[3];
    encode_level_p<false>(A.table, L, u, oob, a0, a1, g0, g1, A.table_f16);
    __builtin_nontemporal_store(a0, A.enc + 2 * ((long long)lv * A.Bs + i));
    __builtin_nontemporal_store(a1, A.enc + 2 * ((long long)lv * A.Bs + i) + 1);
}

template <int MODE, bool PLANES = false>
__global__ __launch_bounds__(64 * WAVES, MNRF_EXP_TCNN_MINWG) void tcnn_mfma_kernel(TcnnArgs A, int n_tiles) {
    live_rows(A);
    if (A.n_live) n_tiles = (int)((A.B + TILE - 1) / TILE);
    if (n_tiles <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // ---- weight tiles (hi | lo) and the few fp32 rows used outside the GEMMs
    {
        constexpr int NT = NT_FWD;
        for (int q = tid; q < NT * 512; q += 64 * WAVES) {   // one (tile, lane, half) per step
            const int id = q >> 9, l = (q >> 3) & 63, e = q & 7;
            const float w = tile_value(A.weights, id, l, e);
            const _Float16 hi = (_Float16)w;
            _Float16* dst = (_Float16*)(smem_mf + id * PAIR_B + l * 16) + e;
            dst[0] = hi;
            dst[512] = (_Float16)(w - (float)hi);
        }
        float* ws1 = (float*)(smem_mf + LDS_WS1);
        if (tid < 64) ws1[tid] = A.weights[W_S1 + tid];
        if (tid < 32) ws1[64 + tid] = A.weights[B_M0 + tid];
        if (tid == 0) ws1[96] = A.weights[B_M1];
    }
    __syncthreads();
    const float* ws1 = (const float*)(smem_mf + LDS_WS1);
    LevelP lvl[4];                          // levels 4g .. 4g+3 (static indices into the argument arrays, selected per lane)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lvl[q] = LevelP{0.f, 0u, 0u, 0u, true};
#pragma unroll
        for (int lv = 0; lv < NL; ++lv)
            if (lv == 4 * g + q)
                lvl[q] = LevelP{A.scale[lv], A.res[lv], A.off[lv], A.off[lv + 1] - A.off[lv] - 1u, A.mode[lv] == 0u};
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        long long idx[NG];
        bool live[NG];
        float u[NG][3];
        bool oob[NG];
        u32x4 eh[NG][1], el[NG][1];
        // ---- positions and the lane's four levels of the encoding
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            long long i = (long long)tile * TILE + wave * (NG * 16) + gi * 16 + j;
            live[gi] = i < A.B;
            if (!live[gi]) i = A.B - 1;
            idx[gi] = i;
            float x[3];
            if (A.xyz) {
                const float* p = A.xyz + i * A.xyz_stride;
                x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
            } else {
                const float* r = A.rays + (i / A.spr) * 8;
                const float z = A.z_vals[i];
#pragma unroll
                for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
            }
            oob[gi] = false;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                u[gi][a] = (x[a] + A.bound) / (2.f * A.bound);
                oob[gi] |= u[gi][a] < 0.f || u[gi][a] > 1.f;
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            float f8[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float g0[3], g1[3];
                if (PLANES) {      // written level by level by tcnn_encode_kernel
#ifdef MNRF_EXP_TCNN_NO_PLANE_LOADS      // experiment: the MLP launch without its input traffic
                    const float2 v = make_float2(0.01f * (float)((idx[gi] + q) & 63), 0.02f * (float)((idx[gi] >> 3) & 31));
#else
                    const float2 v = ((const float2*)A.enc)[(long long)(4 * g + q) * A.Bs + idx[gi]];
#endif
                    f8[2 * q] = v.x; f8[2 * q + 1] = v.y;
                } else {
                    encode_level_p<false>(A.table, lvl[q], u[gi], oob[gi], f8[2 * q], f8[2 * q + 1], g0, g1, A.table_f16);
                }
            }
            to_b<MODE>(f8, eh[gi][0], el[gi][0]);
        }
        // ---- sigma net: 32 -> 64 (ReLU) -> 16
        u32x4 sh_[NG][2], sl_[NG][2];
        {
            f32x4 acc[NG][4];
            layer<MODE, 4, 1>(0, lane, eh, el, acc);
            next_b<MODE, 2, 1>(acc, sh_, sl_, nullptr);
        }
        f32x4 s1[NG][1];      // row 0 = sigma (lane group 0, register 0), rows 1..15 = geo_feat
        layer<MODE, 1, 2>(4, lane, sh_, sl_, s1);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (!live[gi]) continue;
            if (A.sigma && g == 0) TOUT(A.sigma + idx[gi], s1[gi][0][0]);
            if (A.geo_feat) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * g + r;
                    if (row >= 1) TOUT(A.geo_feat + idx[gi] * 15 + row - 1, s1[gi][0][r]);
                }
            }
        }
        // ---- B operand of the three heads that read geo_feat: slots 0-3 = the lane's rows of sigma_net.1's output
        u32x4 qh[NG][1], ql[NG][1];
        {
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const float v[8] = {s1[gi][0][0], s1[gi][0][1], s1[gi][0][2], s1[gi][0][3], 0.f, 0.f, 0.f, 0.f};
                to_b<MODE>(v, qh[gi][0], ql[gi][0]);
            }
        }
        // ---- predicted normal: 15 -> 64 (ReLU) -> 3, l2-normalised
        if (A.pred_normal) {
            u32x4 nh[NG][2], nl[NG][2];
            {
                f32x4 acc[NG][4];
                layer<MODE, 4, 1>(6, lane, qh, ql, acc);
                next_b<MODE, 2, 1>(acc, nh, nl, nullptr);
            }
            f32x4 o3[NG][1];
            layer<MODE, 1, 2>(10, lane, nh, nl, o3);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const float v0 = o3[gi][0][0], v1 = o3[gi][0][1], v2 = o3[gi][0][2];
                const float inv = 1.f / sqrtf(fmaxf(v0 * v0 + v1 * v1 + v2 * v2, EPS32));
                if (live[gi] && g == 0) { float* o = A.pred_normal + idx[gi] * 3; TOUT(o, v0 * inv); TOUT(o + 1, v1 * inv); TOUT(o + 2, v2 * inv); }
            }
        }
        // ---- mirror probability: 15 -> 32 LeakyReLU(0.01) -> 1 sigmoid, with biases
        if (A.is_mirror) {
            u32x4 mh[NG][1], ml[NG][1];
            {
                f32x4 acc[NG][2];
                layer<MODE, 2, 1>(12, lane, qh, ql, acc);
                float b4[8];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[4 * nb + r] = ws1[64 + 16 * nb + 4 * g + r];
                next_b<MODE, 1, 2>(acc, mh, ml, b4);
            }
            f32x4 o1[NG][1];
            layer<MODE, 1, 1>(14, lane, mh, ml, o1);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
                if (live[gi] && g == 0) TOUT(A.is_mirror + idx[gi], 1.f / (1.f + expf(-(o1[gi][0][0] + ws1[96]))));
        }
        // ---- colour: cat[SH4(d), geo_feat] -> 64 -> 64 -> 3 sigmoid; slots 4-7 = SH components 4g .. 4g+3
        if (A.rgb) {
            u32x4 ch[NG][1], cl[NG][1];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const float* dp;
                if (A.xyz) dp = A.xyz + idx[gi] * A.xyz_stride + 3;
                else { const long long ray = idx[gi] / A.spr; dp = A.dirs ? A.dirs + ray * A.dir_stride : A.rays + ray * 8 + 3; }
                const float X = dp[0], Y = dp[1], Z = dp[2];
                const float xy = X * Y, xz = X * Z, yz = Y * Z, x2 = X * X, y2 = Y * Y, z2 = Z * Z;
                float sh4[4];
                if (g == 0) {
                    sh4[0] = 0.28209479177387814f; sh4[1] = -0.48860251190291987f * Y;
                    sh4[2] = 0.48860251190291987f * Z; sh4[3] = -0.48860251190291987f * X;
                } else if (g == 1) {
                    sh4[0] = 1.0925484305920792f * xy; sh4[1] = -1.0925484305920792f * yz;
                    sh4[2] = 0.94617469575755997f * z2 - 0.31539156525251999f; sh4[3] = -1.0925484305920792f * xz;
                } else if (g == 2) {
                    sh4[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; sh4[1] = 0.59004358992664352f * Y * (-3.0f * x2 + y2);
                    sh4[2] = 2.8906114426405538f * xy * Z; sh4[3] = 0.45704579946446572f * Y * (1.0f - 5.0f * z2);
                } else {
                    sh4[0] = 0.3731763325901154f * Z * (5.0f * z2 - 3.0f); sh4[1] = 0.45704579946446572f * X * (1.0f - 5.0f * z2);
                    sh4[2] = 1.4453057213202769f * Z * (x2 - y2); sh4[3] = 0.59004358992664352f * X * (-x2 + 3.0f * y2);
                }
                const float v[8] = {s1[gi][0][0], s1[gi][0][1], s1[gi][0][2], s1[gi][0][3], sh4[0], sh4[1], sh4[2], sh4[3]};
                to_b<MODE>(v, ch[gi][0], cl[gi][0]);
            }
            u32x4 h1[NG][2], l1[NG][2];
            {
                f32x4 acc[NG][4];
                layer<MODE, 4, 1>(15, lane, ch, cl, acc);
                next_b<MODE, 2, 1>(acc, h1, l1, nullptr);
            }
            u32x4 h2[NG][2], l2[NG][2];
            {
                f32x4 acc[NG][4];
                layer<MODE, 4, 2>(19, lane, h1, l1, acc);
                next_b<MODE, 2, 1>(acc, h2, l2, nullptr);
            }
            f32x4 o3[NG][1];
            layer<MODE, 1, 2>(27, lane, h2, l2, o3);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
                if (live[gi] && g == 0) {
                    float* o = A.rgb + idx[gi] * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) TOUT(o + k, 1.f / (1.f + expf(-o3[gi][0][k])));
                }
        }
    }
}

}  // namespace mf

// ------------------------------------------------------------------------------------------------------------
// Backward of the hash-grid field (training of config 5; autograd equivalent of loss.backward() through
// models/mirror_nerf_tcnn.py:220-259, models/gridencoder/src/gridencoder.cu:275-380 [kernel_grid_backward: atomicAdd of
// w * dL/dout into the table gradient] and models/shencoder/src/shencoder.cu:81-160 [dL/d direction]).
//
// One thread per sample, 256-sample tiles, persistent workgroups (one per CU).  Nothing is saved by the forward: the tile is
// re-evaluated head by head and every head is differentiated as soon as it has been evaluated, so that at most one layer's
// input X (<= 64 features) and half a layer's pre-activation gradient G (32 features) sit in LDS, feature-major
// [feature][sample] -- which is exactly the operand layout of the weight-gradient product dW[n][k] = sum_s G[n][s] X[k][s]:
// each wave keeps its share of the 52 16x16 gradient tiles in registers across ALL tiles of the workgroup and feeds them
// with v_mfma_f32_16x16x4_f32 (exact fp32; the four k slots of a step are four consecutive samples of one ds_read_b128 --
// the sum over samples does not care about their order).  Row stride 260 floats: the (16 rows x 4 sample quads) of a
// ds_read_b128 pass hit 16 different bank quads.  Activation gradients go through the same rows: dX = W^T G is a loop over
// the rows of G (wave-uniform weight rows from LDS, as in the forward).  At the end every workgroup adds its tiles to the
// gradient blob (<= 256 atomic adds per weight); the table gradient is scattered with global_atomic_add_f32 like the
// reference does.  This kernel is the first-order part; tcnn_bwd2_kernel adds the term through the density-gradient normal.
constexpr int BT = 256;                         // samples per tile = threads
constexpr int RS = BT + 4;                      // row stride (floats)
constexpr int XO = BWD_W_FLOATS;                     // X rows: 64
constexpr int GO = XO + 64 * RS;                // G rows: 32
constexpr int BWD_LDS_FLOATS = GO + 32 * RS;
#define XR(k) wlds[XO + (k) * RS + threadIdx.x]
#define GR(k) wlds[GO + (k) * RS + threadIdx.x]

struct TcnnBwdArgs {
    TcnnArgs f;                                  // inputs as in the forward (outputs unused)
    const float* g_sigma; const float* g_rgb; const float* g_pn; const float* g_m;   // dL/d outputs, any may be null
    const float* g_normal;                       // dL/d (density-gradient normal), (B,3) or null: second-order pass (tcnn_bwd2_kernel)
    float* d_table;                              // (entries, 2), zero-initialised by the caller; accumulated
    float* d_weights;                            // W_TOTAL floats, zero-initialised by the caller; accumulated
    float* d_xyz;                                // (B,3) or null
    float* d_dir;                                // (B,3) or null
    // Coarse levels are hit by every sample (level 0 of a 1 M-sample batch: 1 700 adds per entry): their gradient goes
    // to PRIVATE COPIES selected by the workgroup index (copy = blockIdx % copies; workgroups are dealt to the 8 XCDs
    // round robin, so a copy is only ever touched from one XCD and its lines stay in that XCD's L2 instead of
    // bouncing between the eight), summed into d_table by tcnn_fold_kernel.  Measured at 1 M samples: the scatter of
    // levels 0-3 straight into d_table took 19 ms of a 33 ms step.
    float* copies;                               // workspace (zero-initialised) or null
    int cp_n[NL];                                // copies of level lv (0: straight into d_table)
    long long cp_off[NL];                        // float offset of the level's first copy in `copies`
    int agg_levels;                              // levels [0, agg_levels) sum runs of equal cells inside the wave first
    int exp_noscatter;                           // experiment (MNRF_EXP_TCNN_NOSCATTER): skip the table atomics
    // gradient steering (models/mirror_nerf_tcnn.py:186-215, the --detach_density_* options): a head that sees geo_feat.detach()
    // still gets its own weight gradients but adds nothing to dL/d geo_feat
    unsigned cut;                                // MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD
    const float* keep_mirror;                    // per ray (per sample with xyz) or null: 0 = cut the mirror head for this ray's samples
    // MNRF_TCNN_GRAD_F16 (round 3; what tinycudann does, models/mirror_nerf_tcnn.py:36-49 under train.py:586 precision=16): the
    // levels WITHOUT private copies -- the big hashed ones, where every add is a random line -- accumulate 2^k-scaled gradients
    // in a table of half2 with ONE packed atomic per entry instead of two fp32 atomics.  The scatter is bound by the number
    // of atomics (measured: with one of the two fp32 adds compiled out the 1024-ray step takes 3.02 instead of 4.09 ms).
    __half2* g16;                                // (entries) half2, zero-initialised by the caller; null: fp32 atomics
    float g16_scale;                             // gradients are multiplied by this on the way in, divided on the way out
    // MNRF_TCNN_GRAD_FIXED (round 4, default of the Python shim): the levels without private copies accumulate BOTH features of an
    // entry with ONE 64-bit integer atomic -- two 32-bit fixed-point numbers, hi * 2^32 + lo added as a signed 64-bit integer, which
    // is exact (the carries of the low half are undone when decoding) and order-independent.  The scatter is bound by the NUMBER of
    // atomics (one u64 add instead of two fp32 adds: 4.06 -> 2.89 ms per 1024-ray step; two u64 adds: 3.77 ms).  Fixed point needs
    // a scale under which NO entry can overflow: tcnn_bwd_kernel writes dL/d encoding as planes [NL][B] float2 and, per level,
    // S = sum over the samples of max(|e0|, |e1|) -- the interpolation weights of a sample add up to 1, so no entry of the level can
    // collect more than S in either feature, however the samples collide (all rays of a batch leave one camera: 6 400 adds to one
    // entry of level 4 in a 1024-ray batch) -- and tcnn_scatter_fx_kernel scatters with the power of two that puts S at 2^30.
    // The fixed-point step is S * 2^-30: with B = 196 608 samples typically 2^-16 of the level's largest contribution.
    float2* genc;                                // planes (caller's workspace) or null
    double* ssum;                                // [NL] S per level (zeroed by the launcher)
    unsigned long long* fx;                      // (entries) packed fixed-point sums (zeroed by the launcher)
};

// Round 4, measured and NOT kept: a level-major second launch in which every 128-byte line of d_table is only touched from one
// XCD (owner = row bits, workgroups pick their owner from HW_REG_XCC_ID, global tile queues): the atomics of a 1024-ray step took
// 2.2 ms there against 2.1 ms inside this kernel (and the 8x repeated index arithmetic another 1.8 ms) -- the fp32 atomics run
// at ~32 G/s whether their lines are L2-resident or not, so the bound is the L2's atomic units, not line traffic; without the
// scatter this kernel takes 0.92 of its 3.0 ms.  What does help is fewer atomics (run aggregation below; packed f16 pairs).
__device__ __forceinline__ void fadd(float* p, float v) { unsafeAtomicAdd(p, v); }

// acc[r] += sum over the tile's samples of G[g0 + 4*(lane>>4) + r][s] * X[x0 + (lane&15)][s]
__device__ __forceinline__ void dw_tile(f32x4& acc, int g0, int x0) {
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
    const float* gp = wlds + GO + (g0 + i) * RS + 4 * q;
    const float* xp = wlds + XO + (x0 + i) * RS + 4 * q;
#pragma unroll 4
    for (int s = 0; s < BT; s += 16) {
        const f32x4 a = *(const f32x4*)(gp + s);
        const f32x4 b = *(const f32x4*)(xp + s);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
    }
}

// acc[i] += sum_r G[r] * W[woff + r*K + i]   (this thread's column of the G rows; wave-uniform weight rows)
template <int K>
__device__ __forceinline__ void back_rows(int nrows, int woff, float (&acc)[K]) {
#pragma unroll 1
    for (int r = 0; r < nrows; ++r) {
        const float s = GR(r);
#pragma unroll
        for (int i = 0; i < K; i += 4) {
            const f32x4 w = *(const f32x4*)(wlds + woff + r * K + i);
            acc[i] = fmaf(s, w[0], acc[i]); acc[i + 1] = fmaf(s, w[1], acc[i + 1]);
            acc[i + 2] = fmaf(s, w[2], acc[i + 2]); acc[i + 3] = fmaf(s, w[3], acc[i + 3]);
        }
    }
}

// add a register tile to the gradient blob: rows n < nmax of a (N x K) matrix at woff; column `bias_col` (>= 0) of the
// tile is the bias gradient and goes to boff + n
__device__ __forceinline__ void flush_tile(float* dst, const f32x4& acc, int woff, int K, int n0, int k0, int nmax,
                                           int kmax, int bias_col, int boff) {
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * q + r, k = k0 + i;
        if (n >= nmax) continue;
        if (k < kmax) fadd(dst + woff + n * K + k, acc[r]);
        else if (k == bias_col) fadd(dst + boff + n, acc[r]);
    }
}

__device__ __forceinline__ void sh4(const float (&d)[3], float* in) {
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
}

// dL/d direction from dL/d SH (the derivative polynomials of the 16 basis functions above)
__device__ __forceinline__ void sh4_backward(const float (&d)[3], const float* g, float (&gd)[3]) {
    const float X = d[0], Y = d[1], Z = d[2];
    const float x2 = X * X, y2 = Y * Y, z2 = Z * Z;
    const float a = 0.48860251190291987f, b = 1.0925484305920792f, c = 0.94617469575755997f, e = 0.54627421529603959f;
    const float f = 0.59004358992664352f, gg = 2.8906114426405538f, h = 0.45704579946446572f, k = 0.3731763325901154f;
    const float m = 1.4453057213202769f;
    gd[0] = -a * g[3] + b * Y * g[4] - b * Z * g[7] + 2.f * e * X * g[8] - 6.f * f * X * Y * g[9] + gg * Y * Z * g[10]
            + h * (1.f - 5.f * z2) * g[13] + 2.f * m * X * Z * g[14] + 3.f * f * (y2 - x2) * g[15];
    gd[1] = -a * g[1] + b * X * g[4] - b * Z * g[5] - 2.f * e * Y * g[8] + 3.f * f * (y2 - x2) * g[9] + gg * X * Z * g[10]
            + h * (1.f - 5.f * z2) * g[11] - 2.f * m * Y * Z * g[14] + 6.f * f * X * Y * g[15];
    gd[2] = a * g[2] - b * Y * g[5] + 2.f * c * Z * g[6] - b * X * g[7] + gg * X * Y * g[10] - 10.f * h * Y * Z * g[11]
            + k * (15.f * z2 - 3.f) * g[12] - 10.f * h * X * Z * g[13] + m * (x2 - y2) * g[14];
}

// Runs of consecutive lanes in the same cell of level lv (< agg_levels) are summed into their first lane; returns whether this
// lane is the head of its run (always true on the finer levels).  Every lane of the wave must call it (shuffles).
__device__ __forceinline__ bool aggregate_runs(const TcnnBwdArgs& P, int lv, const unsigned (&pg)[3], float (&v0)[8], float (&v1)[8],
                                               bool active, int lane) {
    bool head = true;
    if (lv < P.agg_levels) {
        const unsigned key = pg[0] | (pg[1] << 10) | (pg[2] << 20);          // (agg_levels: res < 1024)
        const unsigned prev = __shfl_up(key, 1);
        const int prev_active = __shfl_up((int)active, 1);
        head = !(lane > 0 && active && prev_active && prev == key);
        const unsigned long long heads = __ballot(head);
        const int run = __popcll(heads & (~0ull >> (63 - lane)));
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int run_d = __shfl_down(run, dlt);          // (every lane executes the shuffle: no short circuit)
            const bool ok = (lane + dlt < 64) & (run_d == run);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float t0 = __shfl_down(v0[c], dlt), t1 = __shfl_down(v1[c], dlt);
                v0[c] += ok ? t0 : 0.f;
                v1[c] += ok ? t1 : 0.f;
            }
        }
    }
    return head;
}

// Table-gradient scatter of one level for this thread's sample: v0[c], v1[c] = the two feature gradients at corner c.
// Consecutive samples of a ray sit in the same cell of a coarse level (6-19 of them at level 0): on levels whose cell
// key fits 3 x 10 bits each run of lanes with equal cell is summed into its first lane (segmented suffix sum, 6 shuffle
// steps) and only that lane scatters.  Every lane of the wave must call it (shuffles).
__device__ __forceinline__ void scatter_corners(const TcnnBwdArgs& P, int lv, const unsigned (&pg)[3], float (&v0)[8], float (&v1)[8],
                                                bool active, int lane, float* dtab, unsigned hsize, unsigned res, int own = -1) {
    const TcnnArgs& A = P.f;
    const bool head = aggregate_runs(P, lv, pg, v0, v1, active, lane);
    if (active && head && !P.exp_noscatter && P.g16 && !P.cp_n[lv]) {      // (wave-uniform choice)
        __half2* h = P.g16 + A.off[lv];
        const float k = P.g16_scale;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned idx = grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, A.mode[lv]);
            unsafeAtomicAdd(h + idx, __floats2half2_rn(v0[c] * k, v1[c] * k));
        }
    } else if (active && head && !P.exp_noscatter) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned idx = grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, A.mode[lv]);
            fadd(dtab + 2ll * idx, v0[c]);
#ifndef MNRF_EXP_TCNN_HALF_SCATTER      // experiment (wrong gradients): is the scatter bound by the NUMBER of atomics?  If so, one packed
            fadd(dtab + 2ll * idx + 1, v1[c]);      // 2 x f16 atomic per entry (tinycudann's choice) would halve it.
#endif
        }
    }
}

__global__ __launch_bounds__(BT) void tcnn_bwd_kernel(TcnnBwdArgs P) {
    live_rows(P.f);
    const TcnnArgs& A = P.f;
    __shared__ double wg_ssum[NL];        // MNRF_TCNN_GRAD_FIXED: this workgroup's sum of max(|e0|, |e1|) per level
    if (threadIdx.x < NL) wg_ssum[threadIdx.x] = 0.0;
    for (int k = threadIdx.x; k < W_TOTAL; k += BT) wlds[k] = A.weights[k];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's gradient tiles (see the tile maps at their dw_tile calls)
    f32x4 t_s0[2] = {z4, z4}, t_s1 = z4, t_c0[2] = {z4, z4}, t_c1[2][2] = {{z4, z4}, {z4, z4}}, t_c2 = z4;
    f32x4 t_n0[2] = {z4, z4}, t_n1 = z4, t_m0 = z4, t_m1 = z4;
    float db_m1 = 0.f;
    const long long ntiles = (A.B + BT - 1) / BT;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        long long i = tile * BT + threadIdx.x;
        const bool live = i < A.B;
        if (!live) i = A.B - 1;
        float x[3], d[3];
        if (A.xyz) {
            const float* p = A.xyz + i * A.xyz_stride;
            x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
            d[0] = p[3]; d[1] = p[4]; d[2] = p[5];
        } else {
            const long long ray = i / A.spr;
            const float* r = A.rays + ray * 8;
            const float z = A.z_vals[i];
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
            const float* dp = A.dirs ? A.dirs + ray * A.dir_stride : r + 3;
            d[0] = dp[0]; d[1] = dp[1]; d[2] = dp[2];
        }
        const float g_sigma = (live && P.g_sigma) ? P.g_sigma[i] : 0.f;
        float g_rgb[3], g_pn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g_rgb[a] = (live && P.g_rgb) ? P.g_rgb[i * 3 + a] : 0.f;
            g_pn[a] = (live && P.g_pn) ? P.g_pn[i * 3 + a] : 0.f;
        }
        const float g_m = (live && P.g_m) ? P.g_m[i] : 0.f;
        const float keep_n = (P.cut & MNRF_CUT_NORMAL_HEAD) ? 0.f : 1.f;
        float keep_mh = (P.cut & MNRF_CUT_MIRROR_HEAD) ? 0.f : 1.f;
        if (P.keep_mirror && P.keep_mirror[A.xyz ? i : i / A.spr] == 0.f) keep_mh = 0.f;
        float u[3];
        bool oob = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            u[a] = (x[a] + A.bound) / (2.f * A.bound);
            oob |= u[a] < 0.f || u[a] > 1.f;
        }
        // ---- forward trunk: encoding -> h1 -> (sigma, geo)
        unsigned long long bits_h1 = 0;
        float geo[16];
        {
#pragma unroll 1
            for (int lv = 0; lv < NL; ++lv) {
                float a0, a1, g0[3], g1[3];
                encode_level<false>(A, lv, u, oob, a0, a1, g0, g1);
                XR(2 * lv) = a0;
                XR(2 * lv + 1) = a1;
            }
            float in[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) in[k] = XR(k);
#pragma unroll 1
            for (int o = 0; o < 64; ++o) {
                const float pre = dot_row_lds(in, W_S0 + o * 32);
                bits_h1 |= (unsigned long long)(pre > 0.f) << o;
                XR(o) = fmaxf(pre, 0.f);
            }
            float h1[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) h1[k] = XR(k);
#pragma unroll 1
            for (int o = 1; o < 16; ++o) GR(o) = dot_row_lds(h1, W_S1 + o * 64);
#pragma unroll
            for (int k = 0; k < 15; ++k) geo[k] = GR(1 + k);
            geo[15] = 0.f;
        }
        float g_geo[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) g_geo[k] = 0.f;

        // ---- predicted-normal head: geo -> hn (ReLU) -> v -> l2-normalise
        if (P.g_pn) {
#pragma unroll 1
            for (int o = 0; o < 64; ++o) XR(o) = fmaxf(dot_row_lds(geo, W_N0 + o * 16), 0.f);
            float hn[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) hn[k] = XR(k);
            float v[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) v[o] = dot_row_lds(hn, W_N1 + o * 64);
            const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            const float inv = 1.f / sqrtf(fmaxf(n2, EPS32));
            float g_v[3];
            {   // y = v * inv:  dL/dv = inv * (g - y (y.g)) above the clamp, inv * g below it
                const float y0 = v[0] * inv, y1 = v[1] * inv, y2 = v[2] * inv;
                const float dotp = n2 > EPS32 ? y0 * g_pn[0] + y1 * g_pn[1] + y2 * g_pn[2] : 0.f;
                g_v[0] = inv * (g_pn[0] - y0 * dotp); g_v[1] = inv * (g_pn[1] - y1 * dotp); g_v[2] = inv * (g_pn[2] - y2 * dotp);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) GR(k) = k < 3 ? g_v[k] : 0.f;
            __syncthreads();
            dw_tile(t_n1, 0, 16 * wave);                       // normal_net.1 (3 x 64): k block = wave
            // dL/dhn, masked by the ReLU
#pragma unroll
            for (int o = 0; o < 64; ++o) {
                const float gh = fmaf(wlds[W_N1 + o], g_v[0], fmaf(wlds[W_N1 + 64 + o], g_v[1], wlds[W_N1 + 128 + o] * g_v[2]));
                hn[o] = hn[o] > 0.f ? gh : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) XR(k) = geo[k];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 32; ++k) GR(k) = hn[32 * h + k];
                __syncthreads();
                if (wave < 2) dw_tile(t_n0[h], 16 * wave, 0);   // normal_net.0 (64 x 16): row block 2h + wave
                float g_head[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) g_head[k] = 0.f;
                back_rows<16>(32, W_N0 + 32 * h * 16, g_head);
#pragma unroll
                for (int k = 0; k < 16; ++k) g_geo[k] += keep_n * g_head[k];      // (detach_density_for_normal_loss: 0)
                __syncthreads();
            }
        }
        // ---- mirror head: geo -> hm (LeakyReLU, bias) -> sigmoid (bias)
        if (P.g_m) {
            unsigned bits_m = 0;
#pragma unroll 1
            for (int o = 0; o < 32; ++o) {
                const float v = dot_row_lds(geo, W_M0 + o * 16) + wlds[B_M0 + o];
                bits_m |= (unsigned)(v > 0.f) << o;
                XR(o) = v > 0.f ? v : 0.01f * v;
            }
            float hm[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) hm[k] = XR(k);
            const float m = 1.f / (1.f + expf(-(dot_row_lds(hm, W_M1) + wlds[B_M1])));
            const float g_z = g_m * m * (1.f - m);
            db_m1 += g_z;
#pragma unroll
            for (int k = 0; k < 16; ++k) GR(k) = k == 0 ? g_z : 0.f;
            __syncthreads();
            if (wave < 2) dw_tile(t_m1, 0, 16 * wave);          // is_mirror_net.2 (1 x 32): k block = wave
#pragma unroll
            for (int o = 0; o < 32; ++o) hm[o] = wlds[W_M1 + o] * g_z * (((bits_m >> o) & 1u) ? 1.f : 0.01f);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) XR(k) = k < 15 ? geo[k] : 1.f;      // feature 15 = 1: its column is the bias gradient
#pragma unroll
            for (int k = 0; k < 32; ++k) GR(k) = hm[k];
            __syncthreads();
            if (wave < 2) dw_tile(t_m0, 16 * wave, 0);          // is_mirror_net.0 (32 x 16): row block = wave
            float g_head[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) g_head[k] = 0.f;
            back_rows<16>(32, W_M0, g_head);                    // (column 15 of the padded weights is zero)
#pragma unroll
            for (int k = 0; k < 16; ++k) g_geo[k] += keep_mh * g_head[k];         // (detach_density_*_for_mask_loss: 0)
            __syncthreads();
        }
        // ---- colour head: [SH4(d), geo] -> c1 (ReLU) -> c2 (ReLU) -> sigmoid
        float g_dir[3] = {0.f, 0.f, 0.f};
        if (P.g_rgb) {
            float in[32];
            sh4(d, in);
#pragma unroll
            for (int k = 0; k < 15; ++k) in[16 + k] = geo[k];
            in[31] = 0.f;
#pragma unroll 1
            for (int o = 0; o < 64; ++o) XR(o) = fmaxf(dot_row_lds(in, W_C0 + o * 32), 0.f);
            float c1[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) c1[k] = XR(k);
            unsigned long long bits_c1 = 0, bits_c2 = 0;
#pragma unroll
            for (int k = 0; k < 64; ++k) bits_c1 |= (unsigned long long)(c1[k] > 0.f) << k;
#pragma unroll 1
            for (int o = 0; o < 64; ++o) {
                const float pre = dot_row_lds(c1, W_C1 + o * 64);
                bits_c2 |= (unsigned long long)(pre > 0.f) << o;
                XR(o) = fmaxf(pre, 0.f);
            }
            float g_z[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 64; ++j) a = fmaf(wlds[W_C2 + k * 64 + j], XR(j), a);
                const float c = 1.f / (1.f + expf(-a));
                g_z[k] = g_rgb[k] * c * (1.f - c);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) GR(k) = k < 3 ? g_z[k] : 0.f;
            __syncthreads();
            dw_tile(t_c2, 0, 16 * wave);                        // color_net.2 (3 x 64): k block = wave
            float g_c2[64];
#pragma unroll
            for (int o = 0; o < 64; ++o) {
                const float gh = fmaf(wlds[W_C2 + o], g_z[0], fmaf(wlds[W_C2 + 64 + o], g_z[1], wlds[W_C2 + 128 + o] * g_z[2]));
                g_c2[o] = ((bits_c2 >> o) & 1ull) ? gh : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 64; ++k) XR(k) = c1[k];
            float g_c1[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) g_c1[k] = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 32; ++k) GR(k) = g_c2[32 * h + k];
                __syncthreads();
                dw_tile(t_c1[h][0], 0, 16 * wave);              // color_net.1 (64 x 64): row blocks 2h, 2h+1; k block = wave
                dw_tile(t_c1[h][1], 16, 16 * wave);
                back_rows<64>(32, W_C1 + 32 * h * 64, g_c1);
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) g_c1[k] = ((bits_c1 >> k) & 1ull) ? g_c1[k] : 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) XR(k) = in[k];
            float g_in[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) g_in[k] = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 32; ++k) GR(k) = g_c1[32 * h + k];
                __syncthreads();
                dw_tile(t_c0[h], 16 * (wave >> 1), 16 * (wave & 1));   // color_net.0 (64 x 32): row block 2h + wave/2, k block wave%2
                back_rows<32>(32, W_C0 + 32 * h * 32, g_in);
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) g_geo[k] += g_in[16 + k];
            if (P.d_dir) sh4_backward(d, g_in, g_dir);
        }
        if (P.d_dir && live) { P.d_dir[i * 3] = g_dir[0]; P.d_dir[i * 3 + 1] = g_dir[1]; P.d_dir[i * 3 + 2] = g_dir[2]; }

        // ---- trunk: (g_sigma, g_geo) -> sigma_net.1 -> h1 (ReLU) -> sigma_net.0 -> encoding -> table
        float enc[32];
        {
#pragma unroll 1
            for (int lv = 0; lv < NL; ++lv) {
                float a0, a1, g0[3], g1[3];
                encode_level<false>(A, lv, u, oob, a0, a1, g0, g1);
                XR(2 * lv) = a0;
                XR(2 * lv + 1) = a1;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) enc[k] = XR(k);
#pragma unroll 1
            for (int o = 0; o < 64; ++o) XR(o) = fmaxf(dot_row_lds(enc, W_S0 + o * 32), 0.f);
        }
        GR(0) = g_sigma;
#pragma unroll
        for (int k = 0; k < 15; ++k) GR(1 + k) = g_geo[k];
        __syncthreads();
        dw_tile(t_s1, 0, 16 * wave);                            // sigma_net.1 (16 x 64): k block = wave
        float g_h1[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) g_h1[k] = 0.f;
        back_rows<64>(16, W_S1, g_h1);
#pragma unroll
        for (int k = 0; k < 64; ++k) g_h1[k] = ((bits_h1 >> k) & 1ull) ? g_h1[k] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) XR(k) = enc[k];
        float g_enc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) g_enc[k] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 32; ++k) GR(k) = g_h1[32 * h + k];
            __syncthreads();
            dw_tile(t_s0[h], 16 * (wave >> 1), 16 * (wave & 1));       // sigma_net.0 (64 x 32)
            back_rows<32>(32, W_S0 + 32 * h * 32, g_enc);
            __syncthreads();
        }
        // ---- encoding backward: table gradient (gridencoder.cu:275-380) and dL/dx through the interpolation weights
#pragma unroll
        for (int k = 0; k < 32; ++k) GR(k) = g_enc[k];           // (own column; indexed by level below)
        float gx[3] = {0.f, 0.f, 0.f};
        const bool active = live && !oob;
        if (P.genc) {      // MNRF_TCNN_GRAD_FIXED: hand dL/d encoding and its magnitude per level to tcnn_scatter_fx_kernel
#pragma unroll 1
            for (int lv = 0; lv < NL; ++lv) {
                const float e0 = active ? GR(2 * lv) : 0.f, e1 = active ? GR(2 * lv + 1) : 0.f;
                if (live) P.genc[(long long)lv * A.Bs + i] = float2{e0, e1};
                float mx = fmaxf(fabsf(e0), fabsf(e1));
                if (!(mx < 1.0e30f)) mx = 1.0e30f;                      // (inf / nan upstream: a finite scale; the sums are garbage either way)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx += __shfl_xor(mx, o);
                if (lane == 0 && mx > 0.f) atomicAdd(&wg_ssum[lv], (double)mx);      // (LDS; one global atomic per level at the end)
            }
        }
#pragma unroll 1
        for (int lv = 0; lv < NL; ++lv) {
            if (P.genc && !P.d_xyz) break;      // (nothing left per level here; with d_xyz: the gathers below, no scatter)
            const float e0 = active ? GR(2 * lv) : 0.f, e1 = active ? GR(2 * lv + 1) : 0.f;
            const float scale = A.scale[lv];
            const unsigned res = A.res[lv];
            const unsigned hsize = A.off[lv + 1] - A.off[lv];
            const bool coarse = P.cp_n[lv] != 0;                          // (wave-uniform)
            float* dtab = coarse ? P.copies + P.cp_off[lv] + 2ll * hsize * (blockIdx.x % (unsigned)P.cp_n[lv])
                                 : P.d_table + 2ll * A.off[lv];
            unsigned pg[3];
            float fr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float pos = u[a] * scale + 0.5f;
                const float fl = floorf(pos);
                pg[a] = active ? (unsigned)fl : 0u;
                fr[a] = pos - fl;
            }
            float v0[8], v1[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wx = (c & 1) ? fr[0] : 1.f - fr[0];
                const float wy = (c & 2) ? fr[1] : 1.f - fr[1];
                const float wz = (c & 4) ? fr[2] : 1.f - fr[2];
                const float w = wx * wy * wz;
                v0[c] = w * e0;
                v1[c] = w * e1;
                if (P.d_xyz && active) {
                    const float2 v = tab_fetch(A.table, A.table_f16, A.off[lv] + grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, A.mode[lv]));
                    const float ev = e0 * v.x + e1 * v.y;
                    gx[0] += ((c & 1) ? scale : -scale) * wy * wz * ev;
                    gx[1] += ((c & 2) ? scale : -scale) * wx * wz * ev;
                    gx[2] += ((c & 4) ? scale : -scale) * wx * wy * ev;
                }
            }
            if (!P.genc) scatter_corners(P, lv, pg, v0, v1, active, lane, dtab, hsize, res);
        }
        if (P.d_xyz && live) {
            const float s = 1.f / (2.f * A.bound);
            P.d_xyz[i * 3] = gx[0] * s; P.d_xyz[i * 3 + 1] = gx[1] * s; P.d_xyz[i * 3 + 2] = gx[2] * s;
        }
    }
    if (P.genc) {
        __syncthreads();
        if (threadIdx.x < NL && wg_ssum[threadIdx.x] > 0.0) atomicAdd(P.ssum + threadIdx.x, wg_ssum[threadIdx.x]);
    }
    // ---- this workgroup's share of the weight gradients
    float* D = P.d_weights;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        flush_tile(D, t_s0[h], W_S0, 32, 32 * h + 16 * (wave >> 1), 16 * (wave & 1), 64, 32, -1, 0);
        flush_tile(D, t_c0[h], W_C0, 32, 32 * h + 16 * (wave >> 1), 16 * (wave & 1), 64, 32, -1, 0);
        flush_tile(D, t_c1[h][0], W_C1, 64, 32 * h, 16 * wave, 64, 64, -1, 0);
        flush_tile(D, t_c1[h][1], W_C1, 64, 32 * h + 16, 16 * wave, 64, 64, -1, 0);
        if (wave < 2) flush_tile(D, t_n0[h], W_N0, 16, 32 * h + 16 * wave, 0, 64, 16, -1, 0);
    }
    flush_tile(D, t_s1, W_S1, 64, 0, 16 * wave, 16, 64, -1, 0);
    flush_tile(D, t_c2, W_C2, 64, 0, 16 * wave, 3, 64, -1, 0);
    flush_tile(D, t_n1, W_N1, 64, 0, 16 * wave, 3, 64, -1, 0);
    if (wave < 2) {
        flush_tile(D, t_m0, W_M0, 16, 16 * wave, 0, 32, 15, 15, B_M0);
        flush_tile(D, t_m1, W_M1, 32, 0, 16 * wave, 1, 32, -1, 0);
    }
    // bias of is_mirror_net.2: sum of g_z over the workgroup's samples
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) db_m1 += __shfl_xor(db_m1, o);
    if (lane == 0) fadd(D + B_M1, db_m1);
}

// ------------------------------------------------------------------------------------------------------------
// Second-order term: the gradient that reaches the table, sigma_net and x through  normal = l2n(-J),
// J = d sigma / dx = s * sum_f genc_f * dfeat_f/du   (s = 1 / (2 bound), u = the [0,1] coordinates),
// genc = W_s0^T (m . w1): m = ReLU mask of sigma_net.0, w1 = row 0 of sigma_net.1 -- what autograd.grad(sigma, x,
// create_graph=True) differentiates in models/mirror_nerf_tcnn.py:172-218 (utils/func.py:10-25).  With J^ = dL/dJ:
//   * table:        dL/dT[idx_c][ch] += s * genc_{lv,ch} * sum_a J^_a dw_c/du_a          (a scatter like the first-order one)
//   * sigma_net.0:  dL/dW_s0[k][f]   += (m_k w1_k) * Df_f,   Df_f = s * sum_a J^_a dfeat_f/du_a   (feature tangent along J^)
//   * sigma_net.1:  dL/dw1_k         += m_k * (W_s0 Df)_k
//   * x:            dL/dx_b          += s^2 * sum_{lv,c,ch} genc * T[idx_c][ch] * sum_{a != b} J^_a d2w_c/du_a du_b
//     (trilinear interpolation: the pure second derivatives vanish inside a cell, the mixed ones do not)
// One thread per sample, same LDS layout and tile helpers as tcnn_bwd_kernel; three gather passes (features for the mask,
// derivatives for J, derivatives again for the scatter); runs before tcnn_fold_kernel and ADDS to d_table / d_weights / d_xyz.
__global__ __launch_bounds__(BT) void tcnn_bwd2_kernel(TcnnBwdArgs P) {
    live_rows(P.f);
    const TcnnArgs& A = P.f;
    for (int k = threadIdx.x; k < W_TOTAL; k += BT) wlds[k] = A.weights[k];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 t_s0[2] = {z4, z4}, t_s1 = z4;
    const float sc = 1.f / (2.f * A.bound);
    const long long ntiles = (A.B + BT - 1) / BT;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        long long i = tile * BT + threadIdx.x;
        const bool live = i < A.B;
        if (!live) i = A.B - 1;
        float x[3];
        if (A.xyz) {
            const float* p = A.xyz + i * A.xyz_stride;
            x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
        } else {
            const float* r = A.rays + (i / A.spr) * 8;
            const float z = A.z_vals[i];
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
        }
        float u[3];
        bool oob = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            u[a] = (x[a] + A.bound) / (2.f * A.bound);
            oob |= u[a] < 0.f || u[a] > 1.f;
        }
        const bool active = live && !oob;
        // ---- pass 1: features -> ReLU mask of sigma_net.0 -> q = m . w1 -> genc = W_s0^T q
        unsigned long long bits_h1 = 0;
        {
#pragma unroll 1
            for (int lv = 0; lv < NL; ++lv) {
                float a0, a1, g0[3], g1[3];
                encode_level<false>(A, lv, u, oob, a0, a1, g0, g1);
                XR(2 * lv) = a0;
                XR(2 * lv + 1) = a1;
            }
            float in[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) in[k] = XR(k);
#pragma unroll 1
            for (int o = 0; o < 64; ++o) bits_h1 |= (unsigned long long)(dot_row_lds(in, W_S0 + o * 32) > 0.f) << o;
        }
        float genc[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) genc[k] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 32; ++k) GR(k) = ((bits_h1 >> (32 * h + k)) & 1ull) ? wlds[W_S1 + 32 * h + k] : 0.f;
            back_rows<32>(32, W_S0 + 32 * h * 32, genc);      // (own column of GR: no barrier needed)
        }
        // ---- pass 2: J, then J^ from dL/dnormal through n = -J / max(|J|, sqrt(eps))
        float J[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 32; ++k) XR(k) = genc[k];          // (indexed by level below)
#pragma unroll 1
        for (int lv = 0; lv < NL; ++lv) {
            float a0, a1, g0[3], g1[3];
            encode_level<true>(A, lv, u, oob, a0, a1, g0, g1);
            const float e0 = XR(2 * lv), e1 = XR(2 * lv + 1);
#pragma unroll
            for (int a = 0; a < 3; ++a) J[a] += e0 * g0[a] + e1 * g1[a];
        }
        float Jh[3];
        {
            const float n0 = -J[0] * sc, n1 = -J[1] * sc, n2 = -J[2] * sc;
            const float sq = n0 * n0 + n1 * n1 + n2 * n2;
            const float inv = 1.f / sqrtf(fmaxf(sq, EPS32));
            const float g0 = live ? P.g_normal[i * 3] : 0.f, g1 = live ? P.g_normal[i * 3 + 1] : 0.f, g2 = live ? P.g_normal[i * 3 + 2] : 0.f;
            // normal = v * inv with v = -s J:  dL/dv = (g - n (n.g)) * inv  (clamped branch: g * inv);  dL/dJ = -s dL/dv
            float d0 = g0 * inv, d1 = g1 * inv, d2 = g2 * inv;
            if (sq > EPS32) {
                const float m0 = n0 * inv, m1 = n1 * inv, m2 = n2 * inv, dt = m0 * g0 + m1 * g1 + m2 * g2;
                d0 = (g0 - m0 * dt) * inv; d1 = (g1 - m1 * dt) * inv; d2 = (g2 - m2 * dt) * inv;
            }
            Jh[0] = -sc * d0; Jh[1] = -sc * d1; Jh[2] = -sc * d2;
        }
        // ---- pass 3: feature tangent Df, table scatter, mixed second derivatives for dL/dx
        float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int lv = 0; lv < NL; ++lv) {
            const float e0 = active ? XR(2 * lv) : 0.f, e1 = active ? XR(2 * lv + 1) : 0.f;      // genc of this level
            const float scale = A.scale[lv];
            const unsigned res = A.res[lv];
            const unsigned hsize = A.off[lv + 1] - A.off[lv];
            const bool coarse = P.cp_n[lv] != 0;
            float* dtab = coarse ? P.copies + P.cp_off[lv] + 2ll * hsize * (blockIdx.x % (unsigned)P.cp_n[lv])
                                 : P.d_table + 2ll * A.off[lv];
            unsigned pg[3];
            float fr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float pos = u[a] * scale + 0.5f;
                const float fl = floorf(pos);
                pg[a] = active ? (unsigned)fl : 0u;
                fr[a] = pos - fl;
            }
            float v0[8], v1[8], df0 = 0.f, df1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wx = (c & 1) ? fr[0] : 1.f - fr[0];
                const float wy = (c & 2) ? fr[1] : 1.f - fr[1];
                const float wz = (c & 4) ? fr[2] : 1.f - fr[2];
                const float sx = (c & 1) ? scale : -scale, sy = (c & 2) ? scale : -scale, sz = (c & 4) ? scale : -scale;
                // sum_a J^_a dw_c/du_a
                const float dw = Jh[0] * sx * wy * wz + Jh[1] * sy * wx * wz + Jh[2] * sz * wx * wy;
                v0[c] = dw * e0;
                v1[c] = dw * e1;
                float2 v = float2{0.f, 0.f};
                if (active) v = tab_fetch(A.table, A.table_f16, A.off[lv] + grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, A.mode[lv]));
                df0 += dw * v.x; df1 += dw * v.y;
                if (P.d_xyz) {
                    const float ev = e0 * v.x + e1 * v.y;
                    gx[0] += ev * (Jh[1] * sx * sy * wz + Jh[2] * sx * sz * wy);
                    gx[1] += ev * (Jh[0] * sx * sy * wz + Jh[2] * sy * sz * wx);
                    gx[2] += ev * (Jh[0] * sx * sz * wy + Jh[1] * sy * sz * wx);
                }
            }
            GR(2 * lv) = active ? df0 : 0.f;                   // (own column; collected into registers after the loop)
            GR(2 * lv + 1) = active ? df1 : 0.f;
            scatter_corners(P, lv, pg, v0, v1, active, lane, dtab, hsize, res);
        }
        float Df[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) Df[k] = GR(k);
        if (P.d_xyz && live) {
            P.d_xyz[i * 3] += gx[0] * sc; P.d_xyz[i * 3 + 1] += gx[1] * sc; P.d_xyz[i * 3 + 2] += gx[2] * sc;
        }
        // ---- weight gradients: dW_s0 += q (x) Df over the tile's samples; dw1 += m . (W_s0 Df)
#pragma unroll 1
        for (int o = 0; o < 64; ++o) XR(o) = ((bits_h1 >> o) & 1ull) ? dot_row_lds(Df, W_S0 + o * 32) : 0.f;
        __syncthreads();
        GR(0) = 1.f;
#pragma unroll
        for (int k = 1; k < 16; ++k) GR(k) = 0.f;
        __syncthreads();
        dw_tile(t_s1, 0, 16 * wave);                            // row 0 of sigma_net.1 (the other 15 rows of the tile get zeros)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) XR(k) = Df[k];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 32; ++k) GR(k) = ((bits_h1 >> (32 * h + k)) & 1ull) ? wlds[W_S1 + 32 * h + k] : 0.f;
            __syncthreads();
            dw_tile(t_s0[h], 16 * (wave >> 1), 16 * (wave & 1));       // sigma_net.0 (64 x 32), as in tcnn_bwd_kernel
            __syncthreads();
        }
    }
    float* D = P.d_weights;
#pragma unroll
    for (int h = 0; h < 2; ++h) flush_tile(D, t_s0[h], W_S0, 32, 32 * h + 16 * (wave >> 1), 16 * (wave & 1), 64, 32, -1, 0);
    flush_tile(D, t_s1, W_S1, 64, 0, 16 * wave, 1, 64, -1, 0);
}
#undef XR
#undef GR

// d_table[level entries] += sum of the level's private copies
// power of two k with k * S <= 2^30, S = the level's bound on any entry's sum (TcnnBwdArgs::ssum)
__device__ __forceinline__ float fx_scale(double S) {
    if (!(S > 0.0)) return 0.f;                                           // nothing to add on this level
    int e;
    (void)frexp(S, &e);                                                   // S = m 2^e, m in [0.5, 1): S < 2^e
    const int p = 30 - e;
    if (p < -120 || p > 120) return 0.f;                                  // (gradients outside any sane range: leave the level out)
    return __uint_as_float((unsigned)(127 + p) << 23);
}

// MNRF_TCNN_GRAD_FIXED: the scatter as its own launch over the planes tcnn_bwd_kernel wrote.  One thread per sample, all levels;
// levels with private copies add fp32 into them as before, the others one packed 64-bit integer atomic per corner.
__global__ __launch_bounds__(256) void tcnn_scatter_fx_kernel(TcnnBwdArgs P) {
    live_rows(P.f);
    const TcnnArgs& A = P.f;
    const int lane = threadIdx.x & 63;
    const long long ntiles = (A.B + 255) / 256;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long i = tile * 256 + threadIdx.x;
        const bool live = i < A.B;
        if (!live) i = A.B - 1;
        float x[3];
        if (A.xyz) {
            const float* p = A.xyz + i * A.xyz_stride;
            x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
        } else {
            const float* r = A.rays + (i / A.spr) * 8;
            const float z = A.z_vals[i];
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a] = r[a] + r[3 + a] * z;
        }
        float u[3];
        bool oob = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            u[a] = (x[a] + A.bound) / (2.f * A.bound);
            oob |= u[a] < 0.f || u[a] > 1.f;
        }
        const bool active = live && !oob;
#pragma unroll 1
        for (int lv = 0; lv < NL; ++lv) {
            const float2 e = active ? P.genc[(long long)lv * A.Bs + i] : float2{0.f, 0.f};
            const float scale = A.scale[lv];
            const unsigned res = A.res[lv];
            const unsigned hsize = A.off[lv + 1] - A.off[lv];
            unsigned pg[3];
            float fr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float pos = u[a] * scale + 0.5f;
                const float fl = floorf(pos);
                pg[a] = active ? (unsigned)fl : 0u;
                fr[a] = pos - fl;
            }
            float v0[8], v1[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = ((c & 1) ? fr[0] : 1.f - fr[0]) * ((c & 2) ? fr[1] : 1.f - fr[1]) * ((c & 4) ? fr[2] : 1.f - fr[2]);
                v0[c] = w * e.x;
                v1[c] = w * e.y;
            }
            if (P.cp_n[lv]) {                                             // (wave-uniform) coarse level: fp32 into this workgroup's private copy
                float* dtab = P.copies + P.cp_off[lv] + 2ll * hsize * (blockIdx.x % (unsigned)P.cp_n[lv]);
                scatter_corners(P, lv, pg, v0, v1, active, lane, dtab, hsize, res);
                continue;
            }
            const bool head = aggregate_runs(P, lv, pg, v0, v1, active, lane);
            const float k = fx_scale(P.ssum[lv]);
            if (!(active && head) || k == 0.f || P.exp_noscatter) continue;
            unsigned long long* fx = P.fx + A.off[lv];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned idx = grid_index(pg[0] + (c & 1), pg[1] + ((c >> 1) & 1), pg[2] + ((c >> 2) & 1), hsize, res, A.mode[lv]);
                const long long q0 = (long long)__float2int_rn(v0[c] * k), q1 = (long long)__float2int_rn(v1[c] * k);      // (|v k| <= 2^30)
                if (q0 | q1) atomicAdd(fx + idx, (unsigned long long)((q1 << 32) + q0));
            }
        }
    }
}

// d_table += decode(fx) / scale of the entry's level, for the levels that accumulated in fixed point (no private copies)
__global__ void tcnn_fold_fx_kernel(TcnnBwdArgs P) {
    const TcnnArgs& A = P.f;
    const int lv = blockIdx.y;
    if (P.cp_n[lv]) return;
    const float k = fx_scale(P.ssum[lv]);
    if (k == 0.f) return;
    const float inv = 1.f / k;
    const unsigned n = A.off[lv + 1] - A.off[lv];
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const long long t = (long long)P.fx[A.off[lv] + e];
        if (t == 0) continue;
        const long long lo = (long long)(int)(unsigned)(t & 0xffffffffll);      // the low half, sign-extended ...
        const long long hi = (t - lo) >> 32;                                    // ... and its carries taken back out of the high half
        float* d = P.d_table + 2ll * (A.off[lv] + e);
        d[0] += (float)lo * inv;
        d[1] += (float)hi * inv;
    }
}

__global__ void tcnn_fold_kernel(TcnnBwdArgs P) {
    const TcnnArgs& A = P.f;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long base = 0;
    for (int lv = 0; lv < NL; ++lv) {
        if (!P.cp_n[lv]) continue;
        const long long n = 2ll * (A.off[lv + 1] - A.off[lv]);
        if (q < base + n) {
            const long long e = q - base;
            float a = 0.f;
            for (int c = 0; c < P.cp_n[lv]; ++c) a += P.copies[P.cp_off[lv] + c * n + e];
            P.d_table[2ll * A.off[lv] + e] += a;
            return;
        }
        base += n;
    }
}

// copies per level: 32 for levels of <= 32 Ki entries, 8 (one per XCD) up to 256 Ki entries, none above (the hashed levels:
// random addresses, little contention).  Returns the workspace size in floats.
long long plan_copies(const int64_t* off17, int* cp_n, long long* cp_off, long long* folded_floats) {
    long long total = 0, folded = 0;
    for (int lv = 0; lv < NL; ++lv) {
        const long long n = off17[lv + 1] - off17[lv];
        cp_n[lv] = n <= 32768 ? 32 : (n <= 262144 ? 8 : 0);
        cp_off[lv] = total;
        total += 2 * n * cp_n[lv];
        if (cp_n[lv]) folded += 2 * n;
    }
    if (folded_floats) *folded_floats = folded;
    return total;
}

}  // namespace

static void level_modes(TcnnArgs& A) {
    for (int l = 0; l < NL; ++l) {
        const unsigned long long hsize = A.off[l + 1] - A.off[l], r1 = (unsigned long long)A.res[l] + 1;
        // the running stride of get_grid_index stays <= hsize through all three axes iff (res+1)^3 <= hsize
        const bool dense = r1 <= hsize && r1 * r1 <= hsize && r1 * r1 * r1 <= hsize;
        A.mode[l] = dense ? 0u : ((hsize & (hsize - 1)) == 0 ? 1u : 2u);
    }
}

extern "C" int mnrf_tcnn_weight_floats(void) { return W_TOTAL; }

// ---- the weight blob from the 11 parameter tensors in ONE launch (round 4: the host side built it with ten pads, a cat and a
// final pad -- 23 launches per model after every optimizer step, 8 % of config 5's training step)
namespace {
struct BlobSeg { const float* src; int rows, src_cols, dst_cols, dst_off; };
struct BlobArgs { BlobSeg seg[11]; float* dst; };
__global__ void tcnn_pack_kernel(BlobArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_TOTAL) return;
    float v = 0.f;      // padded columns and the tail stay zero
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const BlobSeg s = A.seg[k];
        const int j = i - s.dst_off;
        if (j >= 0 && j < s.rows * s.dst_cols) {
            const int r = j / s.dst_cols, c = j - r * s.dst_cols;
            if (c < s.src_cols) v = s.src[r * s.src_cols + c];
        }
    }
    A.dst[i] = v;
}
}  // namespace

extern "C" int mnrf_tcnn_pack_weights(const float* const* params, float* weights, void* stream) {
    if (!params || !weights) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_pack_weights: null pointer");
    // rows, source columns, blob columns, blob offset of [sigma_net.0.weight, sigma_net.1.weight, color_net.0/1/2.weight,
    // normal_net.0/1.weight, is_mirror_net.0.weight, .0.bias, .2.weight, .2.bias] (models/mirror_nerf_tcnn.py:51-149)
    static const int shape[11][4] = {{64, 32, 32, W_S0}, {16, 64, 64, W_S1}, {64, 31, 32, W_C0}, {64, 64, 64, W_C1}, {3, 64, 64, W_C2},
                                     {64, 15, 16, W_N0}, {3, 64, 64, W_N1}, {32, 15, 16, W_M0}, {1, 32, 32, B_M0}, {1, 32, 32, W_M1},
                                     {1, 1, 1, B_M1}};
    BlobArgs A;
    for (int k = 0; k < 11; ++k) {
        if (!params[k]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_pack_weights: null parameter pointer");
        A.seg[k] = BlobSeg{params[k], shape[k][0], shape[k][1], shape[k][2], shape[k][3]};
    }
    A.dst = weights;
    hipLaunchKernelGGL(tcnn_pack_kernel, dim3((W_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream, A);
    return mnrf_check_launch("mnrf_tcnn_pack_weights");
}

static int tcnn_forward_impl(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                             int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                             const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                             const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                             float* is_mirror, float* normal, float* geo_feat, float* enc_workspace, const int32_t* n_live, void* stream) {
    if (n_live && xyz) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward_n: a live row count needs ray mode (rays + z_vals)");
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
    A.n_live = n_live; A.Bs = B;
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
    level_modes(A);
    A.sigma = sigma; A.rgb = rgb; A.pred_normal = pred_normal; A.is_mirror = is_mirror; A.normal = normal; A.geo_feat = geo_feat;
    A.enc = enc_workspace;
    A.table_f16 = (flags & MNRF_TCNN_TABLE_F16) ? 1u : 0u;
    hipStream_t s = (hipStream_t)stream;
    // MLPs on the matrix pipe (default); MNRF_TCNN_VALU=1 selects the one-thread-per-sample VALU kernel (A/B measurements)
    static const bool env_valu = [] { const char* e = getenv("MNRF_TCNN_VALU"); return e && e[0] == '1'; }();
    bool any_modulo = false;
    for (int l = 0; l < NL; ++l) any_modulo |= A.mode[l] == 2u;
    // The matrix-pipe kernel takes the full evaluations without the density-gradient normal (see its header for the
    // measurements); everything else, and tables with a hashed level whose size is not a power of two (integer modulo per
    // corner), stays on the VALU kernel.
    // MNRF_TCNN_F16: single-pass f16 MLPs (one MFMA per product); then the sigma-only launches run on the matrix pipe too.  With three
    // products the ONE-launch form measured 1.11 ms against the VALU kernel's 0.98 ms per 2.1 M samples and stays on the VALU
    // kernel; behind the level-major encoding launch (enc_workspace) the sigma-only MLPs take the matrix pipe in both arithmetics
    const bool f16 = (flags & MNRF_TCNN_F16) != 0;
    const bool encode_only = (flags & 0x80000000u) != 0;       // internal: mnrf_tcnn_encode
    const bool valu = !encode_only && (env_valu || (flags & MNRF_TCNN_VALU) || any_modulo || (sigma_only && !f16 && !enc_workspace) || grad);
    if (encode_only && any_modulo) return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_tcnn_encode: hashed level of a non-power-of-two size");
    if (!valu) {
        const long long n_tiles = (B + mf::TILE - 1) / mf::TILE;
        if (n_tiles > 0x7fffffff) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: too many samples for one launch");
        static const bool attr = [] {
            (void)hipFuncSetAttribute((const void*)mf::tcnn_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, mf::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)mf::tcnn_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, mf::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)mf::tcnn_mfma_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mf::LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)mf::tcnn_mfma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mf::LDS_BYTES);
            return true;
        }();
        (void)attr;
        // persistent: two workgroups per CU (round 6, alternating on one box: 256 workgroups 2.09 ms per chunk / 1.72 with f16 MLPs,
        // 512: 2.04 / 1.62, 768 and 1024: the same)
        const dim3 g2((unsigned)(n_tiles < 512 ? n_tiles : 512)), b2(64 * mf::WAVES);
        if (A.enc) {      // level-major encoding into the caller's planes, then the MLPs from the planes
            const long long nb = (B + 255) / 256;
            if (nb > 0x7fffffff) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_forward: too many samples for one launch");
            hipLaunchKernelGGL(mf::tcnn_encode_kernel, dim3((unsigned)nb, NL), dim3(256), 0, s, A);
            if (encode_only) return mnrf_check_launch("mnrf_tcnn_encode");
            if (f16) hipLaunchKernelGGL((mf::tcnn_mfma_kernel<1, true>), g2, b2, mf::LDS_BYTES, s, A, (int)n_tiles);
            else hipLaunchKernelGGL((mf::tcnn_mfma_kernel<0, true>), g2, b2, mf::LDS_BYTES, s, A, (int)n_tiles);
        } else if (f16) hipLaunchKernelGGL(mf::tcnn_mfma_kernel<1>, g2, b2, mf::LDS_BYTES, s, A, (int)n_tiles);
        else hipLaunchKernelGGL(mf::tcnn_mfma_kernel<0>, g2, b2, mf::LDS_BYTES, s, A, (int)n_tiles);
        return mnrf_check_launch("mnrf_tcnn_forward");
    }
    const dim3 grid((unsigned)((B + TPB - 1) / TPB)), block(TPB);
    const size_t lds = (VEC_OFF + 64 * TPB) * sizeof(float);
    if (sigma_only && !grad) hipLaunchKernelGGL((tcnn_kernel<true, false>), grid, block, lds, s, A);
    else if (sigma_only) hipLaunchKernelGGL((tcnn_kernel<true, true>), grid, block, lds, s, A);
    else if (!grad) hipLaunchKernelGGL((tcnn_kernel<false, false>), grid, block, lds, s, A);
    else hipLaunchKernelGGL((tcnn_kernel<false, true>), grid, block, lds, s, A);
    return mnrf_check_launch("mnrf_tcnn_forward");
}
extern "C" int mnrf_tcnn_forward(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                 int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                                 const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                                 const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                 float* is_mirror, float* normal, float* geo_feat, float* enc_workspace, void* stream) {
    return tcnn_forward_impl(table, offsets17_host, log2_per_level_scale, base_resolution, bound, weights, flags, B, xyz, xyz_stride, rays,
                             z_vals, spr, dirs, dir_stride, sigma, rgb, pred_normal, is_mirror, normal, geo_feat, enc_workspace, nullptr, stream);
}
extern "C" int mnrf_tcnn_forward_n(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                   int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                                   const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                                   const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                   float* is_mirror, float* normal, float* geo_feat, float* enc_workspace, const int32_t* n_live, void* stream) {
    return tcnn_forward_impl(table, offsets17_host, log2_per_level_scale, base_resolution, bound, weights, flags, B, xyz, xyz_stride, rays,
                             z_vals, spr, dirs, dir_stride, sigma, rgb, pred_normal, is_mirror, normal, geo_feat, enc_workspace, n_live, stream);
}

// d_table += g16 / scale for the entries [e0, e1) of the levels that accumulated in half2 (MNRF_TCNN_GRAD_F16)
// A sum that left the f16 range (|scaled gradient| > 65504: tinycudann pairs its f16 gradients with a loss scale and an inf check)
// is clamped to the largest finite value instead of carrying inf / nan into d_table and the optimizer state, and raises *overflow
// (the word behind the half2 table in the workspace; the Python shim reads it one step late, warns and falls back to fp32 atomics).
__global__ void tcnn_fold16_kernel(const __half2* __restrict__ g16, float2* __restrict__ d_table, long long e0, long long e1, float inv,
                                   unsigned* __restrict__ overflow) {
    const long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e1) return;
    float2 v = __half22float2(g16[e]);
    if (!(fabsf(v.x) <= 65504.f) || !(fabsf(v.y) <= 65504.f)) {
        v.x = v.x != v.x ? 0.f : fminf(fmaxf(v.x, -65504.f), 65504.f);
        v.y = v.y != v.y ? 0.f : fminf(fmaxf(v.y, -65504.f), 65504.f);
        atomicOr(overflow, 1u);
    }
    float2 d = d_table[e];
    d.x += v.x * inv;
    d.y += v.y * inv;
    d_table[e] = d;
}

extern "C" int64_t mnrf_tcnn_backward_workspace_floats(const int64_t* offsets17_host) {
    int n[NL];
    long long o[NL];
    return offsets17_host ? plan_copies(offsets17_host, n, o, nullptr) : 0;
}

// with MNRF_TCNN_GRAD_F16 in `flags` the workspace also holds the half2 gradient table (one 4-byte slot per entry) behind the copies
extern "C" int64_t mnrf_tcnn_backward_workspace_floats2(const int64_t* offsets17_host, unsigned flags) {
    if (!offsets17_host) return 0;
    const int64_t base = mnrf_tcnn_backward_workspace_floats(offsets17_host);
    return base + ((flags & MNRF_TCNN_GRAD_F16) ? offsets17_host[NL] + 4 : 0);      // (+ the overflow word, 16-byte padded)
}

// MNRF_TCNN_GRAD_FIXED: [private copies][2 floats per table entry: the packed 64-bit sums][32 * B floats: dL/d encoding planes]
// [32 words: the levels' sums (16 doubles)].  The launcher zeroes what must be zero; the caller zero-fills the copies (the first
// mnrf_tcnn_backward_workspace_floats() floats), as before.
extern "C" int64_t mnrf_tcnn_backward_workspace_floats3(const int64_t* offsets17_host, unsigned flags, int64_t B) {
    if (!offsets17_host) return 0;
    if (!(flags & MNRF_TCNN_GRAD_FIXED)) return mnrf_tcnn_backward_workspace_floats2(offsets17_host, flags);
    return mnrf_tcnn_backward_workspace_floats(offsets17_host) + 2 * offsets17_host[NL] + 32 * (B > 0 ? B : 0) + 32;
}

static int tcnn_backward_impl(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                              int base_resolution, float bound, const float* weights, int64_t B, const float* xyz,
                              int64_t xyz_stride, const float* rays, const float* z_vals, int spr, const float* dirs,
                              int64_t dir_stride, const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                              const float* g_is_mirror, const float* g_normal, float* workspace, float* d_table,
                              float* d_weights, float* d_xyz, float* d_dir, const float* keep_mirror, unsigned flags,
                              const int32_t* n_live, void* stream) {
    if (n_live && xyz) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward_n: a live row count needs ray mode (rays + z_vals)");
    if (!table || !offsets17_host || !weights || !d_table || !d_weights)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: null pointer");
    if (B < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: negative sample count");
    if (B == 0) return MNRF_OK;
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: need xyz or rays+z_vals");
    if (xyz && xyz_stride < 6) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: xyz_stride too small");
    if (spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: samples per ray must be >= 1");
    if (bound <= 0.f) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_backward: bound must be positive");
    TcnnBwdArgs P;
    TcnnArgs& A = P.f;
    A.n_live = n_live; A.Bs = B;
    A.table = table; A.weights = weights; A.B = B; A.xyz = xyz; A.xyz_stride = xyz_stride; A.rays = rays;
    A.z_vals = z_vals; A.spr = spr; A.dirs = dirs; A.dir_stride = dir_stride; A.bound = bound;
    for (int l = 0; l < NL; ++l) {
        A.scale[l] = (float)(exp2((double)l * log2_per_level_scale) * (double)base_resolution - 1.0);
        A.res[l] = (unsigned)ceilf(A.scale[l]) + 1u;
        A.off[l] = (unsigned)offsets17_host[l];
    }
    A.off[NL] = (unsigned)offsets17_host[NL];
    level_modes(A);
    A.enc = nullptr;
    A.table_f16 = (flags & MNRF_TCNN_TABLE_F16) ? 1u : 0u;
    A.sigma = A.rgb = A.pred_normal = A.is_mirror = A.normal = A.geo_feat = nullptr;
    P.g_sigma = g_sigma; P.g_rgb = g_rgb; P.g_pn = g_pred_normal; P.g_m = g_is_mirror; P.g_normal = g_normal;
    P.d_table = d_table; P.d_weights = d_weights; P.d_xyz = d_xyz; P.d_dir = d_dir;
    P.cut = flags & (MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD);
    P.keep_mirror = keep_mirror;
    P.copies = workspace;
    P.g16 = nullptr;
    P.g16_scale = 1.f;
    P.genc = nullptr; P.ssum = nullptr; P.fx = nullptr;
    const bool fixed = (flags & MNRF_TCNN_GRAD_FIXED) && workspace && !(flags & MNRF_TCNN_GRAD_F16);
    if (fixed) {
        float* w = workspace + mnrf_tcnn_backward_workspace_floats(offsets17_host);
        P.fx = (unsigned long long*)w;
        P.genc = (float2*)(w + 2 * offsets17_host[NL]);
        P.ssum = (double*)(w + 2 * offsets17_host[NL] + 32 * B);
        if (hipMemsetAsync(P.fx, 0, (size_t)offsets17_host[NL] * 8, (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(P.ssum, 0, NL * sizeof(double), (hipStream_t)stream) != hipSuccess)
            return mnrf_fail(MNRF_ERR_LAUNCH, "mnrf_tcnn_backward: hipMemsetAsync");
    }
    if ((flags & MNRF_TCNN_GRAD_F16) && workspace) {
        P.g16 = (__half2*)(workspace + mnrf_tcnn_backward_workspace_floats(offsets17_host));
        static const float scale = [] { const char* e = getenv("MNRF_TCNN_GRAD_SCALE"); return e && atof(e) > 0 ? (float)atof(e) : 1024.f; }();
        P.g16_scale = scale;       // tinycudann's loss scale is 128; 1024 keeps 1e-7-sized contributions above f16's subnormal step
    }
    P.exp_noscatter = getenv("MNRF_EXP_TCNN_NOSCATTER") != nullptr;
    P.agg_levels = 0;
    // measured (1 M samples, bound 6): 15.96 / 15.00 / 14.12 / 13.77 / 13.70 ms per step with runs summed up to resolution
    // 64 / 128 / 256 / 512 / 1000 -- the shuffles are cheap next to an atomic, so every level the key can hold takes part
    const int agg_res = getenv("MNRF_TCNN_AGG_RES") ? atoi(getenv("MNRF_TCNN_AGG_RES")) : 1022;
    while (P.agg_levels < NL && (int)A.res[P.agg_levels] <= agg_res && A.res[P.agg_levels] < 1023u) ++P.agg_levels;
    long long folded = 0;
    plan_copies(offsets17_host, P.cp_n, P.cp_off, &folded);
    if (!workspace || getenv("MNRF_TCNN_NO_COPIES")) {         // (null workspace: every level straight into d_table)
        for (int l = 0; l < NL; ++l) P.cp_n[l] = 0;
        folded = 0;
    }
    const long long ntiles = (B + BT - 1) / BT;
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(BT);      // persistent: one workgroup per CU of the MI355X
    const size_t lds = (size_t)BWD_LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL(tcnn_bwd_kernel, grid, block, lds, (hipStream_t)stream, P);
    if (fixed) {
        const long long nt = (B + 255) / 256;
        hipLaunchKernelGGL(tcnn_scatter_fx_kernel, dim3((unsigned)(nt < 2048 ? nt : 2048)), dim3(256), 0, (hipStream_t)stream, P);
        TcnnBwdArgs P2 = P;
        P2.genc = nullptr;        // the second-order kernel below scatters as before (fp32 atomics into copies / d_table)
        P = P2;
    }
    if (g_normal) hipLaunchKernelGGL(tcnn_bwd2_kernel, grid, block, lds, (hipStream_t)stream, P);     // adds the second-order term
    if (fixed) hipLaunchKernelGGL(tcnn_fold_fx_kernel, dim3(256, NL), dim3(256), 0, (hipStream_t)stream, P);
    if (folded) hipLaunchKernelGGL(tcnn_fold_kernel, dim3((unsigned)((folded + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    if (P.g16) {      // everything from the first level without copies on (copied levels in between only add the zeros they hold)
        long long e0 = -1;
        for (int l = 0; l < NL && e0 < 0; ++l)
            if (!P.cp_n[l]) e0 = A.off[l];
        const long long e1 = A.off[NL];
        if (e0 >= 0 && e1 > e0)
            hipLaunchKernelGGL(tcnn_fold16_kernel, dim3((unsigned)((e1 - e0 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P.g16,
                               (float2*)d_table, e0, e1, 1.f / P.g16_scale, (unsigned*)(P.g16 + offsets17_host[NL]));
    }
    return mnrf_check_launch("mnrf_tcnn_backward");
}
extern "C" int mnrf_tcnn_backward(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                  int base_resolution, float bound, const float* weights, int64_t B, const float* xyz,
                                  int64_t xyz_stride, const float* rays, const float* z_vals, int spr, const float* dirs,
                                  int64_t dir_stride, const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                  const float* g_is_mirror, const float* g_normal, float* workspace, float* d_table,
                                  float* d_weights, float* d_xyz, float* d_dir, const float* keep_mirror, unsigned flags,
                                  void* stream) {
    return tcnn_backward_impl(table, offsets17_host, log2_per_level_scale, base_resolution, bound, weights, B, xyz, xyz_stride, rays, z_vals,
                              spr, dirs, dir_stride, g_sigma, g_rgb, g_pred_normal, g_is_mirror, g_normal, workspace, d_table, d_weights,
                              d_xyz, d_dir, keep_mirror, flags, nullptr, stream);
}
extern "C" int mnrf_tcnn_backward_n(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                    int base_resolution, float bound, const float* weights, int64_t B, const float* xyz,
                                    int64_t xyz_stride, const float* rays, const float* z_vals, int spr, const float* dirs,
                                    int64_t dir_stride, const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                    const float* g_is_mirror, const float* g_normal, float* workspace, float* d_table,
                                    float* d_weights, float* d_xyz, float* d_dir, const float* keep_mirror, unsigned flags,
                                    const int32_t* n_live, void* stream) {
    return tcnn_backward_impl(table, offsets17_host, log2_per_level_scale, base_resolution, bound, weights, B, xyz, xyz_stride, rays, z_vals,
                              spr, dirs, dir_stride, g_sigma, g_rgb, g_pred_normal, g_is_mirror, g_normal, workspace, d_table, d_weights,
                              d_xyz, d_dir, keep_mirror, flags, n_live, stream);
}


// ---------------------------------------------------------------------------------------------------------- gather ceiling
// What bounds the hash-grid field is not HBM bandwidth (the 53 MB table lives in the 256 MB Infinity Cache) but the RATE of
// random small gathers.  This micro-benchmark measures that ceiling on the very table: every thread issues `iters`
// independent loads of `bytes` (8 = one float2 entry, 4 = what an fp16 table would fetch) at pseudo-random entries and
// folds them into a checksum.  bench.py prices the field kernel's 128 gathers per sample against it.
template <typename T>
__global__ __launch_bounds__(256) void gather_bench_kernel(const T* __restrict__ table, unsigned n_entries, int iters, float* __restrict__ out) {
    unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it += 8) {
        unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            state = state * 1664525u + 1013904223u;
            idx[u] = (unsigned)(((unsigned long long)(state >> 4) * n_entries) >> 28);      // uniform in [0, n_entries)
        }
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = table[idx[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (sizeof(T) == 8) acc += v[u].x + v[u].y;
            else acc += __builtin_bit_cast(float, v[u]) * 1e-30f;
        }
    }
    if (acc == 123.456f) out[0] = acc;      // keep the loads alive
}

// The multiresolution hash encoding alone (tcnn_encode_kernel): planes[level][sample] = the level's two features, for samples
// given as rows of `xyz` or as rays + z_vals.  This is the first of the two launches of mnrf_tcnn_forward(enc_workspace != null);
// exposed for measurement (bench.py times it against the L2 roofline) and for callers that want the encoding itself.
extern "C" int mnrf_tcnn_encode_flags(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                      int base_resolution, float bound, int64_t B, const float* xyz, int64_t xyz_stride,
                                      const float* rays, const float* z_vals, int spr, float* planes, unsigned flags, void* stream) {
    if (!planes) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_encode: null output");
    if (flags & ~MNRF_TCNN_TABLE_F16) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_encode: flags = 0 or MNRF_TCNN_TABLE_F16");
    return mnrf_tcnn_forward(table, offsets17_host, log2_per_level_scale, base_resolution, bound, table /* unused */,
                             0x80000000u | MNRF_SIGMA_ONLY | flags,
                             B, xyz, xyz_stride, rays, z_vals, spr, nullptr, 3, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                             planes, stream);
}
extern "C" int mnrf_tcnn_encode(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                                int base_resolution, float bound, int64_t B, const float* xyz, int64_t xyz_stride,
                                const float* rays, const float* z_vals, int spr, float* planes, void* stream) {
    return mnrf_tcnn_encode_flags(table, offsets17_host, log2_per_level_scale, base_resolution, bound, B, xyz, xyz_stride, rays, z_vals,
                                  spr, planes, 0u, stream);
}

// fp32 master table -> the half2 copy the MNRF_TCNN_TABLE_F16 launches read (round to nearest even, like a .half() cast)
__global__ void tcnn_table_half_kernel(const float2* __restrict__ src, __half2* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __float22half2_rn(src[i]);
}
extern "C" int mnrf_tcnn_table_half(const float* table, int64_t entries, void* table_half, void* stream) {
    if (!table || !table_half || entries < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_tcnn_table_half: bad argument");
    if (entries == 0) return MNRF_OK;
    hipLaunchKernelGGL(tcnn_table_half_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)table, (__half2*)table_half, (long long)entries);
    return mnrf_check_launch("mnrf_tcnn_table_half");
}

extern "C" int mnrf_bench_gather(const void* table, int64_t table_bytes, int bytes_per_gather, int64_t n_threads, int iters,
                                 float* out, void* stream) {
    if (!table || !out || table_bytes < 64 || n_threads < 256 || iters < 8 || (bytes_per_gather != 4 && bytes_per_gather != 8))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_bench_gather: bad argument");
    const unsigned blocks = (unsigned)(n_threads / 256);
    iters = iters / 8 * 8;
    if (bytes_per_gather == 8)
        hipLaunchKernelGGL(gather_bench_kernel<float2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float2*)table,
                           (unsigned)(table_bytes / 8), iters, out);
    else
        hipLaunchKernelGGL(gather_bench_kernel<unsigned>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)table,
                           (unsigned)(table_bytes / 4), iters, out);
    return mnrf_check_launch("mnrf_bench_gather");
}
