// mnrf_field.hip -- fused  positions -> encoding -> 8x256 trunk -> {sigma, rgb, normal, mirror}
// heads (-> closed-form density gradient) for gfx950.
//
// Replaces, per sample, models/mirror_nerf.py:101-212 (MirrorNeRF.forward and its four
// forward_* chains), the Embedding of mirror_nerf.py:20-38, the position generation of
// models/rendering.py:302 and the chunked model loop of rendering.py:134-179.
//
// Shape of the computation (see mnrf_layout.h for the operand layout):
//   * one workgroup = 4 waves = one wave per SIMD; every wave owns 32 samples (two groups
//     of 16) for the whole network and keeps their activations in registers;
//   * every Linear is Out^T = W . In^T on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain);
//     accumulators of layer l are directly the B operands of layer l+1;
//   * the 2.7 MB weight image streams L2 -> LDS in 8 KiB chunks with global_load_lds
//     (double buffered, one barrier per chunk = per 4096 MFMA cycles) and is shared by the
//     four waves; A operands come from LDS with one conflict-free ds_read_b128 per 8 MFMAs.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mnrf_layout.h"

namespace mnrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S = 2;                       // 16-sample groups per wave
constexpr int WAVES = 4;
constexpr int WG_THREADS = 64 * WAVES;
constexpr int WG_SAMPLES = WAVES * S * 16;  // 128
constexpr int RING_SLOTS = 3;
constexpr int LDS_RING = RING_SLOTS * CHUNK_BYTES;  // staging ring
constexpr int LDS_BIAS = LDS_RING;         // bias block
constexpr int LDS_MASK = LDS_BIAS + BIAS_FLOATS * 4;   // relu masks [8 layers][S][256 threads] x 8 B (GRAD only)
constexpr int LDS_BYTES = LDS_MASK;
constexpr int LDS_BYTES_GRAD = LDS_MASK + 8 * S * WG_THREADS * 8;

struct FieldArgs {
    const float* packed;
    unsigned flags;
    long long B;
    const float* xyz;
    long long xyz_stride;
    const float* rays;
    const float* z_vals;
    int spr;
    const float* dir_emb;
    long long dir_stride;
    float* sigma;
    float* rgb;
    float* pred_normal;
    float* is_mirror;
    float* normal;
    float* geo_feat;
};

// ------------------------------------------------------------------ weight stream
// The packed image is consumed strictly in order, 8 KiB chunk by chunk, through a ring of
// three LDS slots: one being read, one landed, one in flight.  `advance` is called PF tiles
// before the first read of a chunk, so A operands can be prefetched across chunk seams.
struct Stream {
    const char* src;   // this lane's source of chunk 0 (wave and lane offsets included)
    int next;          // next chunk to issue
    int end;           // chunks in this stream
};

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ void issue_chunk(const Stream& st, int chunk, int wave) {
    const char* g = st.src + (long long)chunk * CHUNK_BYTES;
    char* l = smem + (chunk % RING_SLOTS) * CHUNK_BYTES + wave * 2 * TILE_BYTES;
    // two 1 KiB pieces per wave: lane i's 16 bytes land at l + 16*i (wave-uniform base)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + TILE_BYTES),
                                     (__attribute__((address_space(3))) void*)(l + TILE_BYTES), 16, 0, 0);
}

// Make chunk (st.next-1) readable and start the copy of chunk st.next.  The slot that copy
// overwrites held chunk st.next-3, which every wave finished before reaching this barrier.
__device__ __forceinline__ void advance(Stream& st, int wave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of chunk next-1 have landed
    __syncthreads();                                    // so have everybody else's
    if (st.next < st.end) issue_chunk(st, st.next, wave);
    st.next += 1;
}

__device__ __forceinline__ void open_stream(Stream& st, const float* base, int n_tiles, int wave, int lane) {
    __syncthreads();   // all waves are done with the ring (previous stream)
    st.src = (const char*)base + wave * 2 * TILE_BYTES + lane * 16;
    st.end = n_tiles / CHUNK_TILES;
    st.next = 1;
    issue_chunk(st, 0, wave);
}

// ------------------------------------------------------------------ GEMM building block
// acc[s][nb] += sum over the part's k-steps of  A(tile) x b[s][4*tq + j].
// `tile0` = index in the stream of the part's first tile (a multiple of CHUNK_TILES); every
// part starts on a chunk seam.  A operands are read PF tiles ahead of their MFMAs.
constexpr int PF = 2;

__device__ __forceinline__ f32x4 read_tile(int tile, int lane16) {
    return *(const f32x4*)(smem + (tile % (RING_SLOTS * CHUNK_TILES)) * TILE_BYTES + lane16);
}

template <int NTQ, int NB, int NACC, int NT>
__device__ __forceinline__ void gemm_part(f32x4 (&acc)[S][NACC], const float (&b)[S][NT], Stream& st,
                                          int& tile0, int wave, int lane16) {
    constexpr int NTILES = NTQ * NB;
    static_assert(NTILES % CHUNK_TILES == 0, "parts start and end on chunk seams");
    f32x4 a[PF + 1];
    advance(st, wave);
#pragma unroll
    for (int i = 0; i < PF; ++i) a[i] = read_tile(tile0 + i, lane16);
#pragma unroll
    for (int tq = 0; tq < NTQ; ++tq) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int t = tq * NB + nb;
            if (t + PF < NTILES) {
                if ((t + PF) % CHUNK_TILES == 0) advance(st, wave);
                a[(t + PF) % (PF + 1)] = read_tile(tile0 + t + PF, lane16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t % (PF + 1)][j], b[s][4 * tq + j],
                                                                      acc[s][nb], 0, 0, 0);
                }
            }
        }
    }
    tile0 += NTILES;
}

// acc[s][nb][r] = bias[16*nb + 4*g + r]   (bias block lives in LDS behind the ring)
template <int NB, int NACC>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[S][NACC], int bias_off, int g) {
    const f32x4* bl = (const f32x4*)(smem + LDS_BIAS) + (bias_off >> 2);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const f32x4 v = bl[nb * 4 + g];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][nb] = v;
    }
}

template <int NB, int NACC>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[S][NACC]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// h = relu(acc); with WANT_MASK the 64-bit relu mask of the layer (bit t <=> the
// pre-activation feeding k-step t is > 0) is parked in LDS for the gradient pass.
template <bool WANT_MASK>
__device__ __forceinline__ void relu_to(float (&h)[S][64], const f32x4 (&acc)[S][16], int layer, int tid) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint64_t m = 0;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[s][nb][r];
                if (WANT_MASK) m |= (uint64_t)(v > 0.f) << (4 * nb + r);
                h[s][4 * nb + r] = fmaxf(v, 0.f);
            }
        }
        if (WANT_MASK) ((uint64_t*)(smem + LDS_MASK))[(layer * S + s) * WG_THREADS + tid] = m;
    }
}

__device__ __forceinline__ void apply_mask(float (&gr)[S][64], int layer, int tid) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint64_t m = ((const uint64_t*)(smem + LDS_MASK))[(layer * S + s) * WG_THREADS + tid];
#pragma unroll
        for (int t = 0; t < 64; ++t) gr[s][t] = ((m >> t) & 1ull) ? gr[s][t] : 0.f;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------ the kernel
template <bool SIGMA_ONLY, bool GRAD>
__global__ __launch_bounds__(WG_THREADS, 1) void field_kernel(FieldArgs A) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;     // lane group = k-slot / row quad
    const int m = lane & 15;     // sample within the group
    const int lane16 = lane * 16;

    // bias block -> LDS (read by init_bias, visible after the first advance() barrier)
    {
        const f32x4* src = (const f32x4*)(A.packed + OFF_BIAS);
        f32x4* dst = (f32x4*)(smem + LDS_BIAS);
        for (int i = tid; i < BIAS_FLOATS / 4; i += WG_THREADS) dst[i] = src[i];
    }

    Stream st;
    open_stream(st, A.packed + OFF_FWD, SIGMA_ONLY ? FWD_TILES_SIGMA : FWD_TILES, wave, lane);

    // ---- sample positions (rendering.py:302: multiply, then add -- no FMA)
    long long idx[S];
    bool valid[S];
    float x[S][3];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        long long i = (long long)blockIdx.x * WG_SAMPLES + wave * (S * 16) + s * 16 + m;
        valid[s] = i < A.B;
        if (!valid[s]) i = A.B - 1;
        idx[s] = i;
        if (A.xyz) {
            const float* p = A.xyz + i * A.xyz_stride;
            x[s][0] = p[0]; x[s][1] = p[1]; x[s][2] = p[2];
        } else {
            const long long ray = i / A.spr;
            const float* r = A.rays + ray * 8;
            const float z = A.z_vals[i];
#pragma unroll
            for (int a = 0; a < 3; ++a) x[s][a] = r[a] + r[3 + a] * z;
        }
    }

    // ---- xyz encoding, this lane's 16 of the 64 (padded) channels: pairs P = 8g + pp
    float enc[S][16];
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int P = 8 * g + pp;
            const int f = P / 3;
            const int a = P - 3 * f;
            const float xa = a == 0 ? x[s][0] : (a == 1 ? x[s][1] : x[s][2]);
            float sn, cs;
            sincosf(ldexpf(xa, f), &sn, &cs);   // 2^f * x is exact (mirror_nerf.py:17, 36)
            if (P >= 30) {                       // raw coordinates ride in the last two pairs
                sn = P == 30 ? x[s][0] : x[s][2];
                cs = P == 30 ? x[s][1] : 0.f;
            }
            enc[s][2 * pp] = sn;
            enc[s][2 * pp + 1] = cs;
        }
    }

    int tile0 = 0;
    float h[S][64];
    {
        // ---- L1: enc(64) -> 256
        f32x4 acc[S][16];
        __syncthreads();   // bias block visible
        init_bias<16>(acc, BIAS_L, g);
        gemm_part<4, 16>(acc, enc, st, tile0, wave, lane16);
        relu_to<GRAD>(h, acc, 0, tid);
        // ---- L2..L4
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            init_bias<16>(acc, BIAS_L + 256 * (1 + l), g);
            gemm_part<16, 16>(acc, h, st, tile0, wave, lane16);
            relu_to<GRAD>(h, acc, 1 + l, tid);
        }
        // ---- L5: cat[enc, h] -> 256  (encoding first: mirror_nerf.py:192-193)
        init_bias<16>(acc, BIAS_L + 256 * 4, g);
        gemm_part<4, 16>(acc, enc, st, tile0, wave, lane16);
        gemm_part<16, 16>(acc, h, st, tile0, wave, lane16);
        relu_to<GRAD>(h, acc, 4, tid);
        // ---- L6..L8
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            init_bias<16>(acc, BIAS_L + 256 * (5 + l), g);
            gemm_part<16, 16>(acc, h, st, tile0, wave, lane16);
            relu_to<GRAD>(h, acc, 5 + l, tid);
        }
    }
    // h = geo_feat (mirror_nerf.py:195)
    if (A.geo_feat) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (valid[s]) {
                f32x4* o = (f32x4*)(A.geo_feat + idx[s] * 256);
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
                    o[nb * 4 + g] = f32x4{h[s][4 * nb], h[s][4 * nb + 1], h[s][4 * nb + 2], h[s][4 * nb + 3]};
            }
        }
    }
    // ---- sigma: 256 -> 1 (row 0 of a padded 16-row block lives in lane group 0, reg 0)
    {
        f32x4 acc[S][1];
        init_bias<1>(acc, BIAS_SIG, g);
        gemm_part<16, 1>(acc, h, st, tile0, wave, lane16);
        if (A.sigma && g == 0) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (valid[s]) A.sigma[idx[s]] = acc[s][0][0];
        }
    }

    if (!SIGMA_ONLY) {
        // ---- predicted normal: 256 -> 128 -> 3, no activation in between (mirror_nerf.py:85-88)
        {
            float hn[S][32];
            {
                f32x4 acc[S][8];
                init_bias<8>(acc, BIAS_NRM1, g);
                gemm_part<16, 8>(acc, h, st, tile0, wave, lane16);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[s][4 * nb + r] = acc[s][nb][r];
            }
            f32x4 acc[S][1];
            init_bias<1>(acc, BIAS_NRM2, g);
            gemm_part<8, 1>(acc, hn, st, tile0, wave, lane16);
            if (A.pred_normal && g == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (valid[s]) {
                        const float a0 = acc[s][0][0], a1 = acc[s][0][1], a2 = acc[s][0][2];
                        // utils/func.py:5-7: eps clamps the squared norm
                        const float inv = 1.f / sqrtf(fmaxf(a0 * a0 + a1 * a1 + a2 * a2, 1.1920928955078125e-07f));
                        float* o = A.pred_normal + idx[s] * 3;
                        o[0] = a0 * inv; o[1] = a1 * inv; o[2] = a2 * inv;
                    }
            }
        }
        // ---- mirror probability: 256 -> 128 LeakyReLU(0.01) -> 1 sigmoid (mirror_nerf.py:94-99)
        {
            float hm[S][32];
            {
                f32x4 acc[S][8];
                init_bias<8>(acc, BIAS_MIR1, g);
                gemm_part<16, 8>(acc, h, st, tile0, wave, lane16);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[s][nb][r];
                            hm[s][4 * nb + r] = v > 0.f ? v : 0.01f * v;
                        }
            }
            f32x4 acc[S][1];
            init_bias<1>(acc, BIAS_MIR2, g);
            gemm_part<8, 1>(acc, hm, st, tile0, wave, lane16);
            if (A.is_mirror && g == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (valid[s]) A.is_mirror[idx[s]] = sigmoidf_(acc[s][0][0]);
            }
        }
        // ---- colour: final(256->256, no act) ; cat[final, dir] -> 128 relu ; 128 -> 3 sigmoid
        {
            float fin[S][64];
            {
                f32x4 acc[S][16];
                init_bias<16>(acc, BIAS_FIN, g);
                gemm_part<16, 16>(acc, h, st, tile0, wave, lane16);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) fin[s][4 * nb + r] = acc[s][nb][r];
            }
            float de[S][8];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float* dp = A.dir_emb + (idx[s] / A.spr) * A.dir_stride;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int e = 16 * (t >> 2) + 4 * g + (t & 3);
                    de[s][t] = e < ENC_DIR ? dp[e] : 0.f;
                }
            }
            float hd[S][32];
            {
                f32x4 acc[S][8];
                init_bias<8>(acc, BIAS_DIR, g);
                gemm_part<16, 8>(acc, fin, st, tile0, wave, lane16);
                gemm_part<2, 8>(acc, de, st, tile0, wave, lane16);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) hd[s][4 * nb + r] = fmaxf(acc[s][nb][r], 0.f);
            }
            f32x4 acc[S][1];
            init_bias<1>(acc, BIAS_RGB, g);
            gemm_part<8, 1>(acc, hd, st, tile0, wave, lane16);
            if (A.rgb && g == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (valid[s]) {
                        float* o = A.rgb + idx[s] * 3;
                        o[0] = sigmoidf_(acc[s][0][0]);
                        o[1] = sigmoidf_(acc[s][0][1]);
                        o[2] = sigmoidf_(acc[s][0][2]);
                    }
            }
        }
    }

    if (GRAD) {
        // ---- d sigma / d xyz in closed form: g = w_sigma; for i = 8..1:
        //      g = (g * relu_mask_i) W_i, the 63 encoding columns of layer 5 and layer 1 feed g_enc.
        open_stream(st, A.packed + OFF_BWD, BWD_TILES, wave, lane);
        tile0 = 0;
        float gr[S][64];
        {
            const f32x4* ws = (const f32x4*)(smem + LDS_BIAS) + (BIAS_WSIG >> 2);
#pragma unroll
            for (int nb = 0; nb < 16; ++nb) {
                const f32x4 v = ws[nb * 4 + g];
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gr[s][4 * nb + r] = v[r];
            }
        }
        float genc[S][16];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < 16; ++t) genc[s][t] = 0.f;

        // layers 8, 7, 6
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            apply_mask(gr, 7 - l, tid);
            f32x4 acc[S][16];
            zero_acc<16>(acc);
            gemm_part<16, 16>(acc, gr, st, tile0, wave, lane16);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gr[s][4 * nb + r] = acc[s][nb][r];
        }
        // layer 5: first its 4 encoding row blocks, then its 16 hidden row blocks
        {
            apply_mask(gr, 4, tid);
            {
                f32x4 acc[S][4];
                zero_acc<4>(acc);
                gemm_part<16, 4>(acc, gr, st, tile0, wave, lane16);
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) genc[s][4 * nb + r] = acc[s][nb][r];
            }
            f32x4 acc[S][16];
            zero_acc<16>(acc);
            gemm_part<16, 16>(acc, gr, st, tile0, wave, lane16);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gr[s][4 * nb + r] = acc[s][nb][r];
        }
        // layers 4, 3, 2
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            apply_mask(gr, 3 - l, tid);
            f32x4 acc[S][16];
            zero_acc<16>(acc);
            gemm_part<16, 16>(acc, gr, st, tile0, wave, lane16);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gr[s][4 * nb + r] = acc[s][nb][r];
        }
        // layer 1: 4 encoding row blocks
        {
            apply_mask(gr, 0, tid);
            f32x4 acc[S][4];
            zero_acc<4>(acc);
            gemm_part<16, 4>(acc, gr, st, tile0, wave, lane16);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) genc[s][4 * nb + r] += acc[s][nb][r];
        }
        // encoding Jacobian: d/dx_a = g[x_a] + sum_f 2^f (g[sin] cos - g[cos] sin)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int P = 8 * g + pp;
                const int f = P / 3;
                const int a = P - 3 * f;
                const float gs = genc[s][2 * pp], gc = genc[s][2 * pp + 1];
                if (P < 30) {
                    const float xa = a == 0 ? x[s][0] : (a == 1 ? x[s][1] : x[s][2]);
                    float sn, cs;
                    sincosf(ldexpf(xa, f), &sn, &cs);   // recomputed: cheaper than 32 live registers
                    const float c = ldexpf(gs * cs - gc * sn, f);
                    d0 += a == 0 ? c : 0.f;
                    d1 += a == 1 ? c : 0.f;
                    d2 += a == 2 ? c : 0.f;
                } else if (P == 30) {
                    d0 += gs;
                    d1 += gc;
                } else {
                    d2 += gs;
                }
            }
            d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16); d2 += __shfl_xor(d2, 16);
            d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
            if (A.normal && g == 0 && valid[s]) {
                // normal = l2_normalize(-grad)  (mirror_nerf.py:145-146)
                const float n0 = -d0, n1 = -d1, n2 = -d2;
                const float inv = 1.f / sqrtf(fmaxf(n0 * n0 + n1 * n1 + n2 * n2, 1.1920928955078125e-07f));
                float* o = A.normal + idx[s] * 3;
                o[0] = n0 * inv; o[1] = n1 * inv; o[2] = n2 * inv;
            }
        }
    }
}

// ------------------------------------------------------------------ weight packer
struct PackArgs {
    const float* params[32];
    float* packed;
};

struct PartTable {
    Part fwd[N_FWD_PARTS];
    Part bwd[N_BWD_PARTS];
};

__host__ __device__ inline void build_parts(PartTable& T) {
    int n = 0, tile = 0;
    auto add = [&](Part* arr, int& cnt, int param, int n_true, int ld, int ntq, int nb, int col_off, int kind) {
        arr[cnt] = Part{param, n_true, ld, ntq, nb, col_off, kind, tile};
        tile += ntq * nb;
        cnt++;
    };
    add(T.fwd, n, 0, 256, 63, 4, 16, 0, KIND_ENC);                             // L1
    for (int i = 1; i < 4; ++i) add(T.fwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // L2..L4
    add(T.fwd, n, 8, 256, 319, 4, 16, 0, KIND_ENC);                            // L5 encoding columns
    add(T.fwd, n, 8, 256, 319, 16, 16, 63, KIND_H);                            // L5 hidden columns
    for (int i = 5; i < 8; ++i) add(T.fwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // L6..L8
    add(T.fwd, n, 20, 1, 256, 16, 1, 0, KIND_H);                               // sigma
    add(T.fwd, n, 24, 128, 256, 16, 8, 0, KIND_H);                             // normal_net.0
    add(T.fwd, n, 26, 3, 128, 8, 1, 0, KIND_H);                                // normal_net.1
    add(T.fwd, n, 28, 128, 256, 16, 8, 0, KIND_H);                             // is_mirror_net.0
    add(T.fwd, n, 30, 1, 128, 8, 1, 0, KIND_H);                                // is_mirror_net.2
    add(T.fwd, n, 16, 256, 256, 16, 16, 0, KIND_H);                            // xyz_encoding_final
    add(T.fwd, n, 18, 128, 283, 16, 8, 0, KIND_H);                             // dir_encoding: final part
    add(T.fwd, n, 18, 128, 283, 2, 8, 256, KIND_DIR);                          // dir_encoding: view part
    add(T.fwd, n, 22, 3, 128, 8, 1, 0, KIND_H);                                // rgb
    // backward: A = W_i^T, rows = input columns of layer i, contraction over its 256 outputs
    n = 0;
    tile = 0;
    for (int i = 7; i >= 5; --i) add(T.bwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // layers 8,7,6
    add(T.bwd, n, 8, 256, 319, 16, 4, 0, KIND_ENC);                             // layer 5: 4 encoding row blocks
    add(T.bwd, n, 8, 256, 319, 16, 16, ENC_XYZ, KIND_H);                        // layer 5: 16 hidden row blocks
    for (int i = 3; i >= 1; --i) add(T.bwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // layers 4,3,2
    add(T.bwd, n, 0, 256, 63, 16, 4, 0, KIND_ENC);                              // layer 1: 4 encoding row blocks
}

__global__ void pack_kernel(PackArgs P, PartTable T) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PACKED_FLOATS) return;
    float v = 0.f;
    if (p < OFF_BIAS) {
        // ---- forward tiles
        const int tile = (int)(p / TILE_FLOATS);
        const int within = (int)(p % TILE_FLOATS);
        const int lane = within >> 2, j = within & 3, g = lane >> 4, i = lane & 15;
        int k = 0;
        while (k + 1 < N_FWD_PARTS && T.fwd[k + 1].tile0 <= tile) ++k;
        const Part pt = T.fwd[k];
        const int lt = tile - pt.tile0;
        const int tq = lt / pt.nb, nb = lt % pt.nb;
        const int n = 16 * nb + i;
        const int t = 4 * tq + j;
        int col;
        if (pt.kind == KIND_ENC) col = enc_col(t, g);
        else {
            col = 16 * (t >> 2) + 4 * g + (t & 3);
            if (pt.kind == KIND_DIR && col >= ENC_DIR) col = -1;
        }
        if (n < pt.n_true && col >= 0) v = P.params[pt.param][(long long)n * pt.ld + pt.col_off + col];
    } else if (p < OFF_BWD) {
        // ---- bias block
        const int b = (int)(p - OFF_BIAS);
        if (b < 2048) v = P.params[2 * (b / 256) + 1][b % 256];
        else if (b < BIAS_FIN) { if (b - BIAS_SIG < 1) v = P.params[21][0]; }
        else if (b < BIAS_DIR) v = P.params[17][b - BIAS_FIN];
        else if (b < BIAS_RGB) v = P.params[19][b - BIAS_DIR];
        else if (b < BIAS_NRM1) { if (b - BIAS_RGB < 3) v = P.params[23][b - BIAS_RGB]; }
        else if (b < BIAS_NRM2) v = P.params[25][b - BIAS_NRM1];
        else if (b < BIAS_MIR1) { if (b - BIAS_NRM2 < 3) v = P.params[27][b - BIAS_NRM2]; }
        else if (b < BIAS_MIR2) v = P.params[29][b - BIAS_MIR1];
        else if (b < BIAS_WSIG) { if (b - BIAS_MIR2 < 1) v = P.params[31][0]; }
        else if (b < BIAS_WSIG + 256) v = P.params[20][b - BIAS_WSIG];
    } else {
        // ---- backward tiles: A[row rho][contraction c] = W_i[c][column(rho)]
        const long long q = p - OFF_BWD;
        const int tile = (int)(q / TILE_FLOATS);
        const int within = (int)(q % TILE_FLOATS);
        const int lane = within >> 2, j = within & 3, g = lane >> 4, i = lane & 15;
        int k = 0;
        while (k + 1 < N_BWD_PARTS && T.bwd[k + 1].tile0 <= tile) ++k;
        const Part pt = T.bwd[k];
        const int lt = tile - pt.tile0;
        const int tq = lt / pt.nb, nb = lt % pt.nb;
        const int rho = 16 * nb + i;                 // output row of the transposed product
        const int c = 16 * tq + 4 * g + j;           // contraction index = output unit of layer i
        int col;
        if (pt.kind == KIND_ENC) {
            // C-form row rho = 16*nbe + 4*g' + r  <->  k-step t' = 4*nbe + r of lane group g'
            const int nbe = rho >> 4, gq = (rho >> 2) & 3, r = rho & 3;
            col = enc_col(4 * nbe + r, gq);
        } else {
            col = pt.col_off + rho;
        }
        if (col >= 0) v = P.params[pt.param][(long long)c * pt.ld + col];
    }
    P.packed[p] = v;
}

}  // namespace mnrf

// ====================================================================== C ABI
#include "../../include/mnrf.h"
#include "mnrf_error.h"

using namespace mnrf;

extern "C" int64_t mnrf_packed_floats(void) { return PACKED_FLOATS; }

extern "C" int mnrf_pack_weights(const float* const* params, float* packed, void* stream) {
    if (!params || !packed) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null pointer");
    PackArgs P;
    for (int i = 0; i < MNRF_N_PARAMS; ++i) {
        if (!params[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null parameter pointer");
        P.params[i] = params[i];
    }
    P.packed = packed;
    PartTable T;
    build_parts(T);
    const int threads = 256;
    const int blocks = (int)((PACKED_FLOATS + threads - 1) / threads);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, P, T);
    return mnrf_check_launch("mnrf_pack_weights");
}

extern "C" int mnrf_field_forward(const float* packed, unsigned flags, int64_t B, const float* xyz,
                                  int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                                  const float* dir_emb, int64_t dir_stride, float* sigma, float* rgb,
                                  float* pred_normal, float* is_mirror, float* normal, float* geo_feat,
                                  void* stream) {
    if (!packed) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: packed weights are null");
    if (B < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: negative sample count");
    if (B == 0) return MNRF_OK;
    const bool sigma_only = flags & MNRF_SIGMA_ONLY;
    const bool grad = flags & MNRF_GRAD_NORMAL;
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: need xyz or rays+z_vals");
    if (xyz && xyz_stride < 3) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: xyz_stride < 3");
    if (spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: samples per ray must be >= 1");
    if (!xyz && B % spr != 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: B not a multiple of spr");
    if (!sigma_only && !dir_emb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: dir_emb required unless SIGMA_ONLY");
    if (grad && !normal) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: GRAD_NORMAL needs the normal output");
    FieldArgs A{packed, flags, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, dir_emb,
                (long long)dir_stride, sigma, rgb, pred_normal, is_mirror, normal, geo_feat};
    const int64_t blocks64 = (B + WG_SAMPLES - 1) / WG_SAMPLES;
    if (blocks64 > 0x7fffffff) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: too many samples for one launch");
    const dim3 grid((unsigned)blocks64), block(WG_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (sigma_only && !grad) hipLaunchKernelGGL((field_kernel<true, false>), grid, block, LDS_BYTES, s, A);
    else if (sigma_only && grad) hipLaunchKernelGGL((field_kernel<true, true>), grid, block, LDS_BYTES_GRAD, s, A);
    else if (!grad) hipLaunchKernelGGL((field_kernel<false, false>), grid, block, LDS_BYTES, s, A);
    else hipLaunchKernelGGL((field_kernel<false, true>), grid, block, LDS_BYTES_GRAD, s, A);
    return mnrf_check_launch("mnrf_field_forward");
}
