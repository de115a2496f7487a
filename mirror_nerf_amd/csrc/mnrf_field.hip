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
#include <stdlib.h>

#include "mnrf_layout.h"
#include "mnrf_field_args.h"

namespace mnrf {

extern __shared__ __attribute__((aligned(16))) char smem[];

// Two tunings of the same kernel body:
//   s2: 32 samples per wave, one wave per SIMD (<= 512 registers), 128-sample workgroups
//   s1: 16 samples per wave, <= 256 registers so two workgroups share a CU and overlap each
//       other's barriers and epilogues, 64-sample workgroups
namespace s2 {
constexpr int S = 2;
constexpr int MIN_WAVES_PER_SIMD = 1;
constexpr int CHUNK_TILES = 16;   // 32-tile chunks (half the barriers) were tried: 137.4 vs 139.0 TFLOP/s
#include "mnrf_field_impl.inc"
#include "mnrf_field_bwd.inc"
}  // namespace s2
namespace s1 {
constexpr int S = 1;
constexpr int MIN_WAVES_PER_SIMD = 2;
constexpr int CHUNK_TILES = 16;   // two workgroups per CU share the 160 KiB of LDS
#include "mnrf_field_impl.inc"
}  // namespace s1

// ------------------------------------------------------------------ weight packer
struct PackArgs {
    const float* params[32];
    float* packed;
};
// up to PACK_BATCH models per launch (blockIdx.y = model): a training step re-packs its coarse and its fine model behind
// every optimizer step -- one launch instead of one per model (mnrf_pack_weights_n)
constexpr int PACK_BATCH = 4;
struct PackBatch {
    PackArgs m[PACK_BATCH];
};

__global__ void pack_kernel(PackBatch PB, PartTable T) {
    const PackArgs& P = PB.m[blockIdx.y];
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < DEVICE_STATE_WORDS) P.packed[OFF_REDUCE_PAIR + p] = 0.f;      // device state of the launches that use the image (mnrf_layout.h)
    if (p >= PACKED_F32_FLOATS) return;
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
        const int tq = lt / pt.nb, nb = lt % pt.nb;   // tq >= ntq: chunk padding behind a short part
        const int n = 16 * nb + i;
        const int t = 4 * tq + j;
        int col;
        if (pt.kind == KIND_ENC) col = enc_col(t, g);
        else {
            col = 16 * (t >> 2) + 4 * g + (t & 3);
            if (pt.kind == KIND_DIR && col >= ENC_DIR) col = -1;
        }
        if (tq < pt.ntq && n < pt.n_true && col >= 0) v = P.params[pt.param][(long long)n * pt.ld + pt.col_off + col];
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
        // ---- transposed tiles (density-gradient trunk stream, then the head-backward stream):
        //      A[row rho][contraction c] = W[c][column(rho)]
        const bool heads = p >= OFF_HBWD;
        const long long q = p - (heads ? OFF_HBWD : OFF_BWD);
        const int tile = (int)(q / TILE_FLOATS);
        const int within = (int)(q % TILE_FLOATS);
        const int lane = within >> 2, j = within & 3, g = lane >> 4, i = lane & 15;
        const Part* parts = heads ? T.hbwd : T.bwd;
        const int nparts = heads ? N_HBWD_PARTS - 1 : N_BWD_PARTS;
        int k = 0;
        while (k + 1 < nparts && parts[k + 1].tile0 <= tile) ++k;
        const Part pt = parts[k];
        const int lt = tile - pt.tile0;
        const int tq = lt / pt.nb, nb = lt % pt.nb;   // tq >= ntq: chunk padding
        const int rho = 16 * nb + i;                 // output row of the transposed product
        const int c = 16 * tq + 4 * g + j;           // contraction index = row of W
        int col;
        if (pt.kind == KIND_ENC) {
            // C-form row rho = 16*nbe + 4*g' + r  <->  k-step t' = 4*nbe + r of lane group g'
            const int nbe = rho >> 4, gq = (rho >> 2) & 3, r = rho & 3;
            col = enc_col(4 * nbe + r, gq);
        } else if (pt.kind == KIND_DIR) {
            col = rho < ENC_DIR ? pt.col_off + rho : -1;
        } else {
            col = pt.col_off + rho;
        }
        if (tq < pt.ntq && c < pt.n_true && col >= 0) v = P.params[pt.param][(long long)c * pt.ld + col];
    }
    P.packed[p] = v;
}

}  // namespace mnrf

// ====================================================================== C ABI
#include "../../include/mnrf.h"
#include "mnrf_dw.h"
#include "mnrf_error.h"
#include "mnrf_fill.h"

using namespace mnrf;

extern "C" int64_t mnrf_packed_floats(void) { return PACKED_FLOATS; }

extern "C" int mnrf_pack_weights_n(int n_models, const float* const* params, float* const* packed, void* stream) {
    if (n_models < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights_n: negative model count");
    if (n_models == 0) return MNRF_OK;
    if (!params || !packed) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null pointer");
    for (int i = 0; i < n_models * MNRF_N_PARAMS; ++i)
        if (!params[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null parameter pointer");
    for (int m = 0; m < n_models; ++m)
        if (!packed[m]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null image pointer");
    PartTable T;
    build_parts(T);
    const int threads = 256;
    const int blocks = (int)((PACKED_F32_FLOATS + threads - 1) / threads);
    for (int m0 = 0; m0 < n_models; m0 += PACK_BATCH) {
        const int nb = n_models - m0 < PACK_BATCH ? n_models - m0 : PACK_BATCH;
        PackBatch PB{};
        float* images[PACK_BATCH] = {};
        for (int m = 0; m < nb; ++m) {
            for (int i = 0; i < MNRF_N_PARAMS; ++i) PB.m[m].params[i] = params[(m0 + m) * MNRF_N_PARAMS + i];
            PB.m[m].packed = images[m] = packed[m0 + m];
        }
        // (the device-state words at the image's end -- reduction pair, tile-queue pairs, range-guard word -- are zeroed by pack_kernel
        // itself: split_pack_kernel, which may raise MNRF_GUARD_WEIGHT, runs behind it on the stream)
        hipLaunchKernelGGL(pack_kernel, dim3(blocks, nb), dim3(threads), 0, (hipStream_t)stream, PB, T);
        launch_split_pack(images, nb, (hipStream_t)stream);   // hi/lo f16 streams of the split tunings, from the fp32 tiles
        if (split32_enabled())      // stream of the 32x32x16 tuning (MNRF_SPLIT32=1 only)
            for (int m = 0; m < nb; ++m) launch_split32_pack(params + (m0 + m) * MNRF_N_PARAMS, images[m], (hipStream_t)stream);
    }
    return mnrf_check_launch("mnrf_pack_weights");
}

extern "C" int mnrf_pack_weights(const float* const* params, float* packed, void* stream) {
    if (!params || !packed) return mnrf_fail(MNRF_ERR_ARG, "mnrf_pack_weights: null pointer");
    return mnrf_pack_weights_n(1, params, &packed, stream);
}

extern "C" int mnrf_field_forward(float* packed, unsigned flags, int64_t B, const float* xyz,
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
    // Tunings.  fp32 MFMA (bit-exact fmaf chains): s2 (32 samples/wave, one wave per SIMD) is fastest for the
    // forward-only kernels, s1 (16 samples/wave, two workgroups per CU) for the ones with the density-gradient
    // pass.  MNRF_SPLIT_F16 selects the split-f16 tuning (fp32 operands as hi/lo f16 pairs on the f16 matrix
    // pipe).  MNRF_FIELD_VARIANT=s1|s2|h|h2|hx forces one (experiments; h = split with its own default).
    static const int forced = [] {
        const char* e = getenv("MNRF_FIELD_VARIANT");
        if (e && e[0] == 's' && (e[1] == '1' || e[1] == '2')) return e[1] - '0';
        if (e && e[0] == 'h') return e[1] == 'x' ? 5 : (e[1] == '2' ? 4 : 3);
        return 0;
    }();
    const int variant = forced ? forced : ((flags & MNRF_SPLIT_F16) ? 3 + (int)((flags >> 3) & 3u) : (grad ? 1 : 2));   // bits 3-4: experimental split tunings
    const int rc = variant >= 3 ? launch_split(A, sigma_only, grad, variant - 3, (hipStream_t)stream)
                 : variant == 1 ? s1::launch(A, sigma_only, grad, (hipStream_t)stream)
                                : s2::launch(A, sigma_only, grad, (hipStream_t)stream);
    if (rc != 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward: too many samples for one launch");
    return mnrf_check_launch("mnrf_field_forward");
}

// ---------------------------------------------------------------------- ray-fused fine pass (eval, maps only)
extern "C" int mnrf_fused_samples_per_ray(void) { return split48_ray_samples(); }

extern "C" int mnrf_field_composite_fused(float* packed, int64_t n_rays, const float* rays, const float* z_vals,
                                          const float* dir_emb, int64_t dir_stride, int white_back,
                                          float* weights, float* opacity, float* rgb_map, float* depth, float* mirror_mask,
                                          float* surf_normal, float* x_surface, void* stream) {
    if (!packed || !rays || !z_vals || !dir_emb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_composite_fused: null pointer");
    if (n_rays < 0) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_composite_fused: negative ray count");
    if (n_rays == 0) return MNRF_OK;
    if (!split48_enabled() || split32_enabled())
        return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_field_composite_fused: needs the 48-samples-per-wave tuning (MNRF_SPLIT48 != 0, MNRF_SPLIT32 unset)");
    const int spr = split48_ray_samples();
    FieldArgs A{packed, MNRF_SPLIT_F16, (long long)n_rays * spr, nullptr, 3, rays, z_vals, spr, dir_emb, (long long)dir_stride,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    A.fuse = 1;
    A.white_back = white_back;
    A.f_weights = weights; A.f_opacity = opacity; A.f_rgb_map = rgb_map; A.f_depth = depth; A.f_mirror_mask = mirror_mask;
    A.f_surf_normal = surf_normal; A.f_x_surface = x_surface;
    if (launch_split48(A, false, (hipStream_t)stream) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_composite_fused: too many samples for one launch");
    return mnrf_check_launch("mnrf_field_composite_fused");
}

// ---------------------------------------------------------------------- training entry points
static inline int64_t train_tiles(int64_t B) { return (B + s2::WG_SAMPLES - 1) / s2::WG_SAMPLES; }

extern "C" int64_t mnrf_train_save_floats(int64_t B) { return (int64_t)SAVE_FLOATS * B; }
extern "C" int64_t mnrf_train_mask_words(int64_t B) { return train_tiles(B) * N_MASKS * s2::S * s2::WG_THREADS; }
extern "C" int64_t mnrf_train_workspace_floats(int64_t B) { return (int64_t)DY_FLOATS * B + dw_workspace_floats(B); }

#ifdef MNRF_EXP_CYCLES      // experiment builds only (scripts/exp_train_marks.py): where the s_memtime marks of the training forward go
static void* g_exp_marks = nullptr;
extern "C" void mnrf_exp_set_marks(void* p) { g_exp_marks = p; }
#endif
static int field_forward_train_impl(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                    const float* rays, const float* z_vals, int spr, const float* dir_emb,
                                    int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                    float* is_mirror, float* normal, float* save_x, uint64_t* save_mask,
                                    float* save_inv, float* save_invj, unsigned flags, const int32_t* n_live, void* stream) {
    if (n_live && !(flags & MNRF_TRAIN_PLANES))
        return mnrf_fail(MNRF_ERR_UNSUPPORTED, "mnrf_field_forward_train_n: a live row count needs MNRF_TRAIN_PLANES (the fp32-row layout depends on B)");
    if (!packed || !save_x || !save_mask || !save_inv) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: null pointer");
    const bool planes = flags & MNRF_TRAIN_PLANES;
    if (planes && !(flags & MNRF_SPLIT_F16))
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: MNRF_TRAIN_PLANES needs MNRF_SPLIT_F16 (the planes are the split kernel's operands)");
    if (B <= 0) return B == 0 ? MNRF_OK : mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: negative sample count");
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: need xyz or rays+z_vals");
    if (spr < 1 || !dir_emb) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: bad spr / dir_emb");
    if (!sigma || !rgb || !pred_normal || !is_mirror)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: the four head outputs are required (the backward reads them)");
    FieldArgs A{packed, MNRF_GRAD_NORMAL, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, dir_emb,
                (long long)dir_stride, sigma, rgb, pred_normal, is_mirror, normal, nullptr,
                planes ? nullptr : save_x, (unsigned long long*)save_mask, save_inv, save_invj, planes ? (char*)save_x : nullptr};
    A.n_live = n_live;
#ifdef MNRF_EXP_CYCLES
    A.geo_feat = (float*)g_exp_marks;
#endif
    // always a 128-sample tiling with the mask-producing (GRAD) body: the backward kernel shares its tile map.
    // MNRF_SPLIT_F16: the split-f16 tuning (same saved quantities, fp32 activations from its fp32 accumulators)
    const int rc = (flags & MNRF_SPLIT_F16) ? launch_split(A, false, true, 0, (hipStream_t)stream)
                                            : s2::launch(A, false, true, (hipStream_t)stream);
    if (rc != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_forward_train: too many samples for one launch");
    return mnrf_check_launch("mnrf_field_forward_train");
}
extern "C" int mnrf_field_forward_train(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                        const float* rays, const float* z_vals, int spr, const float* dir_emb,
                                        int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                        float* is_mirror, float* normal, float* save_x, uint64_t* save_mask,
                                        float* save_inv, float* save_invj, unsigned flags, void* stream) {
    return field_forward_train_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, dir_emb, dir_stride, sigma, rgb, pred_normal, is_mirror,
                                    normal, save_x, save_mask, save_inv, save_invj, flags, nullptr, stream);
}
extern "C" int mnrf_field_forward_train_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                          const float* rays, const float* z_vals, int spr, const float* dir_emb,
                                          int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                                          float* is_mirror, float* normal, float* save_x, uint64_t* save_mask,
                                          float* save_inv, float* save_invj, unsigned flags, const int32_t* n_live, void* stream) {
    return field_forward_train_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, dir_emb, dir_stride, sigma, rgb, pred_normal, is_mirror,
                                    normal, save_x, save_mask, save_inv, save_invj, flags, n_live, stream);
}

extern "C" int mnrf_field_backward(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                   const float* rays, const float* z_vals, int spr,
                                   const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                   const float* g_is_mirror, const float* rgb, const float* pred_normal,
                                   const float* is_mirror, const float* save_x, const uint64_t* save_mask,
                                   const float* save_inv, float* workspace, float* const* d_params, float* d_xyz,
                                   float* d_dir, const float* keep_mirror, unsigned flags, void* stream) {
    if (!packed || !save_x || !save_mask || !save_inv || !workspace || !d_params)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: null pointer");
    if (B <= 0) return B == 0 ? MNRF_OK : mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: negative sample count");
    if (!g_sigma || !g_rgb || !g_pred_normal || !g_is_mirror)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: all four upstream gradients are required (pass zeros)");
    if (!rgb || !pred_normal || !is_mirror) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: forward outputs missing");
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: need xyz or rays+z_vals");
    for (int i = 0; i < MNRF_N_PARAMS; ++i)
        if (!d_params[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: null gradient pointer");
    float* dY = workspace;
    float* ws = workspace + (int64_t)DY_FLOATS * B;
    FieldBwdArgs A{packed, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, g_sigma, g_rgb, g_pred_normal,
                       g_is_mirror, rgb, pred_normal, is_mirror, (const unsigned long long*)save_mask, save_inv, dY, d_xyz, d_dir,
                       flags & (MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD), keep_mirror, nullptr, nullptr, 6};
    hipStream_t s = (hipStream_t)stream;
    if (((flags & MNRF_SPLIT_F16) ? launch_split_bwd(A, s) : s2::launch_bwd(A, s)) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: too many samples for one launch");
    int rc = mnrf_check_launch("mnrf_field_backward (activation gradients)");
    if (rc != MNRF_OK) return rc;
    if (launch_dw(save_x, dY, g_sigma, (long long)B, ws, d_params, (flags & MNRF_DW_ACCUMULATE) ? 1 : 0, s) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward: workspace accounting error");
    return mnrf_check_launch("mnrf_field_backward (weight gradients)");
}

// ---------------------------------------------------------------------- round 3: operand planes
extern "C" int64_t mnrf_train_planes_bytes(int64_t B) { return B <= 0 ? 0 : dwp_sample_blocks(B) * PLX_SB_BYTES; }
extern "C" int64_t mnrf_train_dy_planes_bytes(int64_t B) { return B <= 0 ? 0 : dwp_sample_blocks(B) * PLY_SB_BYTES; }

static int field_backward_planes_impl(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                      const float* rays, const float* z_vals, int spr,
                                      const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                      const float* g_is_mirror, const float* rgb, const float* pred_normal,
                                      const float* is_mirror, const uint64_t* save_mask, const float* save_inv,
                                      void* dy_planes, uint32_t* seedmax, float* d_xyz, float* d_dir,
                                      const float* keep_mirror, unsigned flags, const int32_t* n_live, void* stream) {
    if (!packed || !save_mask || !save_inv || !dy_planes || !seedmax)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: null pointer");
    if (B <= 0) return B == 0 ? MNRF_OK : mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: negative sample count");
    if (!rgb || !pred_normal || !is_mirror) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: forward outputs missing");
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: need xyz or rays+z_vals");
    hipStream_t s = (hipStream_t)stream;
    if (spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: bad spr");
    launch_seed_max(g_sigma, g_rgb, g_pred_normal, g_is_mirror, rgb, pred_normal, is_mirror, save_inv, (long long)B, seedmax, s, n_live, spr,
                    (unsigned*)(packed + OFF_REDUCE_PAIR));
    FieldBwdArgs A{packed, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, g_sigma, g_rgb, g_pred_normal,
                   g_is_mirror, rgb, pred_normal, is_mirror, (const unsigned long long*)save_mask, save_inv, nullptr, d_xyz, d_dir,
                   flags & (MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD), keep_mirror, (char*)dy_planes, seedmax,
                   6 - (int)((flags >> 16) & 0xfu)};
    A.n_live = n_live;
    A.y_half = (flags & MNRF_PLANES_Y_HALF) ? 1 : 0;
    if (launch_split_bwd(A, s) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward_planes: too many samples for one launch");
    return mnrf_check_launch("mnrf_field_backward_planes");
}
extern "C" int mnrf_field_backward_planes(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                          const float* rays, const float* z_vals, int spr,
                                          const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                          const float* g_is_mirror, const float* rgb, const float* pred_normal,
                                          const float* is_mirror, const uint64_t* save_mask, const float* save_inv,
                                          void* dy_planes, uint32_t* seedmax, float* d_xyz, float* d_dir,
                                          const float* keep_mirror, unsigned flags, void* stream) {
    return field_backward_planes_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, g_sigma, g_rgb, g_pred_normal, g_is_mirror, rgb,
                                      pred_normal, is_mirror, save_mask, save_inv, dy_planes, seedmax, d_xyz, d_dir, keep_mirror, flags,
                                      nullptr, stream);
}
extern "C" int mnrf_field_backward_planes_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                            const float* rays, const float* z_vals, int spr,
                                            const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                            const float* g_is_mirror, const float* rgb, const float* pred_normal,
                                            const float* is_mirror, const uint64_t* save_mask, const float* save_inv,
                                            void* dy_planes, uint32_t* seedmax, float* d_xyz, float* d_dir,
                                            const float* keep_mirror, unsigned flags, const int32_t* n_live, void* stream) {
    return field_backward_planes_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, g_sigma, g_rgb, g_pred_normal, g_is_mirror, rgb,
                                      pred_normal, is_mirror, save_mask, save_inv, dy_planes, seedmax, d_xyz, d_dir, keep_mirror, flags,
                                      n_live, stream);
}

extern "C" int64_t mnrf_dw_planes_workspace_floats(int n_eval, const int64_t* B) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL || !B) return 0;
    return dwp_workspace_floats(n_eval, B, nullptr);
}
extern "C" int64_t mnrf_dw_planes2_workspace_floats(int n_eval, const int64_t* B, const int* kinds) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL || !B) return 0;
    return dwp_workspace_floats(n_eval, B, kinds);
}

extern "C" int mnrf_dw_planes(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                              const uint32_t* const* seedmax, float* workspace, float* const* d_params, int accumulate,
                              void* stream) {
    return mnrf_dw_planes2(n_eval, x_planes, dy_planes, B, seedmax, nullptr, workspace, d_params, accumulate, stream);
}

extern "C" int mnrf_dw_planes2(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                               const uint32_t* const* seedmax, const int* kinds, float* workspace, float* const* d_params,
                               int accumulate, void* stream) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: 1..8 evaluations per call");
    if (!x_planes || !dy_planes || !B || !seedmax || !workspace || !d_params) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: null pointer");
    long long total = 0;
    for (int e = 0; e < n_eval; ++e) {
        if (B[e] <= 0 || !x_planes[e] || !dy_planes[e] || !seedmax[e])
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: every evaluation needs B > 0 and its three buffers");
        total += dwp_sample_blocks(B[e]);
    }
    if (total * (812 + DWP_JOBS * DWP_STAGE_KIB) >= (1LL << 31)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: too many samples for one call (split the evaluations)");
    for (int i = 0; i < MNRF_N_PARAMS; ++i)
        if (!d_params[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: null gradient pointer");
    if (launch_dwp(n_eval, x_planes, dy_planes, B, (const unsigned* const*)seedmax, kinds, workspace, d_params, accumulate ? 1 : 0,
                   (hipStream_t)stream) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes: bad plan");
    return mnrf_check_launch("mnrf_dw_planes");
}

extern "C" int64_t mnrf_dw_planes2_n_workspace_floats(int n_eval) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL) return 0;
    return dwp_workspace_floats_n(n_eval);
}
extern "C" int mnrf_dw_planes2_n(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                                 const int32_t* const* n_live, const int* spr, const uint32_t* const* seedmax, const int* kinds,
                                 float* workspace, float* const* d_params, int accumulate, void* stream) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: 1..8 evaluations per call");
    if (!x_planes || !dy_planes || !B || !seedmax || !workspace || !d_params || !n_live || !spr)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: null pointer");
    long long total = 0;
    for (int e = 0; e < n_eval; ++e) {
        if (B[e] <= 0 || spr[e] < 1 || !x_planes[e] || !dy_planes[e] || !seedmax[e])
            return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: every evaluation needs a capacity B > 0, spr >= 1 and its three buffers");
        total += dwp_sample_blocks(B[e]);
    }
    if (total * (812 + DWP_JOBS * DWP_STAGE_KIB) >= (1LL << 31)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: too many samples for one call (split the evaluations)");
    for (int i = 0; i < MNRF_N_PARAMS; ++i)
        if (!d_params[i]) return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: null gradient pointer");
    if (launch_dwp_n(n_eval, x_planes, dy_planes, B, n_live, spr, (const unsigned* const*)seedmax, kinds, workspace, d_params,
                     accumulate ? 1 : 0, (hipStream_t)stream) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_dw_planes2_n: bad plan");
    return mnrf_check_launch("mnrf_dw_planes2_n");
}

extern "C" int64_t mnrf_train_planes2_bytes(int64_t B) { return B <= 0 ? 0 : dwp_sample_blocks(B) * PL2X_SB_BYTES; }
extern "C" int64_t mnrf_train_dy_planes2_bytes(int64_t B) { return B <= 0 ? 0 : dwp_sample_blocks(B) * PL2Y_SB_BYTES; }

static int field_backward2_planes_impl(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                       const float* rays, const float* z_vals, int spr, const float* g_normal,
                                       const float* normal, const float* save_invj, const uint64_t* save_mask,
                                       void* x2_planes, void* y2_planes, uint32_t* jmax, float* d_xyz, const int32_t* n_live,
                                       void* stream) {
    if (!packed || !g_normal || !normal || !save_invj || !save_mask || !x2_planes || !y2_planes || !jmax)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2_planes: null pointer");
    if (B <= 0) return B == 0 ? MNRF_OK : mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2_planes: negative sample count");
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2_planes: need xyz or rays+z_vals");
    hipStream_t s = (hipStream_t)stream;
    if (spr < 1) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2_planes: bad spr");
    launch_jhat_max(g_normal, normal, save_invj, (long long)B, jmax, s, n_live, spr, (unsigned*)(packed + OFF_REDUCE_PAIR));
    FieldBwd2Args A{packed, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, g_normal, normal, save_invj,
                    (const unsigned long long*)save_mask, nullptr, d_xyz, (char*)x2_planes, (char*)y2_planes, jmax};
    A.n_live = n_live;
    if (launch_split_bwd2(A, s) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2_planes: too many samples for one launch");
    return mnrf_check_launch("mnrf_field_backward2_planes");
}
extern "C" int mnrf_field_backward2_planes(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                           const float* rays, const float* z_vals, int spr, const float* g_normal,
                                           const float* normal, const float* save_invj, const uint64_t* save_mask,
                                           void* x2_planes, void* y2_planes, uint32_t* jmax, float* d_xyz, void* stream) {
    return field_backward2_planes_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, g_normal, normal, save_invj, save_mask, x2_planes,
                                       y2_planes, jmax, d_xyz, nullptr, stream);
}
extern "C" int mnrf_field_backward2_planes_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                             const float* rays, const float* z_vals, int spr, const float* g_normal,
                                             const float* normal, const float* save_invj, const uint64_t* save_mask,
                                             void* x2_planes, void* y2_planes, uint32_t* jmax, float* d_xyz,
                                             const int32_t* n_live, void* stream) {
    return field_backward2_planes_impl(packed, B, xyz, xyz_stride, rays, z_vals, spr, g_normal, normal, save_invj, save_mask, x2_planes,
                                       y2_planes, jmax, d_xyz, n_live, stream);
}

extern "C" int64_t mnrf_train_workspace2_floats(int64_t B) { return (int64_t)SO_FLOATS * B + dw2_workspace_floats(B); }

extern "C" int mnrf_field_backward2(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                    const float* rays, const float* z_vals, int spr, const float* g_normal,
                                    const float* normal, const float* save_invj, const uint64_t* save_mask,
                                    float* workspace, float* const* d_params, float* d_xyz, unsigned flags, void* stream) {
    if (!packed || !g_normal || !normal || !save_invj || !save_mask || !workspace || !d_params)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2: null pointer");
    if (B <= 0) return B == 0 ? MNRF_OK : mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2: negative sample count");
    if (!xyz && (!rays || !z_vals)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2: need xyz or rays+z_vals");
    float* so = workspace;
    float* ws = workspace + (int64_t)SO_FLOATS * B;
    FieldBwd2Args A{packed, (long long)B, xyz, (long long)xyz_stride, rays, z_vals, spr, g_normal, normal, save_invj,
                        (const unsigned long long*)save_mask, so, d_xyz};
    hipStream_t s = (hipStream_t)stream;
    if (((flags & MNRF_SPLIT_F16) ? launch_split_bwd2(A, s) : s2::launch_bwd2(A, s)) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2: too many samples for one launch");
    int rc = mnrf_check_launch("mnrf_field_backward2 (tangent pass)");
    if (rc != MNRF_OK) return rc;
    if (launch_dw2(so, (long long)B, ws, d_params, s) != 0)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_field_backward2: workspace accounting error");
    return mnrf_check_launch("mnrf_field_backward2 (weight gradients)");
}
