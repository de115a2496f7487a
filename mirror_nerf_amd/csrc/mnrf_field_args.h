// mnrf_field_args.h -- argument block of the field kernels, shared by the fp32 (mnrf_field.hip) and the
// split-f16 (mnrf_field_split.hip) translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mnrf.h"
#include "mnrf_layout.h"
#include "mnrf_dwp.h"

namespace mnrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    // training forward: activations / relu masks / normal-head norm kept for the backward pass
    float* save_x;                 // [SAVE_FLOATS sections][B][width]   (null = inference)
    unsigned long long* save_mask; // [tiles][N_MASKS][S][256]
    float* save_inv;               // [B] 1/|v| of normal_net (negative when the eps clamp was active)
    float* save_invj;              // [B] 1/|d sigma/dx| of the density-gradient normal, same convention
    // split arithmetic, round 3: instead of fp32 rows in save_x the inputs of every Linear are kept as the hi/lo f16 operand
    // PLANES the weight-gradient GEMM consumes directly (mnrf_dwp.h): [sample blocks][PLX_FB][hi | lo][1 KiB]
    char* save_planes;
    // ray-fused fine pass (eval, maps only: round 3).  spr == 192 samples of ONE ray per workgroup of the 48-samples-per-wave
    // tuning: the four head outputs go to LDS instead of HBM and the workgroup composites its ray itself (mnrf_composite.inc,
    // the body of composite_kernel: identical maps); sigma / rgb / pred_normal / is_mirror above are then unused.
    int fuse;                      // 1: fused compositing on
    // dynamic tile queue (48-samples-per-wave kernels, round 3): one resident workgroup per CU takes 192-sample tiles from a
    // global counter instead of one workgroup per tile -- the hardware deals workgroup ids to the 8 XCDs round robin, so a
    // static grid ends when the SLOWEST XCD has worked off its eighth (the XCDs of one package run up to 9 % apart under
    // this kernel's power draw); with the queue a faster XCD simply takes more tiles.
    int* tile_queue;               // {next, done}, both zero between launches (the last workgroup out resets them); null: static grid
    int n_tiles;                   // ceil(B / 192)
    int resident;                  // workgroups launched (<= n_tiles): what the chip holds at once
    int white_back;
    float* f_weights;              // (n_rays, spr) or null
    float* f_opacity; float* f_rgb_map; float* f_depth; float* f_mirror_mask; float* f_surf_normal; float* f_x_surface;
    // live row count (round 5, include/mnrf.h "live row counts on the device"): device int32 or null.  B is then the CAPACITY the
    // launch and the buffers are sized for, the kernel evaluates the first *n_live * spr samples (workgroups past them leave at
    // once).  Training forward with operand planes only (the fp32-row layouts depend on B).
    const int* n_live;
};

// training backward (activation gradients): mnrf_field_bwd.inc (fp32) and mnrf_field_split_bwd.inc (split-f16)
struct FieldBwdArgs {
    const float* packed;
    long long B;
    const float* xyz; long long xyz_stride; const float* rays; const float* z_vals; int spr;
    const float* g_sigma; const float* g_rgb; const float* g_pn; const float* g_m;   // upstream, null = 0
    const float* rgb; const float* pn; const float* is_mirror;                        // forward outputs
    const unsigned long long* save_mask; const float* save_inv;
    float* dY;                  // fp32 rows [DY_* sections][B][width] (null with dY_planes)
    float* d_xyz;
    float* d_dir;
    // gradient steering (models/mirror_nerf.py:154-183): heads that see geo_feat.detach() still get their own weight
    // gradients, but add nothing to dL/dh8
    unsigned cut;               // MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD
    const float* keep_mirror;   // per ray (per sample with xyz) or null: 0 = cut the mirror head for this ray's samples
    // split arithmetic, round 3: the pre-activation gradients as operand planes (mnrf_dwp.h) under ONE power-of-two scale
    // per launch, derived from the largest seed magnitude of the evaluation (*seedmax, float bits, filled by seed_max_kernel)
    char* dY_planes;            // [sample blocks][PLY_FB][hi | lo][1 KiB]
    const unsigned* seedmax;
    // where the per-sample scale puts a sample's largest seed: [2^seed_log2, 2^(seed_log2 + 1)).  6 by default; a training loop
    // lowers it when scaled gradients outgrow the f16 range (trained weights amplify them on the way down the trunk) -- the
    // weight-gradient GEMM's finish kernel must be told the same number (mnrf_dw_planes2: kinds)
    int seed_log2;
    const int* n_live;          // live rows (B = capacity): see FieldArgs::n_live; planes route only
    int y_half;                 // round 6, opt-in (MNRF_PLANES_Y_HALF): only the hi tiles of the dY planes reach memory
};
// second-order pass (gradient through the density-gradient normal)
struct FieldBwd2Args {
    const float* packed;
    long long B;
    const float* xyz; long long xyz_stride; const float* rays; const float* z_vals; int spr;
    const float* g_normal;     // dL/d normal (B,3)
    const float* normal;       // forward output (B,3)
    const float* save_invj;    // 1/|J| (negative: eps clamp)
    const unsigned long long* save_mask;
    float* so;                 // [SO_FLOATS sections][B][width]   (null with the planes below)
    float* d_xyz;              // accumulated into (may be null)
    // round 4: tangents / signals as operand planes (mnrf_dwp.h "second-order pass") under one power-of-two scale per launch,
    // derived from the largest |J^| of the evaluation (*jmax, float bits, filled by jhat_max_kernel)
    char* x2_planes;           // [sample blocks][PL2X_FB][hi | lo][1 KiB]
    char* y2_planes;           // [sample blocks][PL2Y_FB][hi | lo][1 KiB]
    const unsigned* jmax;
    const int* n_live;         // live rows (B = capacity): see FieldArgs::n_live; planes route only
};

// split-f16 tunings (mnrf_field_split.hip).  variant: 0 = default.
int launch_split(const FieldArgs& A, bool sigma_only, bool grad, int variant, hipStream_t s);
// builds the split streams of a packed image from its fp32 streams (same stream, after pack_kernel)
void launch_split_pack(float* const* packed, int n_images, hipStream_t s);      // up to 4 images per launch
int launch_split_bwd(const FieldBwdArgs& A, hipStream_t s);
// 32x32x16 tuning of the forward-only split kernels (mnrf_field_split32.hip) and the packer of its stream (from the
// state_dict-ordered parameter pointers)
// 48-samples-per-wave tuning of the forward-only split kernels (mnrf_field_split3.hip), MNRF_SPLIT48=1
bool split48_enabled();
int launch_split48(const FieldArgs& A, bool sigma_only, hipStream_t s);
int split48_ray_samples();      // samples of one workgroup of that tuning = samples per ray of the ray-fused fine pass (192)
bool split32_enabled();      // MNRF_SPLIT32=1
int launch_split32(const FieldArgs& A, bool sigma_only, hipStream_t s);
void launch_split32_pack(const float* const* params, float* packed, hipStream_t s);
int launch_split_bwd2(const FieldBwd2Args& A, hipStream_t s);

}  // namespace mnrf
