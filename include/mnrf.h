/*
 * mnrf.h -- C ABI of the MI355X-native Mirror-NeRF rendering hot path.
 *
 * The reference has no C/FFI boundary on this path: its boundary is two Python
 * call signatures,
 *     render_rays(models, embeddings, rays, N_samples, use_disp, perturb, noise_std,
 *                 N_importance, chunk, white_back, test_time, **kwargs)   models/rendering.py:54-67
 *     MirrorNeRF.forward(x, compute_normal, sigma_only, embedding_xyz, ...)  models/mirror_nerf.py:101-112
 * which `mirror_nerf_amd/rendering.py` and `mirror_nerf_amd/mirror_nerf.py` keep.
 * Those Python shims do no arithmetic; every number is produced by the entry
 * points below (libmnrf_hip.so), which take plain device pointers and sizes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every
 *     buffer and the library never allocates device memory.  A packed weight image is
 *     read-write: its last 19 words are device state of the launches that use it (a
 *     {value, done} pair of the training backward's single-launch reductions, 16 counters
 *     of the field kernels' dynamic tile queue -- all zero between launches -- and the
 *     range-guard word below); mnrf_pack_weights zeroes them.  Hence `float* packed`, not const, in every entry
 *     point that launches on an image: an image must not live in read-only or IPC-shared memory;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - return value 0 = ok, negative = error (see mnrf_last_error()); nothing throws;
 *   - re-entrant; no global stream or global device state (host side: a cached CU count
 *     per device and a rotating index into the 8 counter pairs of an image -- at most 8
 *     tile-queue launches that share one image may be in flight on different streams);
 *     every launch is capturable in a hipGraph.
 *   - the field architecture is the reference default (train.py:44-66): D=8, W=256,
 *     skip at layer 5, Embedding(10) for xyz (63 ch), Embedding(4) for dir (27 ch),
 *     normal_net and is_mirror_net present.
 */
#ifndef MNRF_H
#define MNRF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNRF_OK 0
#define MNRF_ERR_ARG (-1)      /* bad argument (null pointer, size, flag combination) */
#define MNRF_ERR_LAUNCH (-2)   /* HIP reported an error at launch */
#define MNRF_ERR_UNSUPPORTED (-3)

/* number of parameter tensors of one MirrorNeRF in state_dict order
 * (xyz_encoding_{1..8}.0.{weight,bias}, xyz_encoding_final, dir_encoding.0, sigma,
 *  rgb.0, normal_net.{0,1}, is_mirror_net.{0,2});  utils/__init__.py:109-136 */
#define MNRF_N_PARAMS 32

/* flags of mnrf_field_forward */
#define MNRF_SIGMA_ONLY 1u     /* stop after sigma (rendering.py:139-150) */
#define MNRF_GRAD_NORMAL 2u    /* also emit normal = l2n(-d sigma/d xyz) (mirror_nerf.py:136-146) */
#define MNRF_SPLIT_F16 4u      /* evaluate the Linears on the f16 matrix pipe with every fp32 operand carried as a
                                  hi/lo f16 pair (3 MFMAs per product block, fp32 accumulation; ~2^-20 relative per
                                  product instead of the bit-exact fp32 fmaf chain of the default) */

#define MNRF_TCNN_GRAD_F16 16u  /* mnrf_tcnn_backward: the levels without private copies accumulate their table gradient scaled by 2^10 in
                                  half2 with ONE packed atomic per entry (global_atomic_pk_add_f16) instead of two fp32 atomics --
                                  tinycudann's gradient precision (models/mirror_nerf_tcnn.py:36-49 under train.py:586); the
                                  workspace then has mnrf_tcnn_backward_workspace_floats2() floats */
#define MNRF_TCNN_GRAD_FIXED 512u /* mnrf_tcnn_backward: the levels without private copies accumulate both features of a table entry with ONE
                                  64-bit integer atomic (two 32-bit fixed-point halves under a per-level power-of-two scale taken
                                  from the step's own sum of gradient magnitudes, under which no entry can overflow; exact integer
                                  sums, order-independent; step ~2^-16 of a level's largest contribution); the scatter runs as a second
                                  launch over dL/d encoding planes.  Workspace: mnrf_tcnn_backward_workspace_floats3() floats, of
                                  which the caller zero-fills the first mnrf_tcnn_backward_workspace_floats().  Ignored together
                                  with MNRF_TCNN_GRAD_F16. */
#define MNRF_TCNN_F16 256u      /* mnrf_tcnn_forward: single-pass f16 MLPs -- operands rounded to f16, ONE MFMA per product, fp32 accumulation:
                                  "fp16 MLP on CDNA4 MFMA" as BASELINE config 5 words it and as the reference computes under
                                  tinycudann / precision=16 (train.py:586); ~1e-3 relative; sigma-only launches then take the matrix pipe too */
#define MNRF_TCNN_TABLE_F16 1024u /* mnrf_tcnn_forward / _backward / _encode_flags: `table` points to HALF2 entries (4 B per entry: tinycudann's
                                  storage, models/gridencoder/grid.py:57-58, mirror_nerf_tcnn.py:39-49; SURVEY 8d prices config 5 at 16 x 8 x 4 =
                                  512 B of gathers per sample) made from the fp32 master table by mnrf_tcnn_table_half; gradients still go
                                  to the fp32 d_table (the master the optimizer steps, as tinycudann keeps fp32 master parameters) */
#define MNRF_TCNN_VALU 8u       /* mnrf_tcnn_forward: evaluate the small MLPs with fp32 FMAs on the VALU, one thread per sample
                                  (the first implementation; default: hi/lo f16 tiles on the matrix pipe, ~1e-6 of it) */

/* gradient steering of the training backward (the reference's --detach_density_* options; values are unaffected) */
#define MNRF_CUT_NORMAL_HEAD 32u   /* mnrf_field_backward: normal_net sees geo_feat.detach() (mirror_nerf.py:154-158) */
#define MNRF_CUT_MIRROR_HEAD 64u   /* mnrf_field_backward: is_mirror_net sees geo_feat.detach() (mirror_nerf.py:169-170) */
#define MNRF_DW_ACCUMULATE 128u     /* mnrf_field_backward: ADD the parameter gradients to d_params instead of overwriting them (a
                                      module evaluated more than once per step -- primary and reflected rays -- then needs
                                      no separate accumulation kernels) */
#define MNRF_DETACH_W_MASK 1       /* mnrf_composite_backward: mirror mask = sum(weights.detach() * is_mirror) (rendering.py:223-226) */
#define MNRF_DETACH_W_NORMAL 2     /* mnrf_composite_backward: the normal outputs use weights.detach() (rendering.py:244-264) */

/* Range guard of the split-f16 arithmetic.  The LAST 32-bit word of a packed weight image (index
 * mnrf_packed_floats() - 1) is a sticky device flag: mnrf_pack_weights clears it, every MNRF_SPLIT_F16 kernel ORs in */
#define MNRF_GUARD_SATURATED 1u    /* an operand of a Linear reached the f16 maximum: hi/lo pairs no longer carry fp32 */
#define MNRF_GUARD_WEIGHT 2u       /* a weight is non-finite or >= 65504 in magnitude (set by mnrf_pack_weights) */
#define MNRF_GUARD_IN_FORWARD 128u       /* with SATURATED, which pass: ... an activation of a forward evaluation */
#define MNRF_GUARD_IN_BACKWARD 256u      /* ... a scaled activation gradient of the training backward (mnrf_field_backward_planes: the
                                            caller may lower the gradient scale instead of leaving the split arithmetic, see its flags) */
#define MNRF_GUARD_IN_SECOND_ORDER 512u  /* ... a scaled tangent or signal of the second-order pass (mnrf_field_backward2*) */
#define MNRF_GUARD_ENC_RANGE 4u    /* a position with |x| >= 64: sin/cos arguments beyond 2^15, outside the fast exact reduction */
/* The fp32 kernels never touch it.  Host policy (mirror_nerf_amd.mirror_nerf.check_guard): read it once per frame /
 * training step; non-zero -> the module is switched to the exact fp32 kernels and the work is repeated. */

const char* mnrf_last_error(void);
int mnrf_version(void);

/* Size in floats of the packed weight image of one model (forward stream, bias block,
 * transposed trunk stream for the density gradient). */
int64_t mnrf_packed_floats(void);

/* Re-lay one model's parameters into the MFMA-fragment order the field kernel streams
 * through LDS.  `params` is a HOST array of MNRF_N_PARAMS device pointers in state_dict
 * order (nn.Linear layout (out,in) row-major, models/mirror_nerf.py:59-99).
 * Replaces: the implicit weight reads of nn.Linear at call time. */
int mnrf_pack_weights(const float* const* params, float* packed, void* stream);
/* The same for n_models models at once (a training step re-packs its coarse and its fine model behind every optimizer step: one
 * launch pair instead of one per model): params = n_models x 32 pointers, model after model; packed = n_models images. */
int mnrf_pack_weights_n(int n_models, const float* const* params, float* const* packed, void* stream);

/* Embedding.forward (models/mirror_nerf.py:20-38): x (n, c) -> out (n, c*(2*n_freqs+1)). */
int mnrf_embed(const float* x, int64_t n, int c, int n_freqs, float* out, void* stream);

/* MirrorNeRF.forward on B samples (models/mirror_nerf.py:101-187).
 * Positions: either `xyz` (B rows, `xyz_stride` floats apart), or -- when xyz is null --
 * generated as o + d*z from `rays` (n_rays, 8) and `z_vals` (n_rays, spr) with B = n_rays*spr
 * (rendering.py:302; separate multiply and add).
 * View encoding: `dir_emb` rows of 27 floats, `dir_stride` floats apart; row index is
 * sample/spr (spr = 1 when every sample carries its own row, as in forward(x)).
 * Outputs may be null when not wanted: sigma (B), rgb (B,3), pred_normal (B,3),
 * is_mirror (B), normal (B,3), geo_feat (B,256). */
int mnrf_field_forward(float* packed, unsigned flags, int64_t B,
                       const float* xyz, int64_t xyz_stride,
                       const float* rays, const float* z_vals, int spr,
                       const float* dir_emb, int64_t dir_stride,
                       float* sigma, float* rgb, float* pred_normal, float* is_mirror,
                       float* normal, float* geo_feat, void* stream);

/* Coarse depths (rendering.py:283-300): z = near*(1-t)+far*t, or in disparity;
 * `z_steps` (n_samples) is torch.linspace(0,1,n) from the host (not recomputed: SURVEY 8a
 * hazard 2); `perturb_rand` (n_rays, n_samples) may be null (perturb == 0). */
int mnrf_sample_coarse(const float* rays, int64_t n_rays, const float* z_steps, int n_samples,
                       int use_disp, float perturb, const float* perturb_rand, float* z_vals,
                       void* stream);

/* Alpha compositing along each ray (rendering.py:181-264, 362-367).
 * Inputs per sample: sigma, z_vals (n_rays, S); optional noise (already scaled by noise_std);
 * optional rgb (.,3), is_mirror, pred_normal (.,3), normal (.,3).
 * Outputs (null = skip): weights (n_rays,S), opacity, rgb_map (.,3), depth, mirror_mask,
 * surf_normal (.,3) [sum w*pred_normal], surf_normal_grad (.,3) [sum w*normal],
 * normal_dif [sum w*|normal-pred_normal|^2], x_surface (.,3) = o + d*depth. */
int mnrf_composite(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                   const float* noise, const float* rgb, const float* is_mirror,
                   const float* pred_normal, const float* normal, int white_back,
                   float* weights, float* opacity, float* rgb_map, float* depth, float* mirror_mask,
                   float* surf_normal, float* surf_normal_grad, float* normal_dif, float* x_surface,
                   void* stream);

/* Hierarchical resampling (rendering.py:7-51, 312-326): inverse-CDF samples from
 * weights[:,1:-1] over the mid-points of z_coarse, merged and sorted with z_coarse.
 * `u`: (n_importance) shared by all rays when u_per_ray == 0 (torch.linspace(0,1,n), det=True)
 * or (n_rays, n_importance) when u_per_ray != 0.  z_fine: (n_rays, S + n_importance). */
int mnrf_sample_fine(const float* z_coarse, const float* weights, int64_t n_rays, int S,
                     const float* u, int u_per_ray, int n_importance, float* z_fine, void* stream);

/* Reflected-ray construction + order-preserving compaction
 * (train.py:192-252, eval.py:336-360, 513-548).
 * mask (n_rays) is the 0/1 (or soft) mirror mask; a ray is selected iff mask != 0 (the
 * `.bool()` of the reference).  compact == 0 keeps all rays (count = n_rays).
 * normal_noise (n_rays,3) may be null (eval.py:506-511 roughness).
 * Writes sec_rays (count, 8) = [x_surface, r, near2, far], index (count) int32 of the source
 * ray, *count (int32, device), and optionally reflect_dir (n_rays, 3) for all rays. */
int mnrf_reflect_compact(const float* rays, const float* x_surface, const float* normal,
                         const float* normal_noise, float noise_std, const float* mask,
                         int64_t n_rays, int compact, float near2, float* sec_rays, int32_t* index,
                         int32_t* count, float* reflect_dir, void* stream);

/* Threshold in place exactly like `m[m>0.5]=1; m[m<0.5]=0` (train.py:165-166, eval.py:305-306)
 * and OR-reduce `m != 0` into *any (int32, device, caller zeroes it). */
int mnrf_threshold_mask(float* mask, int64_t n, int32_t* any, void* stream);

/* Blend (train.py:261-296, eval.py:676-697): out = m*part + (1-m)*base, where part is
 * `sec` scattered through `index` (compacted, rows without a source keep `base`) or `sec`
 * itself (index == null).  Optionally writes reflect_out (n,c) = scattered sec (zeros elsewhere). */
int mnrf_blend_scatter(const float* base, const float* sec, const int32_t* index, int64_t n_sec,
                       const float* mask, int64_t n, int c, float* out, float* reflect_out,
                       void* stream);

/* Backward of mnrf_composite (training).  Inputs: the forward inputs (rays, sigma, z_vals, noise,
 * rgb, is_mirror, pred_normal, normal), the forward outputs weights (n_rays,S) and depth (n_rays),
 * and the upstream gradients of every per-ray output (null = zero): g_weights (n_rays,S), g_opacity,
 * g_rgb_map (.,3), g_depth, g_mirror_mask, g_surf_normal (.,3), g_surf_normal_grad (.,3),
 * g_normal_dif, g_x_surface (.,3).
 * Outputs (null = skip): d_sigma (n_rays,S), d_rgb (.,S,3), d_is_mirror (.,S), d_pred_normal (.,S,3),
 * d_normal (.,S,3), d_rays (n_rays,8) [origin and direction gradients through x_surface = o + d*depth].
 * Reference: autograd through models/rendering.py:181-264, 362-367. */
int mnrf_composite_backward(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                            const float* noise, const float* rgb, const float* is_mirror,
                            const float* pred_normal, const float* normal, int white_back,
                            const float* weights, const float* depth,
                            const float* g_weights, const float* g_opacity, const float* g_rgb_map,
                            const float* g_depth, const float* g_mirror_mask, const float* g_surf_normal,
                            const float* g_surf_normal_grad, const float* g_normal_dif, const float* g_x_surface,
                            float* d_sigma, float* d_rgb, float* d_is_mirror, float* d_pred_normal,
                            float* d_normal, float* d_rays,
                            int detach /* MNRF_DETACH_W_* : models/rendering.py:223-247 */,
                            const float* keep_mirror /* (n_rays) or null: 0 = this ray's mirror mask sees weights.detach() */,
                            void* stream);

/* Ray gradients of one field evaluation in ray mode (autograd of models/rendering.py:302 `x = o + d z` and of the per-ray view
 * encoding, rendering.py:275-277): g_rays (n_rays, 8) = [sum_s dL/dx_s | sum_s z_s dL/dx_s | 0 0] from d_xyz (n_rays*spr, 3) and
 * z_vals (n_rays, spr); g_de (n_rays, 27) = sum_s d_dir[s][:27] from d_dir (n_rays*spr, 32).  Either output may be null. */
int mnrf_ray_grads(const float* d_xyz, const float* z_vals, const float* d_dir, int64_t n_rays, int spr, float* g_rays,
                   float* g_de, void* stream);

/* ---- backward of the per-ray glue (training; autograd through train.py:217-296, mirror_nerf.py:20-38)
 * reflect: g_sec (n_sec,8) = dL/d secondary rays -> dL/dx_surface (n_rays,3), dL/d normal (n_rays,3),
 *          dL/d rays (n_rays,8: direction and far columns); rows without a selected ray get zeros.
 * blend:   g_out (n,c) -> dL/d base (n,c) = (1-m) g_out and dL/d sec (n_sec,c) = m g_out gathered by index.
 * embed:   g_out (n, c*(2N+1)) -> dL/dx (n,c). */
int mnrf_reflect_backward(const float* rays, const float* normal, const int32_t* index, int64_t n_sec,
                          const float* g_sec, int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays,
                          void* stream);
int mnrf_blend_backward(const float* g_out, const float* mask, const int32_t* index, int64_t n_sec, int64_t n, int c,
                        float* g_base, float* g_sec, void* stream);
int mnrf_embed_backward(const float* x, const float* g_out, int64_t n, int c, int n_freqs, float* g_x, void* stream);

/* ---- training (forward with saved activations, backward) -----------------------------------
 * Buffer sizes for B samples: activations (floats), ReLU bit masks (uint64 words), workspace of
 * the backward (floats: pre-activation gradients + split-K partials of the weight gradients). */
int64_t mnrf_train_save_floats(int64_t B);
int64_t mnrf_train_mask_words(int64_t B);
int64_t mnrf_train_workspace_floats(int64_t B);

/* mnrf_field_forward for training: all four heads (+ `normal` when non-null), and keeps in
 * save_x / save_mask / save_inv what mnrf_field_backward needs (no recomputation). */
int mnrf_field_forward_train(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                             const float* rays, const float* z_vals, int spr, const float* dir_emb,
                             int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                             float* is_mirror, float* normal, float* save_x, uint64_t* save_mask,
                             float* save_inv, float* save_invj, unsigned flags /* 0 or MNRF_SPLIT_F16 [| MNRF_TRAIN_PLANES] */, void* stream);

/* Backward of the field MLP: given dL/d{sigma (B), rgb (B,3), pred_normal (B,3), is_mirror (B)},
 * writes the gradient of every parameter (d_params: HOST array of MNRF_N_PARAMS device pointers,
 * state_dict order, overwritten), dL/dxyz (B,3) and dL/d(view encoding) (B,32 padded) when non-null.
 * The gradient flowing into `normal` (the normalised density gradient: a second-order term) is
 * handled by mnrf_field_backward2.
 * Autograd equivalent: loss.backward() through models/mirror_nerf.py:101-212. */
int mnrf_field_backward(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                        const float* rays, const float* z_vals, int spr,
                        const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                        const float* g_is_mirror, const float* rgb, const float* pred_normal,
                        const float* is_mirror, const float* save_x, const uint64_t* save_mask,
                        const float* save_inv, float* workspace, float* const* d_params, float* d_xyz,
                        float* d_dir,
                        const float* keep_mirror /* (B/spr) per ray [(B) with xyz] or null: 0 = the mirror head of this ray's
                                                    samples sees geo_feat.detach() (models/mirror_nerf.py:172-183) */,
                        unsigned flags /* MNRF_SPLIT_F16: activation gradients on the f16 pipe; MNRF_CUT_NORMAL_HEAD /
                                          MNRF_CUT_MIRROR_HEAD: that head sees geo_feat.detach() (mirror_nerf.py:157, 169-170);
                                          MNRF_DW_ACCUMULATE: add to d_params */,
                        void* stream);

/* Second-order term of the field backward: the gradient that reaches the trunk weights, sigma.weight
 * and xyz through `normal = l2n(-d sigma/d xyz)` (utils/func.py:10-25 with create_graph=True).
 * g_normal (B,3) = dL/d normal; normal = the forward output; save_invj (B) from the training forward.
 * ADDS to d_params (trunk weights and sigma.weight only) and to d_xyz (when non-null): call it after
 * mnrf_field_backward.  workspace: mnrf_train_workspace2_floats(B) floats. */
int64_t mnrf_train_workspace2_floats(int64_t B);
int mnrf_field_backward2(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                         const float* rays, const float* z_vals, int spr, const float* g_normal,
                         const float* normal, const float* save_invj, const uint64_t* save_mask,
                         float* workspace, float* const* d_params, float* d_xyz, unsigned flags /* 0 or MNRF_SPLIT_F16 */, void* stream);

/* The multiresolution hash encoding alone (what tinycudann's Encoding.forward returns, models/mirror_nerf_tcnn.py:225-227), level
 * by level: planes[level * B + sample] = the level's two features (float2), 32 * B floats.  Samples: rows of `xyz` (stride >= 3)
 * or rays + z_vals.  The first of the two launches of mnrf_tcnn_forward(enc_workspace != null); bench.py prices it against the
 * L2 roofline (`hash_grid_variant.gather_roofline`). */
int mnrf_tcnn_encode(const float* table, const int64_t* offsets17_host, double log2_per_level_scale, int base_resolution,
                     float bound, int64_t B, const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals,
                     int spr, float* planes, void* stream);

/* ... with flags (0 or MNRF_TCNN_TABLE_F16), and the half2 copy of a table those launches read (entries = offsets[16]). */
int mnrf_tcnn_encode_flags(const float* table, const int64_t* offsets17_host, double log2_per_level_scale, int base_resolution,
                           float bound, int64_t B, const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals,
                           int spr, float* planes, unsigned flags, void* stream);
int mnrf_tcnn_table_half(const float* table, int64_t entries, void* table_half, void* stream);

/* Measurement aid for the weight-gradient GEMM (mnrf_dw_planes): the rate at which one persistent 8-wave workgroup per CU reads
 * `bytes` of `buf` with the instruction that GEMM streams its operand planes with (global_load_lds_dwordx4, 1 KiB per
 * wave-instruction, `depth` = 8 or 16 in flight per wave, aux = 0 default policy / 2 non-temporal), computing nothing. */
int mnrf_bench_stream(const void* buf, int64_t bytes, int aux, int depth, void* stream);
/* ... and with the GEMM's address pattern: `windows` jobs, job-major; job w reads, per 32-sample stage s, chunk_a bytes at
 * a + s * stride_a + w * chunk_a and chunk_x bytes at x + s * stride_x + w * chunk_x (whole KiB; windows * n_stages >= 256 units
 * dealt to one workgroup per CU), 16 KiB in flight per wave, nt policy.  `barrier` is a bit set: 1 = a raw workgroup barrier per
 * stage like the GEMM's, 2 = the GEMM ring's half-tile lane pattern, 4 / 8 = (with 1 and 2) its stage-wise wait for all but 4 / 12
 * of a wave's requests instead of an instruction-wise wait for the oldest of 16 (scripts/bw_probe.py). */
int mnrf_bench_stream2(const void* a, const void* x, int n_stages, int64_t stride_a, int64_t stride_x, int chunk_a, int chunk_x,
                       int windows, int barrier, void* stream);

/* Measurement aid for the hash-grid field: the rate of independent random gathers of 8 B (a float2 table entry) or 4 B (what
 * an fp16 table would fetch) from a table of `table_bytes` -- the ceiling that bounds that field's kernels once the table is
 * Infinity-Cache resident (bench.py `hash_grid_variant.roofline`).  n_threads (multiple of 256) threads x iters gathers. */
int mnrf_bench_gather(const void* table, int64_t table_bytes, int bytes_per_gather, int64_t n_threads, int iters, float* out,
                      void* stream);

/* ---- ray-fused fine pass (eval, per-ray maps only; round 3) ------------------------------------------------------------
 * mnrf_field_forward (all four heads, split arithmetic) + mnrf_composite in ONE launch for rays of exactly
 * mnrf_fused_samples_per_ray() (= 192 = 64 coarse + 128 importance) samples: a workgroup evaluates one ray, keeps the head
 * outputs in LDS and composites them with the very body of mnrf_composite's kernel -- the maps are identical bit for bit, and
 * no per-sample tensor (36 B per sample written and read back otherwise) touches HBM.  What the reference's eval caller
 * copies to the CPU and nobody reads (eval.py:735-736, SURVEY 3 "result-dict contract") is simply never produced.
 * noise_std = 0 (test_time), no density-gradient normal.  Null map pointers are skipped; weights (n_rays, 192) optional.
 * Returns MNRF_ERR_UNSUPPORTED when the 48-samples-per-wave tuning is off (MNRF_SPLIT48=0 / MNRF_SPLIT32=1). */
int mnrf_fused_samples_per_ray(void);
int mnrf_field_composite_fused(float* packed, int64_t n_rays, const float* rays, const float* z_vals,
                               const float* dir_emb, int64_t dir_stride, int white_back,
                               float* weights, float* opacity, float* rgb_map, float* depth, float* mirror_mask,
                               float* surf_normal, float* x_surface, void* stream);

/* ---- training, round 3: operand planes (split arithmetic only) --------------------------------------------------------
 * The weight gradients dW = dY^T X contract over samples.  With MNRF_TRAIN_PLANES the training forward keeps the inputs X
 * of every Linear -- and mnrf_field_backward_planes the pre-activation gradients dY -- not as fp32 rows but as the hi/lo
 * f16 operand tiles the GEMM consumes directly (layout: mirror_nerf_amd/csrc/mnrf_dwp.h), and ONE call of mnrf_dw_planes
 * computes the gradients of all 32 parameters over ALL evaluations of a module in a backward pass (primary rays, reflected
 * rays ...: train.py:253-259 evaluates the same models at every recursion level).
 * Autograd equivalent: loss.backward() through models/mirror_nerf.py:101-212, .grad accumulated over the evaluations. */
#define MNRF_TRAIN_PLANES 256u     /* mnrf_field_forward_train: `save_x` is a planes buffer of mnrf_train_planes_bytes(B) bytes */
#define MNRF_PLANES_Y_HALF 0x100000u   /* round 6, OPT-IN (flags of mnrf_field_backward_planes; bit 12 = 0x1000 of the evaluation's kinds[e]
                                          entry of mnrf_dw_planes2): dY travels as ONE f16 per element under the planes' per-sample
                                          scale instead of a hi/lo pair -- the producer's lo tiles never reach memory, the GEMM fetches
                                          and multiplies the hi tiles only (3/4 of the GEMM's bytes, half the producer's stores).
                                          NOT exact: 1.2e-4 .. 7.4e-4 of each weight tensor's largest entry against float64 on the
                                          gradient fixtures (profiles/r06_half_planes_emulation.json; bar 1e-3).  Ring GEMM only. */
int64_t mnrf_train_planes_bytes(int64_t B);      /* X planes of B samples (bytes) */
int64_t mnrf_train_dy_planes_bytes(int64_t B);   /* dY planes of B samples (bytes) */

/* Activation gradients only (no weight gradients): mnrf_field_backward's first half on the split arithmetic.  dy_planes
 * (mnrf_train_dy_planes_bytes(B)) receives dY under one power-of-two scale for the whole call, derived from the largest seed
 * magnitude, whose float bits are left in *seedmax (a device word, overwritten) for mnrf_dw_planes.  A null upstream
 * gradient means zero (no tensor of zeros needed for a head no loss reads). */
int mnrf_field_backward_planes(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                               const float* rays, const float* z_vals, int spr,
                               const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                               const float* g_is_mirror, const float* rgb, const float* pred_normal,
                               const float* is_mirror, const uint64_t* save_mask, const float* save_inv,
                               void* dy_planes, uint32_t* seedmax, float* d_xyz, float* d_dir, const float* keep_mirror,
                               unsigned flags /* MNRF_CUT_NORMAL_HEAD | MNRF_CUT_MIRROR_HEAD | r << 16 with r = 0..15: every sample's
                                                 largest seed is scaled to [2^(6-r), 2^(7-r)) instead of [2^6, 2^7) -- 2^r more room
                                                 for gradients that GROW on their way down the trunk (trained weights), r bits
                                                 less for those that shrink; the same r goes to mnrf_dw_planes2 in kinds[e] */,
                               void* stream);

/* Weight gradients of n_eval (1..8) evaluations of one module: x_planes[e] from mnrf_field_forward_train, dy_planes[e] and
 * seedmax[e] from mnrf_field_backward_planes, B[e] their sample counts (HOST arrays).  d_params: HOST array of MNRF_N_PARAMS
 * device pointers, overwritten (accumulate = 0) or added to.  workspace: mnrf_dw_planes_workspace_floats(n_eval, B) floats. */
int64_t mnrf_dw_planes_workspace_floats(int n_eval, const int64_t* B);
int mnrf_dw_planes(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                   const uint32_t* const* seedmax, float* workspace, float* const* d_params, int accumulate, void* stream);

/* Round 4: the second-order term (mnrf_field_backward2) on the planes route.  mnrf_field_backward2_planes runs the tangent pass
 * only and leaves the tangents a' (x2_planes, mnrf_train_planes2_bytes(B)) and the density-gradient signals b (y2_planes,
 * mnrf_train_dy_planes2_bytes(B)) as operand planes under one power-of-two scale for the call, derived from the largest |J^|
 * whose float bits are left in *jmax (a device word, overwritten); ADDS to d_xyz when non-null.  mnrf_dw_planes2 is
 * mnrf_dw_planes with a KIND per evaluation (HOST array, null = all 0; bits 8-11: the r of that evaluation's
 * mnrf_field_backward_planes call): 0 = (x_planes, dy_planes, seedmax) of the first-order
 * calls above, 1 = (x2_planes, y2_planes, jmax) of this one -- the weight gradients of a module over all evaluations and both
 * orders in ONE launch (the reference: one loss.backward() through utils/func.py:10-25 with create_graph=True). */
int64_t mnrf_train_planes2_bytes(int64_t B);
int64_t mnrf_train_dy_planes2_bytes(int64_t B);
int mnrf_field_backward2_planes(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                const float* rays, const float* z_vals, int spr, const float* g_normal,
                                const float* normal, const float* save_invj, const uint64_t* save_mask,
                                void* x2_planes, void* y2_planes, uint32_t* jmax, float* d_xyz, void* stream);
int64_t mnrf_dw_planes2_workspace_floats(int n_eval, const int64_t* B, const int* kinds);
int mnrf_dw_planes2(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                    const uint32_t* const* seedmax, const int* kinds, float* workspace, float* const* d_params, int accumulate,
                    void* stream);

/* ---- optimizer step (training.FlatAdam) ------------------------------------------------------------------------------------
 * Adam over ONE flat tensor of n floats (16-byte aligned; a field model's 595 k parameters as views of one buffer, its gradient
 * the flat buffer the backward pass produced): torch.optim.Adam's update (non-amsgrad, L2 weight decay; utils/__init__.py
 * get_optimizer builds that optimizer, train.py:101-109) with one thread per four elements -- torch's fused multi-tensor kernel
 * gives such a tensor ten thread blocks.  `step` = the caller's count of calls (this one included), `*skipped` (device, the
 * caller zeroes it once) = how many of them found_inf has voided: with *found_inf != 0 nothing is updated and *skipped grows by
 * one, as torch's fused Adam does under GradScaler; grad_scale / found_inf may be null. */
int mnrf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int64_t step, int32_t* skipped, const float* grad_scale,
                   const float* found_inf, void* stream);

/* ---- live row counts on the device (round 5): a training step without a host round trip -------------------------------------
 * Training traces the reflections of exactly the rays the mirror mask selects (train.py:170-178, 248-252).  How many those are
 * is known on the device only (*count of mnrf_reflect_compact); the reference reads it on the host (`mask.any()`, then boolean
 * indexing: one stream sync in the middle of every step).  Every `_n` entry point below is its namesake with ONE more argument
 * in front of `stream`: `n_live`, a DEVICE int32 holding the number of rows (rays; samples / spr for the field entry points)
 * that exist.  The row-count argument becomes the CAPACITY the buffers are sized for: the launch is sized for it, workgroups
 * past the live rows leave at once, rows past them are neither read nor written.  n_live == null: exactly the namesake.  With
 * them the forward, backward and optimizer launches of a whole step are a fixed sequence -- capturable as ONE hipGraph
 * (mirror_nerf_amd.training.GraphedTrainStep).  Field entry points: MNRF_SPLIT_F16 | MNRF_TRAIN_PLANES route only. */
int mnrf_embed_n(const float* x, int64_t n, int c, int n_freqs, float* out, const int32_t* n_live, void* stream);
int mnrf_embed_backward_n(const float* x, const float* g_out, int64_t n, int c, int n_freqs, float* g_x, const int32_t* n_live, void* stream);
int mnrf_sample_coarse_n(const float* rays, int64_t n_rays, const float* z_steps, int n_samples, int use_disp, float perturb,
                         const float* perturb_rand, float* z_vals, const int32_t* n_live, void* stream);
/* What render_rays does with a ray before the first field evaluation (models/rendering.py:275-300: `embedding_dir(rays_d)`, the
 * coarse depths) as one launch: dir_emb (n_rays, 3 (2 n_freqs_dir + 1)) = mnrf_embed of columns 3..5 of the rays read in place,
 * z_vals = mnrf_sample_coarse; both bit for bit.  n_samples >= 3. */
int mnrf_ray_prologue_n(const float* rays, int64_t n_rays, int n_freqs_dir, const float* z_steps, int n_samples, int use_disp,
                        float perturb, const float* perturb_rand, float* dir_emb, float* z_vals, const int32_t* n_live, void* stream);
/* The gradient of reflected rays that went into several consumers (train.py:205: the reflected rays carry gradient back to the
 * surface) in one launch instead of autograd's pairwise adds:
 * g_rays = ((((g0 + g1) + g2) + g3) + pad(mnrf_embed_backward(rays[:, 3:6], g_dir_a + g_dir_b))), sums in this order; g0..g3
 * (n_rays, 8) and g_dir_a / g_dir_b (n_rays, 3 (2 n_freqs_dir + 1)) may each be null (at least one is not). */
int mnrf_ray_fan_backward_n(const float* g0, const float* g1, const float* g2, const float* g3, const float* rays,
                            const float* g_dir_a, const float* g_dir_b, int64_t n_rays, int n_freqs_dir, float* g_rays,
                            const int32_t* n_live, void* stream);
int mnrf_composite_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                     const float* noise, const float* rgb, const float* is_mirror,
                     const float* pred_normal, const float* normal, int white_back,
                     float* weights, float* opacity, float* rgb_map, float* depth, float* mirror_mask,
                     float* surf_normal, float* surf_normal_grad, float* normal_dif, float* x_surface,
                     const int32_t* n_live, void* stream);
/* mnrf_composite_n and the mnrf_sample_fine_n that follows it in a coarse pass (models/rendering.py:181-264, then 312-326) as one
 * launch: z_fine (n_rays, S + n_importance) from these weights; u / u_per_ray / n_importance as in mnrf_sample_fine.  weights must
 * not be null.  Same maps, weights and depths bit for bit. */
int mnrf_composite_sample_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                            const float* noise, const float* rgb, const float* is_mirror,
                            const float* pred_normal, const float* normal, int white_back,
                            float* weights, float* opacity, float* rgb_map, float* depth, float* mirror_mask,
                            float* surf_normal, float* surf_normal_grad, float* normal_dif, float* x_surface,
                            const float* u, int u_per_ray, int n_importance, float* z_fine,
                            const int32_t* n_live, void* stream);
int mnrf_composite_backward_n(const float* rays, int64_t n_rays, int S, const float* sigma, const float* z_vals,
                              const float* noise, const float* rgb, const float* is_mirror,
                              const float* pred_normal, const float* normal, int white_back,
                              const float* weights, const float* depth,
                              const float* g_weights, const float* g_opacity, const float* g_rgb_map,
                              const float* g_depth, const float* g_mirror_mask, const float* g_surf_normal,
                              const float* g_surf_normal_grad, const float* g_normal_dif, const float* g_x_surface,
                              float* d_sigma, float* d_rgb, float* d_is_mirror, float* d_pred_normal,
                              float* d_normal, float* d_rays, int detach, const float* keep_mirror,
                              const int32_t* n_live, void* stream);
int mnrf_sample_fine_n(const float* z_coarse, const float* weights, int64_t n_rays, int S,
                       const float* u, int u_per_ray, int n_importance, float* z_fine, const int32_t* n_live, void* stream);
int mnrf_threshold_mask_n(float* mask, int64_t n, int32_t* any, const int32_t* n_live, void* stream);
/* n_live: the live rows of the INPUT rays (a second bounce: the first bounce's count); *count as in the namesake */
int mnrf_reflect_compact_n(const float* rays, const float* x_surface, const float* normal,
                           const float* normal_noise, float noise_std, const float* mask,
                           int64_t n_rays, int compact, float near2, float* sec_rays, int32_t* index,
                           int32_t* count, float* reflect_dir, const int32_t* n_live,
                           int32_t* slot /* (n_rays) or null: the inverse of `index` -- the row of sec_rays ray i went to, -1 where it
                                            was not selected: what the gather-form blend below reads */,
                           void* stream);
/* mnrf_reflect_backward in gather form through `slot` (the static route): every live ray's row of the three outputs is written -- zeros
 * where the ray was not reflected -- so nothing is zero-filled in front; same values.  n_live: the live rows of the INPUT rays. */
int mnrf_reflect_backward_gather_n(const float* rays, const float* normal, const int32_t* slot, const float* g_sec, int64_t n_rays,
                                   float* g_x_surface, float* g_normal, float* g_rays, const int32_t* n_live, void* stream);
/* Both blends of a recursion level (rgb_coarse and rgb_fine, train.py:263-296) in one launch, gather form through `slot`:
 * out = m * part + (1 - m) * base, part = sec[slot[i]] where ray i was reflected and base[i] where not -- mnrf_blend_scatter's
 * expressions, bit-identical values.  Tensor b (or a) may be null.  Backward: g_base = (1 - m) g_out for every live row, g_sec[slot[i]]
 * = m g_out where slot[i] >= 0 (rows of g_sec past the reflection's count are not written). */
int mnrf_blend2_n(const float* base_a, const float* sec_a, const float* base_b, const float* sec_b, const int32_t* slot,
                  const float* mask, int64_t n, int c, float* out_a, float* out_b, const int32_t* n_live, void* stream);
int mnrf_blend2_backward_n(const float* g_out_a, const float* g_out_b, const int32_t* slot, const float* mask, int64_t n, int c,
                           float* g_base_a, float* g_sec_a, float* g_base_b, float* g_sec_b, const int32_t* n_live, void* stream);
/* n_sec_live: live rows of sec / index / g_sec (the reflection's count); n_live: live rows of base / mask / out */
int mnrf_blend_scatter_n(const float* base, const float* sec, const int32_t* index, int64_t n_sec,
                         const float* mask, int64_t n, int c, float* out, float* reflect_out,
                         const int32_t* n_sec_live, const int32_t* n_live, void* stream);
int mnrf_reflect_backward_n(const float* rays, const float* normal, const int32_t* index, int64_t n_sec,
                            const float* g_sec, int64_t n_rays, float* g_x_surface, float* g_normal, float* g_rays,
                            const int32_t* n_sec_live, void* stream);
int mnrf_blend_backward_n(const float* g_out, const float* mask, const int32_t* index, int64_t n_sec, int64_t n, int c,
                          float* g_base, float* g_sec, const int32_t* n_sec_live, const int32_t* n_live, void* stream);
int mnrf_ray_grads_n(const float* d_xyz, const float* z_vals, const float* d_dir, int64_t n_rays, int spr, float* g_rays,
                     float* g_de, const int32_t* n_live, void* stream);
int mnrf_field_forward_train_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                               const float* rays, const float* z_vals, int spr, const float* dir_emb,
                               int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                               float* is_mirror, float* normal, float* save_x, uint64_t* save_mask,
                               float* save_inv, float* save_invj, unsigned flags, const int32_t* n_live, void* stream);
int mnrf_field_backward_planes_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                 const float* rays, const float* z_vals, int spr,
                                 const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                                 const float* g_is_mirror, const float* rgb, const float* pred_normal,
                                 const float* is_mirror, const uint64_t* save_mask, const float* save_inv,
                                 void* dy_planes, uint32_t* seedmax, float* d_xyz, float* d_dir, const float* keep_mirror,
                                 unsigned flags, const int32_t* n_live, void* stream);
int mnrf_field_backward2_planes_n(float* packed, int64_t B, const float* xyz, int64_t xyz_stride,
                                  const float* rays, const float* z_vals, int spr, const float* g_normal,
                                  const float* normal, const float* save_invj, const uint64_t* save_mask,
                                  void* x2_planes, void* y2_planes, uint32_t* jmax, float* d_xyz,
                                  const int32_t* n_live, void* stream);
/* mnrf_dw_planes2 with the sample count of evaluation e on the device: *n_live[e] * spr[e] samples (n_live[e] null: B[e]); B[e]
 * = the capacity its planes were sized for.  n_live, spr: HOST arrays of n_eval entries.  The work plan of the GEMM (which
 * workgroup contracts which 32-sample stages) is made by a one-workgroup launch in front of it and lives at the head of the
 * workspace: mnrf_dw_planes2_n_workspace_floats(n_eval) floats, whatever the counts. */
int64_t mnrf_dw_planes2_n_workspace_floats(int n_eval);
int mnrf_dw_planes2_n(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
                      const int32_t* const* n_live, const int* spr, const uint32_t* const* seedmax, const int* kinds,
                      float* workspace, float* const* d_params, int accumulate, void* stream);
/* mnrf_adam_step with every scalar of the step in DEVICE memory (a captured step must not freeze the learning rate or the step
 * count).  mnrf_adam_prep, one thread: *step += 1 (the count of steps, this one included), then state[0..7] = [veto, lr / bias
 * correction 1, sqrt(bias correction 2), beta1, beta2, 1 / grad_scale, eps, weight_decay] from hyper = [lr, beta1, beta2, eps,
 * weight_decay] (doubles; lr, eps and weight_decay are rounded to float as mnrf_adam_step's arguments are) and *skipped (the
 * first model's counter of vetoed steps); veto = *found_inf != 0 or any of the 0..4 range-guard words (HOST array of device words:
 * the last word of a packed image) non-zero.  mnrf_adam_step_dev: the update of one flat tensor from `state`. */
int mnrf_adam_prep(const double* hyper, int64_t* step, const int32_t* skipped, const float* grad_scale, const float* found_inf,
                   const uint32_t* const* guard_words, int n_guard_words, float* state, void* stream);
int mnrf_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* state,
                       int32_t* skipped, void* stream);
/* mnrf_adam_step_dev for up to 4 flat tensors in ONE launch (a step's coarse and fine model): arrays of n_tensors entries each. */
int mnrf_adam_step_dev_n(int n_tensors, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                         const int64_t* n, const float* state, int32_t* const* skipped, void* stream);

/* ---- hash-grid field, BASELINE config 5 (models/mirror_nerf_tcnn.py:151-259) ---------------------
 * table: (entries, 2) fp32 hash-grid features; offsets17_host: 17 level offsets in entries (HOST);
 * log2_per_level_scale, base_resolution, bound: the encoding configuration (mirror_nerf_tcnn.py:36-49);
 * weights: mnrf_tcnn_weight_floats() floats = [sigma_net.0 (64x32) | sigma_net.1 (16x64) | color_net.0
 * (64x32, col 31 zero) | color_net.1 (64x64) | color_net.2 (3x64) | normal_net.0 (64x16, col 15 zero) |
 * normal_net.1 (3x64) | is_mirror_net.0 (32x16, col 15 zero) | its bias (32) | is_mirror_net.2 (1x32) |
 * its bias (1)], rows padded to a multiple of 4.
 * Positions/directions: `xyz` rows [x y z dx dy dz] (stride >= 6, or >= 3 with SIGMA_ONLY), or rays+z_vals
 * with per-ray raw directions `dirs` (null: the ray direction).  sigma is the RAW output h[0] (the ReLU is
 * applied at compositing, mirror_nerf_tcnn.py:235).  geo_feat: (B,15).  Parity vs tinycudann: unpinned. */
int mnrf_tcnn_weight_floats(void);
/* The blob from the model's 11 MLP parameter tensors (HOST array of device pointers in the order above: contiguous fp32 in
 * nn.Linear layout, models/mirror_nerf_tcnn.py:51-149) in one launch; padded columns and the tail are written as zeros. */
int mnrf_tcnn_pack_weights(const float* const* params, float* weights, void* stream);
int mnrf_tcnn_forward(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                      int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                      const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                      const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                      float* is_mirror, float* normal, float* geo_feat, float* enc_workspace, void* stream);
/* enc_workspace: null, or 32 * B floats of the caller's memory.  With it the full / f16 evaluations on the matrix pipe run as
 * TWO launches: a level-major encoding pass (grid y = level: the whole device gathers from one 4 MB level at a time, so the
 * XCDs' L2s hold it) that writes the 16 x float2 encoding of every sample into the workspace, and the MLP kernel reading it.
 * Without it every wave walks all 16 levels itself (one launch; 3.2x the table bytes over the fabric at 32768-ray chunks). */

/* Backward of mnrf_tcnn_forward (training of config 5: autograd through mirror_nerf_tcnn.py:220-259,
 * gridencoder.cu:275-380 kernel_grid_backward, shencoder.cu:81-160).  Same inputs as the forward (full evaluation);
 * g_*: dL/d(sigma (B), rgb (B,3), pred_normal (B,3), is_mirror (B)), any may be null (= zero).  ACCUMULATES into
 * d_table (entries,2) and d_weights (mnrf_tcnn_weight_floats() floats, the layout of `weights`; padded columns
 * receive zeros): the caller zero-initialises both, and `workspace` (mnrf_tcnn_backward_workspace_floats() floats:
 * private per-XCD copies of the coarse levels' gradient, folded into d_table at the end; null = none).  Optional outputs d_xyz (B,3) = dL/d position, d_dir (B,3) =
 * dL/d raw direction.  Nothing is saved by the forward: the kernel re-evaluates each tile.
 * g_normal (B,3) or null: dL/d(`normal` of the forward, the normalised density gradient).  When given, a second kernel
 * adds the SECOND-ORDER term -- the gradient that reaches the table, sigma_net and the position through
 * normal = l2n(-d sigma/dx), i.e. what autograd.grad(sigma, x, create_graph=True) propagates in
 * models/mirror_nerf_tcnn.py:172-218 / utils/func.py:10-25 -- to d_table, d_weights and d_xyz. */
int64_t mnrf_tcnn_backward_workspace_floats(const int64_t* offsets17_host);
int64_t mnrf_tcnn_backward_workspace_floats2(const int64_t* offsets17_host, unsigned flags /* MNRF_TCNN_GRAD_F16 or 0 */);
/* With MNRF_TCNN_GRAD_F16 the LAST FOUR floats of that workspace are an overflow word (uint32, first of the four): non-zero after
 * the call when a half2 sum left the f16 range (it was clamped to +-65504 / scale, not inf) -- fall back to fp32 atomics then. */
int64_t mnrf_tcnn_backward_workspace_floats3(const int64_t* offsets17_host, unsigned flags, int64_t B);
int mnrf_tcnn_backward(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                       int base_resolution, float bound, const float* weights, int64_t B, const float* xyz,
                       int64_t xyz_stride, const float* rays, const float* z_vals, int spr, const float* dirs,
                       int64_t dir_stride, const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                       const float* g_is_mirror, const float* g_normal, float* workspace, float* d_table,
                       float* d_weights, float* d_xyz, float* d_dir,
                       const float* keep_mirror /* (B/spr) per ray [(B) with xyz] or null: 0 = the mirror head of this ray's samples
                                                   sees geo_feat.detach() (models/mirror_nerf_tcnn.py:200-215) */,
                       unsigned flags /* MNRF_CUT_NORMAL_HEAD / MNRF_CUT_MIRROR_HEAD: that head sees geo_feat.detach()
                                         (mirror_nerf_tcnn.py:186-190, 196-199); MNRF_TCNN_GRAD_F16: see there */,
                       void* stream);

/* Round 6 -- live row counts for the hash-grid field (config 5 on the static training route, see "live row counts on the device"
 * below): ray mode only; B = the capacity the buffers (and the [level][B] planes) are sized for, the kernels evaluate / differentiate
 * the first *n_live * spr samples; rows past them are neither read nor written and add nothing to any gradient.  n_live null =
 * mnrf_tcnn_forward / mnrf_tcnn_backward. */
int mnrf_tcnn_forward_n(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                        int base_resolution, float bound, const float* weights, unsigned flags, int64_t B,
                        const float* xyz, int64_t xyz_stride, const float* rays, const float* z_vals, int spr,
                        const float* dirs, int64_t dir_stride, float* sigma, float* rgb, float* pred_normal,
                        float* is_mirror, float* normal, float* geo_feat, float* enc_workspace, const int32_t* n_live, void* stream);
int mnrf_tcnn_backward_n(const float* table, const int64_t* offsets17_host, double log2_per_level_scale,
                         int base_resolution, float bound, const float* weights, int64_t B, const float* xyz,
                         int64_t xyz_stride, const float* rays, const float* z_vals, int spr, const float* dirs,
                         int64_t dir_stride, const float* g_sigma, const float* g_rgb, const float* g_pred_normal,
                         const float* g_is_mirror, const float* g_normal, float* workspace, float* d_table,
                         float* d_weights, float* d_xyz, float* d_dir, const float* keep_mirror, unsigned flags,
                         const int32_t* n_live, void* stream);

/* Pin-hole ray generation on device (datasets/ray_utils.py:6-53): rays (H*W, 8). */
int mnrf_generate_rays(int H, int W, float focal, const float* c2w_host12, float near, float far,
                       float* rays, void* stream);

/* ---- loss reductions of the training step (losses.py:7-255: ColorLoss, MirrorMaskLoss, PlaneConsistentLoss,
 * NormalLoss, NormalRegLoss, TotalLoss), value and gradient in one pass.  Index [0] = coarse, [1] = fine; a null
 * input pointer = "key absent from the result dict" (losses.py tests `f"rgb_{typ}" in inputs`).  Every g_* buffer
 * receives d(total)/d(input) (same shape as the input; may be null).  `out` (6 floats): color, mirror_mask, plane,
 * normal, normal_reg (each already multiplied by its coefficient; 0 when the term is switched off), total.
 * In the train_geometry_stage / invalid-GT branch of ColorLoss the reference thresholds the predicted mask IN PLACE
 * (losses.py:27-33, a detach() shares storage); so does this entry point: mirror_mask[] is not const. */
#define MNRF_LOSS_GEOMETRY_STAGE 1u              /* train_geometry_stage */
#define MNRF_LOSS_WO_MASK_RGB_TO_BLACK 2u        /* hparams.woMaskRGBtoBlack */
#define MNRF_LOSS_NORMAL_ONLY_INSIDE_MIRROR 4u   /* hparams.normal_loss_only_inside_mirror */
#define MNRF_LOSS_EXT_GRAD_NORMAL 8u             /* NormalRegLoss.ext_supervise_grad_normal (default on) */
#define MNRF_LOSS_TCNN_BCE 16u                   /* model_type == "nerf_tcnn": utils/func.py:32-37 instead of nn.BCELoss */
#define MNRF_LOSS_USE_MASK 32u                   /* epoch gates of TotalLoss.forward (losses.py:233-249) */
#define MNRF_LOSS_USE_PLANE 64u
#define MNRF_LOSS_USE_NORMAL 128u
#define MNRF_LOSS_PLANE_ON_DEVICE 256u           /* PlaneConsistentLoss with no host read (round 6): the quadruples are drawn by the
                                                    kernel as floor(plane_u * M) from M = the number of GT-mirror rows the count launch
                                                    of the same call leaves in the workspace; M // 4 of the plane_cap quadruples are
                                                    live (none when a GT entry is invalid, losses.py:116-119); plane_idx / plane_times
                                                    are not read */
typedef struct {
    const float* rgb[2];          /* (N,3) */
    float* mirror_mask[2];        /* (N)   */
    const float* normal_dif[2];   /* (N)   */
    const float* pred_normal[2];  /* (N,S,3) */
    const float* weights[2];      /* (N,S) */
    const float* x_surface[2];    /* (N,3) */
    const float* normal_fine;     /* (N,S_fine,3) */
    int n_samples[2];
    const float* targets;         /* batch["rgbs"] (N,3) */
    const float* gt_mask;         /* batch["mirror_mask"] (N) as float, < 0 = invalid; null = absent */
    const float* rays;            /* batch["rays"] (N,8): directions in columns 3..5 */
    const unsigned char* valid_mask;   /* batch["valid_mask"] (N) or null */
    int64_t n_rays;
    const int64_t* plane_idx[2];  /* (times,4) draws of torch.randint, rows of the GT-mirror subset */
    int64_t plane_times[2];
    const float* plane_u[2];      /* MNRF_LOSS_PLANE_ON_DEVICE: (plane_cap,4) uniform numbers in [0,1); plane_cap >= n_rays / 4 */
    int64_t plane_cap[2];
    float w_color, w_normal, w_normal_reg, w_mask, w_plane;
    unsigned flags;
    float* g_rgb[2];
    float* g_mirror_mask[2];
    float* g_normal_dif[2];
    float* g_pred_normal[2];
    float* g_weights[2];
    float* g_x_surface[2];
    float* g_normal_fine;
    float* out;
} MnrfLossArgs;
int64_t mnrf_loss_workspace_floats(int64_t n_rays, int n_samples_coarse, int n_samples_fine, int64_t plane_times);
int mnrf_total_loss(const MnrfLossArgs* args, float* workspace, void* stream);
/* counts only: workspace[0] = #rays with gt < 0, workspace[1] = #rays with gt != 0 (what the plane loss draws from) */
int mnrf_loss_count(const MnrfLossArgs* args, float* workspace, void* stream);

/* metrics.py:5-15 on device: out[0] = mse = mean((pred - gt)^2) over the n elements whose mask byte is non-zero
 * (mask may be null; element i uses mask[i / per_mask], so per_mask = 3 selects whole RGB pixels), out[1] = psnr =
 * -10 log10(mse), out[2] = number of elements.  `partials`: 2 * mnrf_mse_blocks() floats.  Deterministic. */
int mnrf_mse_blocks(void);
int mnrf_mse_psnr(const float* pred, const float* gt, const unsigned char* mask, int64_t n, int per_mask,
                  float* partials, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNRF_H */
