// mnrf_dw.h -- weight-gradient driver (mnrf_dw.hip), called from the backward C-ABI entry point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace mnrf {
int dw_splits(long long B);
long long dw_workspace_floats(long long B);
// save_x: activations of the training forward; dY: pre-activation gradients of the backward kernel;
// ws: dw_workspace_floats(B) floats; d_params: 32 device pointers in state_dict order (overwritten, or added to
// when `accumulate`).
int launch_dw(const float* save_x, const float* dY, const float* g_sigma, long long B, float* ws, float* const* d_params,
              int accumulate, hipStream_t s);
long long dw2_workspace_floats(long long B);
// second-order pass: so = tangents and density-gradient signals of field_bwd2_kernel; ADDS to d_params
int launch_dw2(const float* so, long long B, float* ws, float* const* d_params, hipStream_t s);
// ---- round 3: weight gradients from operand planes (mnrf_dwp.hip / mnrf_dwp.h)
// kinds (null = all 0): per evaluation 0 = first-order planes, 1 = second-order planes (mnrf_dwp.h)
long long dwp_workspace_floats(int n_eval, const int64_t* B, const int* kinds);
int launch_dwp(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
               const unsigned* const* seedmax, const int* kinds, float* ws, float* const* d_params, int accumulate, hipStream_t s);
// ... with the sample counts of the evaluations on the device (round 5): B = capacities; evaluation e has *n_live[e] * spr[e]
// samples (n_live[e] null: B[e]).  Workspace: dwp_workspace_floats_n(n_eval) floats (sized for any counts up to the capacities).
long long dwp_workspace_floats_n(int n_eval);
int launch_dwp_n(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B, const int32_t* const* n_live,
                 const int* spr, const unsigned* const* seedmax, const int* kinds, float* ws, float* const* d_params, int accumulate,
                 hipStream_t s);
// largest |J^| of a second-order pass (float bits) -> *out, as field_split_bwd2_kernel's prologue forms it
// pair: a {value, done} word pair that is zero between launches (mnrf_layout.h OFF_REDUCE_PAIR of the model's packed image): the
// reduction is then ONE launch -- the last workgroup out stores the maximum to *out and resets the pair; null: *out is zeroed by
// a launch of its own first
void launch_jhat_max(const float* g_normal, const float* normal, const float* save_invj, long long B, unsigned* out, hipStream_t s,
                     const int* n_live = nullptr, int spr = 1, unsigned* pair = nullptr);
// largest seed magnitude of an evaluation (float bits) -> *out; the seeds are those of field_split_bwd_kernel's prologue
void launch_seed_max(const float* g_sigma, const float* g_rgb, const float* g_pn, const float* g_m, const float* rgb,
                     const float* pn, const float* is_mirror, const float* save_inv, long long B, unsigned* out, hipStream_t s,
                     const int* n_live = nullptr, int spr = 1, unsigned* pair = nullptr);
}  // namespace mnrf
