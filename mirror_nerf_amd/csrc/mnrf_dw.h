// mnrf_dw.h -- weight-gradient driver (mnrf_dw.hip), called from the backward C-ABI entry point.
#pragma once
#include <hip/hip_runtime.h>
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
}  // namespace mnrf
