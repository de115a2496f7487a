// exp_f16split.hip -- two questions asked of the hardware before the split-f16 field kernel was written:
//  (1) does v_mfma_f32_16x16x32_f16 honour f16 subnormal A/B inputs (or flush them)?
//  (2) what does the hot loop "2 x ds_read_b128 (hi, lo tile) + 3*NS MFMAs" sustain per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/exp_f16split.hip -o gpurun_out/exp_f16split
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void denorm_kernel(float* out, float av, float bv) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
    a[0] = (_Float16)av;     // lane l: A[row l&15][k = 8*(l>>4)]
    b[0] = (_Float16)bv;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}

template <int NS, int NB>
__global__ __launch_bounds__(256, 1) void loop_kernel(float* out, const float* in, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 4; i += 256) ((float*)lds)[i] = in[i & 1023];
    __syncthreads();
    h8 b_hi[NS][8], b_lo[NS][8];
    for (int s = 0; s < NS; ++s)
        for (int t = 0; t < 8; ++t)
            for (int j = 0; j < 8; ++j) {
                b_hi[s][t][j] = (_Float16)in[(tid + s + t + j) & 1023];
                b_lo[s][t][j] = (_Float16)(in[(tid + 2 * s + t + j) & 1023] * 0.001f);
            }
    f4 acc[NS][NB];
    for (int s = 0; s < NS; ++s)
        for (int nb = 0; nb < NB; ++nb) acc[s][nb] = f4{0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int tile = (t * NB + nb) & 31;
                const h8 a_hi = *(const h8*)(lds + tile * 2048 + lane * 16);
                const h8 a_lo = *(const h8*)(lds + tile * 2048 + 1024 + lane * 16);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi[s][t], acc[s][nb], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi[s][t], acc[s][nb], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo[s][t], acc[s][nb], 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int s = 0; s < NS; ++s)
        for (int nb = 0; nb < NB; ++nb) r += acc[s][nb][0] + acc[s][nb][1] + acc[s][nb][2] + acc[s][nb][3];
    out[blockIdx.x * 256 + tid] = r;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NS, int NB>
void run_loop(const char* name, float* d_out, float* d_in, long long* d_cyc) {
    const int iters = 200, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<NS, NB>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, 10, d_cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<NS, NB>), dim3(blocks), dim3(256), 0, 0, d_out, d_in, iters, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double mfmas = (double)iters * 8 * NB * 3 * NS;
    const double flop = mfmas * 16384.0 * 4 * blocks;
    printf("%s: NS=%d NB=%d  %.1f cycles/MFMA (s_memtime ticks, 100 MHz => x clk/100MHz)  %.1f TFLOP/s executed  %.3f ms\n", name, NS, NB,
           (double)cyc / mfmas, flop / ms * 1e-9, ms);
}

int main() {
    float* d_out; float* d_in; long long* d_cyc;
    hipMalloc(&d_out, 256 * 4096 * 4); hipMalloc(&d_in, 4096 * 4); hipMalloc(&d_cyc, 64);
    float h_in[1024];
    for (int i = 0; i < 1024; ++i) h_in[i] = (float)((i * 37) % 101 - 50) / 64.f;
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    float o[64];
    const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {ldexpf(1.f, -15), 1.f},
                              {ldexpf(1.f, -20), ldexpf(1.f, -20)}, {1.f, 1.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d_out, c[0], c[1]);
        hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        printf("denorm: a=%g b=%g -> mfma %g (exact %g)\n", c[0], c[1], o[0], (double)c[0] * c[1]);
    }
    run_loop<2, 16>("loop", d_out, d_in, d_cyc);
    run_loop<1, 16>("loop", d_out, d_in, d_cyc);
    run_loop<2, 8>("loop", d_out, d_in, d_cyc);
    run_loop<3, 16>("loop", d_out, d_in, d_cyc);
    return 0;
}
