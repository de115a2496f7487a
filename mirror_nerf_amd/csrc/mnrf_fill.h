// mnrf_fill.h -- zero fill as a plain kernel launch.  hipMemsetAsync costs ~38 us of idle GPU per call inside the training
// step (profiles/r03h: nine calls per step, 0.34 ms of idle in front of them); a one-line kernel costs a launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace mnrf {
struct ZeroJobs {
    uint32_t* p[4];
    long long n[4];      // 32-bit words
};
static __global__ void zero_words_kernel(ZeroJobs J) {
    uint32_t* p = J.p[blockIdx.y];
    const long long n = J.n[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
// up to four buffers (sizes in bytes, multiples of 4) in one launch
static inline void zero_fill(hipStream_t s, void* p0, size_t b0, void* p1 = nullptr, size_t b1 = 0, void* p2 = nullptr, size_t b2 = 0,
                             void* p3 = nullptr, size_t b3 = 0) {
    ZeroJobs J{{(uint32_t*)p0, (uint32_t*)p1, (uint32_t*)p2, (uint32_t*)p3}, {(long long)(b0 / 4), (long long)(b1 / 4), (long long)(b2 / 4), (long long)(b3 / 4)}};
    int jobs = 0;
    long long mx = 0;
    for (int i = 0; i < 4; ++i) {
        if (!J.p[i]) J.n[i] = 0;
        if (J.n[i] > 0) jobs = i + 1;
        mx = J.n[i] > mx ? J.n[i] : mx;
    }
    if (!jobs) return;
    long long blocks = (mx + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks, jobs), dim3(256), 0, s, J);
}
}  // namespace mnrf
