// mnrf_field_split.hip -- split-f16 tunings of the fused field kernel and the packer of their
// weight streams (see mnrf_field_split.inc for the arithmetic, mnrf_layout.h for the layout).
// Replaces the same reference code as mnrf_field.hip: models/mirror_nerf.py:101-212, 20-38,
// models/rendering.py:302, 134-179.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>

#include "mnrf_layout.h"
#include "mnrf_field_args.h"

namespace mnrf {

extern __shared__ __attribute__((aligned(16))) char smem[];

// h2: 32 samples per wave, one wave per SIMD; 16 KiB chunks, 64 KiB ring of four slots
namespace h2 {
constexpr int S = 2;
constexpr int MIN_WAVES_PER_SIMD = 1;
constexpr int CHUNK_PAIRS = 8;
constexpr int RING_SLOTS = 4;
#include "mnrf_field_split.inc"
}  // namespace h2
// h2x: 32 KiB chunks (half the seams), 96 KiB ring of three slots
namespace h2x {
constexpr int S = 2;
constexpr int MIN_WAVES_PER_SIMD = 1;
constexpr int CHUNK_PAIRS = 16;
constexpr int RING_SLOTS = 3;
#include "mnrf_field_split.inc"
#include "mnrf_field_split_bwd.inc"
}  // namespace h2x
#ifdef MNRF_EXP_H1
// experiment: 16 samples per wave, <= 256 registers, two workgroups per CU (two waves share each SIMD's matrix pipe)
namespace h1 {
constexpr int S = 1;
constexpr int MIN_WAVES_PER_SIMD = 2;
constexpr int CHUNK_PAIRS = 8;
constexpr int RING_SLOTS = 3;
#define MNRF_SPLIT_NO_GRAD
#include "mnrf_field_split.inc"
#undef MNRF_SPLIT_NO_GRAD
}  // namespace h1
#endif

int launch_split(const FieldArgs& A, bool sigma_only, bool grad, int variant, hipStream_t s) {
#ifdef MNRF_EXP_H1
    if (variant == 3 && !grad) return h1::launch(A, sigma_only, grad, s);
#endif
    // variant 0: measured default -- 16 KiB chunks for the forward-only kernels (17.2 vs 17.3 ms per 6.29 M full
    // samples), 32 KiB chunks when the density-gradient pass is on (32.1 vs 34.1 ms); 1 / 2 force h2 / h2x.
    // A 16-samples-per-wave tuning with two workgroups per CU (as s1 of the fp32 kernel) was tried and dropped:
    // 18.2 / 32.5 ms -- it hides the waits but issues twice the LDS-DMA per sample.  So was a 48-samples-per-wave tuning
    // (a third less LDS-DMA per sample): it needs all 512 registers, spills 200 bytes and ends up 1 % slower.
    // MNRF_SPLIT32=1: the forward-only launches on the 32x32x16 tuning (mnrf_field_split32.inc)
    if (!grad && variant == 0 && split32_enabled()) return launch_split32(A, sigma_only, s);
    // default of the forward-only launches: 48 samples per wave (mnrf_field_split3.hip); geo_feat needs both halves of L8
    // in registers, which that tuning cannot afford
    if (!grad && variant == 0 && !A.geo_feat && split48_enabled()) return launch_split48(A, sigma_only, s);
    const bool big = variant == 0 ? grad : variant == 2;
    return big ? h2x::launch(A, sigma_only, grad, s) : h2::launch(A, sigma_only, grad, s);
}

int launch_split_bwd(const FieldBwdArgs& A, hipStream_t s) { return h2x::launch_bwd(A, s); }
int launch_split_bwd2(const FieldBwd2Args& A, hipStream_t s) { return h2x::launch_bwd2(A, s); }

// ------------------------------------------------------------------ split-stream packer
// One thread per LANE of a pair (round 4; it was one thread per f16: the part search below ran sixteen times as often and the
// kernel took 30 us per model and step).  Half j of lane l of pair (T, nb) of a part is float (j&3) of lane l of the part's
// fp32 tile (2T + (j>>2), nb); hi = f16(w) (round to nearest), lo = f16(w - hi): the thread reads two float4 and writes the 8 hi
// halves (16 B at pair * 2048 + 16 l) and the 8 lo halves (1 KiB further).
struct SplitPackImages {
    float* packed[4];
};
__global__ void split_pack_kernel(SplitPackImages I, PartTable T) {
    float* packed = I.packed[blockIdx.y];
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr long long FWD_LANES = (long long)SPLIT_FWD_PAIRS * 64;
    constexpr long long BWD_LANES = (long long)SPLIT_BWD_PAIRS * 64;
    constexpr long long HBWD_LANES = (long long)SPLIT_HBWD_PAIRS * 64;
    if (q >= FWD_LANES + BWD_LANES + HBWD_LANES) return;
    const int region = q >= FWD_LANES + BWD_LANES ? 2 : (q >= FWD_LANES ? 1 : 0);     // forward, trunk^T, heads^T
    const bool bwd = region != 0;
    const long long r = q - (region == 2 ? FWD_LANES + BWD_LANES : (region == 1 ? FWD_LANES : 0));
    const int pair = (int)(r >> 6);
    const int lane = (int)(r & 63);
    // Segments of the forward stream: the nine trunk parts (L1, L2-4, L5 encoding, L5 hidden, L6-8) are stored in two
    // HALVES of 8 row blocks each (mnrf_field_split.inc: the kernel evaluates a trunk layer half by half), L5's two
    // parts interleaved by half; the head parts and the whole backward stream are stored part by part.
    const Part* parts = region == 2 ? T.hbwd : (region == 1 ? T.bwd : T.fwd);
    int k = 0, half = -1, pair0 = 0;
    if (bwd) {
        const int nparts = region == 2 ? N_HBWD_PARTS - 1 : N_BWD_PARTS;
        for (;;) {
            const int np = (parts[k].ntq + 1) / 2 * parts[k].nb;      // parts follow each other without padding
            if (k + 1 >= nparts || pair < pair0 + np) break;
            pair0 += np;
            ++k;
        }
    } else {
        const signed char seg_part[27] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 5, 4, 5, 6, 6, 7, 7, 8, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};
        const signed char seg_half[27] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 0, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
        for (int sg = 0;; ++sg) {
            k = seg_part[sg];
            half = seg_half[sg];
            const int np = parts[k].ntq / 2 * (half >= 0 ? 8 : parts[k].nb);
            if (sg + 1 >= 27 || pair < pair0 + np) break;
            pair0 += np;
        }
    }
    const Part pt = parts[k];
    const int lp = pair - pair0;
    const int rows = half >= 0 ? 8 : pt.nb;
    const int Tq = lp / rows, nb = (half >= 0 ? 8 * half : 0) + lp % rows;
    const float* src = packed + (region == 2 ? OFF_HBWD : (region == 1 ? OFF_BWD : OFF_FWD));
    float w[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (2 * Tq + h < pt.ntq)       // (an odd ntq leaves the upper half of its last step zero)
            v = *(const f32x4*)(src + (pt.tile0 + (long long)(2 * Tq + h) * pt.nb + nb) * TILE_FLOATS + lane * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) w[4 * h + c] = v[c];
    }
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    h8v hi, lo;
    bool bad = false;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        hi[jj] = (_Float16)w[jj];
        lo[jj] = (_Float16)(w[jj] - (float)hi[jj]);
        bad |= !(fabsf(w[jj]) < 65504.f);      // inf / nan / out of range
    }
    if (bad) atomicOr((unsigned*)(packed + PACKED_FLOATS - 1), MNRF_GUARD_WEIGHT);
    char* dst = (char*)(packed + (region == 2 ? OFF_SPLIT_HBWD : (region == 1 ? OFF_SPLIT_BWD : OFF_SPLIT_FWD))) + (long long)pair * PAIR_BYTES + lane * 16;
    *(h8v*)dst = hi;
    *(h8v*)(dst + PAIR_BYTES / 2) = lo;
}

void launch_split_pack(float* const* packed, int n_images, hipStream_t s) {      // n_images <= 4
    PartTable T;
    build_parts(T);
    SplitPackImages I{};
    for (int m = 0; m < n_images; ++m) I.packed[m] = packed[m];
    const long long n = (long long)(SPLIT_FWD_PAIRS + SPLIT_BWD_PAIRS + SPLIT_HBWD_PAIRS) * 64;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n + 255) / 256), n_images), dim3(256), 0, s, I, T);
}

}  // namespace mnrf
