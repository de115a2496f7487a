// mnrf_field_split3.hip -- 48-samples-per-wave tuning ("h3") of the forward-only split-f16 field kernels: the body of
// mnrf_field_split.inc with S = 3 groups of 16 samples per wave (192 samples per workgroup).  Why: with the weight stream or
// the A-operand LDS reads compiled out the S = 2 tuning draws 60-150 W less, leaves the power limit and runs at the full
// 2.4 GHz (profiles/r02j_energy.txt) -- moving the weights costs about 45 % of the kernel's energy, and it scales with
// 1 / (samples per wave).  S = 3 moves a third less per sample; it fits the 512 registers of a lane only because the
// xyz-encoding operands are parked in LDS between L1 and L5, the lo operands hl[s][0..5] of the heads' input are parked
// there while the heads run (fetched back one k-step ahead), and the A operands are read one unit ahead instead of two.
// Replaces the same reference code as mnrf_field.hip: models/mirror_nerf.py:101-212, 20-38, models/rendering.py:302, 134-179.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>

#include "mnrf_layout.h"
#include "mnrf_field_args.h"

namespace mnrf {

extern __shared__ __attribute__((aligned(16))) char smem[];

namespace h3 {
constexpr int S = 3;
constexpr int MIN_WAVES_PER_SIMD = 1;
constexpr int CHUNK_PAIRS = 8;
constexpr int RING_SLOTS = 4;
#define MNRF_SPLIT_NO_GRAD
#include "mnrf_field_split.inc"
#undef MNRF_SPLIT_NO_GRAD
}  // namespace h3

// Default for the forward-only split launches (measured 2.5-3.5 % faster than S = 2 on both kernels); MNRF_SPLIT48=0 (read
// once) keeps them on the 32-samples-per-wave tuning
bool split48_enabled() {
    static const bool v = [] { const char* e = getenv("MNRF_SPLIT48"); return !(e && atoi(e) == 0); }();
    return v;
}

int launch_split48(const FieldArgs& A, bool sigma_only, hipStream_t s) {
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)h3::field_split_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h3::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)h3::field_split_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h3::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)h3::field_split_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   h3::LDS_BYTES_FUSE);
        return true;
    }();
    (void)once;
    return h3::launch(A, sigma_only, false, s);
}

int split48_ray_samples() { return h3::WG_SAMPLES; }

}  // namespace mnrf
