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

// Dynamic tile queue of the h3 kernels (FieldArgs::tile_queue).  The {next, done} counter pair of a launch lives in the
// CALLER's packed weight image (mnrf_layout.h: OFF_TILE_QUEUE, TQ_PAIRS pairs before the range-guard word; zeroed by
// mnrf_pack_weights): the library allocates nothing and keeps no device state.  Every launch leaves its pair at zero again (the
// last workgroup out resets it); consecutive launches on an image rotate through the TQ_PAIRS pairs, so up to TQ_PAIRS launches
// that share an image may be in flight on different streams (launches on one stream are ordered anyway).
// MNRF_TILE_QUEUE=0 (read once) keeps the static one-workgroup-per-tile grid.
namespace {
constexpr int TQ_DEVICES = 64;
int g_resident[TQ_DEVICES];          // host-side cache of the CU count per device (0 = not asked yet)
unsigned g_next_pair = 0;
bool tile_queue_enabled() {
    static const bool v = [] { const char* e = getenv("MNRF_TILE_QUEUE"); return !(e && atoi(e) == 0); }();
    return v;
}
}  // namespace

static bool tile_queue_slot(FieldArgs& A) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TQ_DEVICES) return false;
    int resident = __atomic_load_n(&g_resident[dev], __ATOMIC_RELAXED);
    if (!resident) {
        // 150 KB of LDS and 512 registers per lane: one workgroup per CU
        if (hipDeviceGetAttribute(&resident, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || resident <= 0) return false;
        __atomic_store_n(&g_resident[dev], resident, __ATOMIC_RELAXED);
    }
    const long long tiles = (A.B + h3::WG_SAMPLES - 1) / h3::WG_SAMPLES;
    if (tiles <= resident || tiles > 0x7fffffff) return false;      // everything is resident at once: nothing to deal
    A.tile_queue = (int*)(const_cast<float*>(A.packed) + OFF_TILE_QUEUE) + 2 * (__atomic_fetch_add(&g_next_pair, 1u, __ATOMIC_RELAXED) % TQ_PAIRS);
    A.n_tiles = (int)tiles;
    A.resident = resident;
    return true;
}

int launch_split48(const FieldArgs& A0, bool sigma_only, hipStream_t s) {
    FieldArgs A = A0;
    A.tile_queue = nullptr;
    if (tile_queue_enabled() && tile_queue_slot(A)) {
        // the pair is zeroed on the launch stream and not only by the last workgroup of the previous launch that used it: a kernel
        // that aborted would otherwise leave it non-zero and the next launch on that pair would skip or repeat tiles
        if (hipMemsetAsync(A.tile_queue, 0, 2 * sizeof(int), s) != hipSuccess) A.tile_queue = nullptr;
    }
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
