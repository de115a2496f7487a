// mnrf_field_split32.hip -- the 32x32x16 tuning of the forward-only split-f16 field kernels and the packer of its
// weight stream (mnrf_field_split32.inc for the kernel, mnrf_layout.h "32x32x16 tuning" for the layout).
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

// h32: 32 samples per wave as ONE MFMA column block, one wave per SIMD; 16 KiB chunks, 64 KiB ring of four slots
namespace h32 {
constexpr int CHUNK_PAIRS = 8;
constexpr int RING_SLOTS = 4;
#include "mnrf_field_split32.inc"
}  // namespace h32

// MNRF_SPLIT32=1 (read once): forward-only split launches take this tuning, and mnrf_pack_weights also builds its stream
bool split32_enabled() {
    static const bool v = [] { const char* e = getenv("MNRF_SPLIT32"); return e && atoi(e) != 0; }();
    return v;
}

int launch_split32(const FieldArgs& A, bool sigma_only, hipStream_t s) { return h32::launch(A, sigma_only, s); }

// ------------------------------------------------------------------ packer of the 32x32x16 forward stream
// One thread per f16.  Segment list = the part sequence of mnrf_layout.h (trunk parts in two halves of four 32-row blocks,
// L5's encoding and hidden parts interleaved by half); pair p of a segment is (T, nb) = (p / rows, p % rows); half e of
// lane l (m = l & 31, h = l >> 5) of that pair is W[32*nb + m][col], col by the part's kind:
//   hidden   16*T + 8*(e>>2) + 4*h + (e&3)      (the accumulator registers of the producing layer, in order)
//   xyz enc  enc_col32(T, h, e)                 ((sin, cos) pairs P = 16h + 4T + (e>>1))
//   view enc 16*T + 8*h + e, zero from 27 on
struct Seg32 { signed char part, half; };
struct Pack32Args { const float* params[32]; };

__global__ void split32_pack_kernel(Pack32Args P, float* packed, PartTable T) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)SPLIT32_FWD_PAIRS * (PAIR_BYTES / 2)) return;
    const int pair = (int)(q / (PAIR_BYTES / 2));
    const int within = (int)(q % (PAIR_BYTES / 2));
    const bool is_lo = within >= 512;
    const int lane = (within & 511) >> 3, e = within & 7, m = lane & 31, h = lane >> 5;
    const signed char seg_part[27] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 5, 4, 5, 6, 6, 7, 7, 8, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};
    const signed char seg_half[27] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 0, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int k = 0, half = -1, pair0 = 0, rows = 0, nt = 0;
    bool found = false;
    for (int sg = 0; sg < 27; ++sg) {
        k = seg_part[sg];
        half = seg_half[sg];
        const Part& pt = T.fwd[k];
        nt = pt.ntq;                                                // k-steps of 16 columns
        rows = half >= 0 ? 4 : (pt.n_true + 31) / 32;               // 32-row blocks of this segment
        const int np = nt * rows;
        if (pair < pair0 + np) { found = true; break; }
        pair0 += np;
    }
    float w = 0.f;
    if (found) {
        const Part pt = T.fwd[k];
        const int lp = pair - pair0;
        const int Tq = lp / rows, nb = (half >= 0 ? 4 * half : 0) + lp % rows;
        const int n = 32 * nb + m;
        int col;
        if (pt.kind == KIND_ENC) col = enc_col32(Tq, h, e);
        else if (pt.kind == KIND_DIR) { col = 16 * Tq + 8 * h + e; if (col >= ENC_DIR) col = -1; }
        else col = 16 * Tq + 8 * (e >> 2) + 4 * h + (e & 3);
        if (n < pt.n_true && col >= 0) w = P.params[pt.param][(long long)n * pt.ld + pt.col_off + col];
    }
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    _Float16* dst = (_Float16*)(packed + OFF_SPLIT32_FWD);
    dst[q] = is_lo ? lo : hi;     // (out-of-range weights are flagged by split_pack_kernel on the same values)
}

void launch_split32_pack(const float* const* params, float* packed, hipStream_t s) {
    PartTable T;
    build_parts(T);
    Pack32Args P;
    for (int i = 0; i < 32; ++i) P.params[i] = params[i];
    const long long n = (long long)SPLIT32_FWD_PAIRS * (PAIR_BYTES / 2);
    hipLaunchKernelGGL(split32_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P, packed, T);
}

}  // namespace mnrf
