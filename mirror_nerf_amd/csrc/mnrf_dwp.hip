// mnrf_dwp.hip -- weight gradients of the field MLP from operand planes (see mnrf_dwp.h for the layout and the reasons).
//
//   dwp_gemm_kernel    persistent workgroups (one per CU, 8 waves): each owns a contiguous span of the cost line of
//                      mnrf_dwp.h, i.e. consecutive 32-sample stages of one or a few (job, evaluation) pairs; per stage the
//                      operand tiles of the job (up to 32 KiB of dY + 32 KiB of X, hi and lo planes) travel HBM -> LDS by
//                      LDS-DMA into a double buffer, MFMA operands come out of LDS through ds_read_b64_tr_b16, three
//                      v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block (hi.hi + hi.lo + lo.hi) into a 256 x 256 fp32 tile
//                      held in registers (wave (wn, wk): rows 64 wn .., columns 128 wk ..); one partial tile per span piece.
//   dwp_finish_kernel  sums the partial tiles of every Linear over workgroups and evaluations (each evaluation with its own
//                      power-of-two scale) in a fixed order and writes -- or adds to -- the gradients in nn.Linear layout.
// All GEMMs of all evaluations of a module in a backward pass are ONE launch (rounds 1-2: 5 launches per evaluation).
// Autograd equivalent: the .grad accumulation of loss.backward() for models/mirror_nerf.py:59-99.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mnrf.h"
#include "mnrf_error.h"
#include "mnrf_fill.h"
#include "mnrf_layout.h"
#include "mnrf_dwp.h"
#include "mnrf_dw.h"

namespace mnrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

extern __shared__ __attribute__((aligned(16))) char dwp_smem[];

constexpr int DWP_BUF = 64 * 1024;         // one stage: [A tiles: 32 KiB][X tiles: 32 KiB]
constexpr int DWP_XOFF = 32 * 1024;
constexpr int DWP_LDS = 2 * DWP_BUF;

struct DwpEval {
    const char* X;            // X planes of the evaluation  [n_sb][PLX_FB][2][1 KiB]
    const char* Y;            // dY planes                   [n_sb][PLY_FB][2][1 KiB]
    const unsigned* seedmax;  // bits of the largest |seed| of the evaluation (float), see mnrf_dwp.h "scale"
    int y_half;               // round 6, opt-in (kinds bit 12 / MNRF_DW_PLANES_HALF): the producer stored the hi tiles of dY only -- the lo
                              // half-tiles are neither fetched (lanes 32-63 of a dY piece stay out of the LDS-DMA) nor multiplied
};
struct DwpArgs {
    DwpEval ev[DWP_MAX_EVAL];
    DwpPlan plan;
    float* part;              // [G + DWP_JOBS * n_eval slots][DWP_SLOT_FLOATS]
    const DwpDevPlan* dev;    // DEV kernels: the plan comes from here (dwp_plan_kernel), `plan` is unused
};

__device__ __forceinline__ f32x4 mfma_h(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// one MFMA operand = feature (lane & 15) of a tile, k slots 8 gg .. 8 gg + 7 = rows {4 gg .. 4 gg + 3} and {16 + 4 gg ..}:
// two transposing reads with lane-linear addresses (tile + 8 lane, tile + 512 + 8 lane)
__device__ __forceinline__ u32x4 read_operand(const char* tile, int lane8) {
    typedef __attribute__((address_space(3))) h4 lds_h4;
    const h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(tile + lane8));
    const h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(tile + 512 + lane8));
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    return u32x4{l.x, l.y, h.x, h.y};
}

// tile t (1 KiB, lane-linear) of a contiguous run of tiles: global -> LDS by LDS-DMA
__device__ __forceinline__ void dma_tile(const char* src, char* dst, int lane16) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane16),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, PL_LOAD_AUX);
}

// MB: dY blocks per wave row (wn), KB: X blocks per wave column (wk); NA / NX: blocks of the job's operands
template <int MB, int KB, int NA, int NX>
__device__ __forceinline__ void dwp_segment_run(const char* __restrict__ Yb, const char* __restrict__ Xb, long long ysb, long long xsb,
                                                int s_lo, int s_hi, bool bias, float* __restrict__ slot) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wk = wave >> 2;
    const int lane8 = lane * 8, lane16 = lane * 16;
    const bool has_a = wn * MB < NA;         // NA = 1: only wn = 0 owns a row block
    constexpr int NT = 2 * NA + 2 * NX;      // 1-KiB tiles per stage
    char* const lds = dwp_smem;

    f32x4 acc[MB][KB], bacc[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const u32x4 ones = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};      // f16 1.0 in every k slot

    // the wave's share of a stage's tiles: t = wave, wave + 8, ...  (A tiles first, then X tiles at DWP_XOFF)
    auto issue = [&](int s, int buf) {
        const char* ya = Yb + (long long)s * ysb;
        const char* xa = Xb + (long long)s * xsb;
        char* base = lds + buf * DWP_BUF;
#pragma unroll
        for (int q = 0; q < (NT + 7) / 8; ++q) {
            const int t = wave + 8 * q;
            if (t < 2 * NA) dma_tile(ya + t * PL_TILE_BYTES, base + t * PL_TILE_BYTES, lane16);
            else if (t < NT) dma_tile(xa + (t - 2 * NA) * PL_TILE_BYTES, base + DWP_XOFF + (t - 2 * NA) * PL_TILE_BYTES, lane16);
        }
    };

    __syncthreads();                 // the previous segment's last stage has been read by every wave
    issue(s_lo, 0);
    for (int s = s_lo; s < s_hi; ++s) {
        const int buf = (s - s_lo) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's tiles of stage s have landed ...
        __syncthreads();                                      // ... and everybody's; buffer buf ^ 1 is free (stage s - 1 consumed)
        if (s + 1 < s_hi) issue(s + 1, buf ^ 1);
        const char* A = lds + buf * DWP_BUF;
        const char* X = A + DWP_XOFF;
        if (has_a) {
            u32x4 ah[MB], al[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah[mb] = read_operand(A + (wn * MB + mb) * PL_FB_BYTES, lane8);
                al[mb] = read_operand(A + (wn * MB + mb) * PL_FB_BYTES + PL_TILE_BYTES, lane8);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const u32x4 bh = read_operand(X + (wk * KB + kb) * PL_FB_BYTES, lane8);
                const u32x4 bl = read_operand(X + (wk * KB + kb) * PL_FB_BYTES + PL_TILE_BYTES, lane8);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    f32x4 c = acc[mb][kb];
                    c = mfma_h(al[mb], bh, c);      // lo . hi
                    c = mfma_h(ah[mb], bl, c);      // hi . lo
                    c = mfma_h(ah[mb], bh, c);      // hi . hi
                    acc[mb][kb] = c;
                }
            }
            if (bias && wk == 0) {                  // column sums of dY: the same operands against a row of ones
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    bacc[mb] = mfma_h(al[mb], ones, bacc[mb]);
                    bacc[mb] = mfma_h(ah[mb], ones, bacc[mb]);
                }
            }
        }
    }
    // partial tile, fragment order: block (nb, kb) = 1 KiB, lane-linear float4 (row 4 g + r, column lane & 15 of the block)
    if (has_a) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                ((f32x4*)slot)[((wn * MB + mb) * NX + (wk * KB + kb)) * 64 + lane] = acc[mb][kb];
        if (bias && wk == 0 && (lane & 15) == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) ((f32x4*)(slot + 256 * 256))[(wn * MB + mb) * 4 + (lane >> 4)] = bacc[mb];
        }
    }
}

// ---- the same segment through a RING of half-stages (round 4; the default).  What capped the two-buffer version above was not
// the memory system: a streaming reader with the same instruction gets 6.06 TB/s (default policy) / 6.87 TB/s (nt) out of these
// buffers (mnrf_bench_stream, scripts/bw_probe.py) where that loop got 4.5 -- and 4.67 with its LDS reads and MFMAs compiled out.
// It refills a whole stage at a time: nothing new is requested until the LAST byte of a stage has landed and every wave has
// passed the barrier, so the bytes in flight swing between 0 and one stage (64 KiB for the big jobs, 18 KiB for the small ones)
// and the memory pipe of the CU runs dry once per stage.  Here a stage travels as two HALF-STAGES (rows 0-15 / 16-31 of every
// tile: one LDS-DMA instruction brings the hi and the lo half-tile of a feature block, lanes 0-31 / 32-63, two 512-byte runs in
// memory, 1 KiB lane-linear in LDS) through a ring of D slots of NP KiB; the stage being multiplied occupies two slots (its two
// halves are read by ONE transposing operand read each and multiplied with the K = 32 MFMA, as above -- a first build of this
// ring multiplied half-stages with the K = 16 MFMA, which runs at half the rate and then bound the loop), D - 2 half-stages are
// on their way, and two more are requested as soon as a stage has been consumed: the bytes in flight swing between D - 4 and
// D - 2 half-stages.  D = 5 for the 16 x 16-block jobs (160 KiB of LDS), up to 12 for the small ones.  Every wave issues the
// same number Q of instructions per half-stage (a wave without work re-loads a pair: identical bytes), so "the two halves of
// stage st have landed" is a counted s_waitcnt (VMEM operations complete in order).
constexpr int DWP_RING_LDS = 160 * 1024;

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 24, "add the immediate");
#define MNRF_WAIT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MNRF_WAIT_CASE(1) MNRF_WAIT_CASE(2) MNRF_WAIT_CASE(3) MNRF_WAIT_CASE(4) MNRF_WAIT_CASE(5) MNRF_WAIT_CASE(6) MNRF_WAIT_CASE(7)
    MNRF_WAIT_CASE(8) MNRF_WAIT_CASE(9) MNRF_WAIT_CASE(10) MNRF_WAIT_CASE(11) MNRF_WAIT_CASE(12) MNRF_WAIT_CASE(13) MNRF_WAIT_CASE(14)
    MNRF_WAIT_CASE(15) MNRF_WAIT_CASE(16) MNRF_WAIT_CASE(17) MNRF_WAIT_CASE(18) MNRF_WAIT_CASE(19) MNRF_WAIT_CASE(20) MNRF_WAIT_CASE(21)
    MNRF_WAIT_CASE(22) MNRF_WAIT_CASE(23) MNRF_WAIT_CASE(24)
#undef MNRF_WAIT_CASE
}
// at most R half-stages (of Q instructions each) may still be on their way: R is a run-time number in [0, RMAX]
template <int Q, int RMAX> __device__ __forceinline__ void wait_half_stages(int r) {
    if (r >= RMAX) wait_vm<RMAX * Q>();
    else if constexpr (RMAX > 0) wait_half_stages<Q, RMAX - 1>(r);
}

// operand of the K = 32 MFMA from the two half-tiles of a feature block that sit in two ring slots: rows 0-15 | rows 16-31
__device__ __forceinline__ u32x4 read_operand2(const char* half0, const char* half1, int lane8) {
    typedef __attribute__((address_space(3))) h4 lds_h4;
    const h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(half0 + lane8));
    const h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(half1 + lane8));
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    return u32x4{l.x, l.y, h.x, h.y};
}

template <int MB, int KB, int NA, int NX>
__device__ __forceinline__ void dwp_segment_run_h(const char* __restrict__ Yb, const char* __restrict__ Xb, long long ysb, long long xsb,
                                                  int s_lo, int s_hi, bool bias, float* __restrict__ slot, bool y_half) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wk = wave >> 2;
    const int lane8 = lane * 8;
    const bool has_a = wn * MB < NA;
    const bool skip_lo = y_half && lane >= 32;   // dY pieces: the lo half-tile (lanes 32-63 of the instruction) does not exist
    constexpr int NP = NA + NX;                  // feature blocks = (hi, lo) half-tile pairs per half-stage = KiB per ring slot
    constexpr int Q = (NP + 7) / 8;              // LDS-DMA instructions per wave and half-stage
    constexpr int SLOT = NP * 1024;
    constexpr int D = DWP_RING_LDS / SLOT < 12 ? DWP_RING_LDS / SLOT : 12;      // ring slots
    static_assert(D >= 5 && (D - 4) * Q <= 24, "ring depth / wait immediates");
    char* const lds = dwp_smem;
    const int lsrc = (lane >> 5) * PL_TILE_BYTES + (lane & 31) * 16;      // hi tile (lanes 0-31) / lo tile (32-63) of the block

    f32x4 acc[MB][KB], bacc[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const u32x4 ones = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};

    const int n_hs = 2 * (s_hi - s_lo);
    auto issue = [&](int hs, int ring_slot) {      // half-stage hs = rows 16 (hs & 1) .. of sample block s_lo + hs / 2
        const int s = s_lo + (hs >> 1);
        const char* ya = Yb + (long long)s * ysb + (hs & 1) * 512 + lsrc;
        const char* xa = Xb + (long long)s * xsb + (hs & 1) * 512 + lsrc;
        char* base = lds + ring_slot * SLOT;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            int pr = wave + 8 * q;
            // no work left for this wave: an identical re-load keeps the counts uniform.  (Skipping it instead -- such a wave then
            // counts Q - 1 requests per half-stage -- removes 10 % of a training step's requests and measured SLOWER: 1.30-1.38
            // against 1.24-1.25 ms, alternating on one box.)
            if (pr >= NP) pr -= NP;
            const char* src = pr < NA ? ya + pr * PL_FB_BYTES : xa + (pr - NA) * PL_FB_BYTES;
            char* dst = base + pr * 1024;        // [dY half-tile pairs: NA KiB][X half-tile pairs: NX KiB]
            // (y_half: the instruction is issued by every wave all the same -- the counted waits count instructions -- with half
            //  its lanes switched off: 512 bytes instead of 1 KiB)
            if (!(skip_lo && pr < NA))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, PL_LOAD_AUX);
        }
    };

    auto issue_piece = [&](int hs, int ring_slot, int q) {      // instruction q of half-stage hs alone
        const int s = s_lo + (hs >> 1);
        int pr = wave + 8 * q;
        if (pr >= NP) pr -= NP;
        const char* src = pr < NA ? Yb + (long long)s * ysb + (hs & 1) * 512 + lsrc + pr * PL_FB_BYTES
                                  : Xb + (long long)s * xsb + (hs & 1) * 512 + lsrc + (pr - NA) * PL_FB_BYTES;
        if (!(skip_lo && pr < NA))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + ring_slot * SLOT + pr * 1024), 16, 0, PL_LOAD_AUX);
    };
    // PACED (default since the end of round 4): the 2 Q requests that refill a stage's slots are spread over the KB column blocks
    // of the stage's MFMAs instead of leaving as one burst of 8 waves x 2 Q instructions right after the barrier, and the X
    // operands are read one column block ahead (below).  Together 1.23 against 1.28-1.36 ms for the two evaluations of a training
    // step -- the time of this loop with its LDS reads and MFMAs compiled out (MNRF_EXP_DWP_NOMATH): the math is hidden now.
#if defined(MNRF_EXP_DWP_BURST) || defined(MNRF_EXP_DWP_NOMATH)
    constexpr bool PACED = false;
#else
    constexpr bool PACED = true;
#endif

    __syncthreads();                 // the previous segment's last stage has been read by every wave
    int next_hs = 0, next_slot = 0;  // next half-stage to request and the slot it goes to (hs mod D)
#pragma unroll 1
    for (; next_hs < D - 2 && next_hs < n_hs; ++next_hs) { issue(next_hs, next_slot); next_slot = next_slot + 1 == D ? 0 : next_slot + 1; }
    int cur = 0;                     // ring slot of half-stage 2 st
#pragma unroll 1
    for (int st = 0; 2 * st < n_hs; ++st) {
        // half-stages 2 st and 2 st + 1 must have landed; the ones requested after them may still be on their way
        wait_half_stages<Q, D - 4>(next_hs - (2 * st + 2));
        // a RAW barrier: __syncthreads() waits for vmcnt(0) first and would drain the ring.  After it everybody's tiles of stage
        // st are there and stage st - 1 has been read (its LDS reads were consumed by its MFMAs): its two slots are free
        __builtin_amdgcn_s_barrier();
        // ... for half-stages hs0, hs0 + 1: request idx of 2 Q = instruction idx % Q of half-stage hs0 + idx / Q
        const int hs0 = next_hs, slot0 = next_slot, slot1 = next_slot + 1 == D ? 0 : next_slot + 1;
        const auto request = [&](int idx) {
            const int k = idx / Q, q = idx - k * Q;
            if (hs0 + k < n_hs) issue_piece(hs0 + k, k == 0 ? slot0 : slot1, q);
        };
#pragma unroll 1
        for (int k = 0; k < 2 && next_hs < n_hs; ++k, ++next_hs) next_slot = next_slot + 1 == D ? 0 : next_slot + 1;
        if (!PACED || !has_a) {      // (a wave without a row block has no MFMAs to spread them over)
#pragma unroll
            for (int idx = 0; idx < 2 * Q; ++idx) request(idx);
        }
        const int cur1 = cur + 1 == D ? 0 : cur + 1;
        const char* A0 = lds + cur * SLOT;
        const char* A1 = lds + cur1 * SLOT;
        const char* X0 = A0 + NA * 1024;
        const char* X1 = A1 + NA * 1024;
#ifdef MNRF_EXP_DWP_NOMATH      // experiment (no sums): the LDS-DMA stream, counted waits and barriers alone -- what the plane reads cost
        if (false) {
#else
        if (has_a) {
#endif
            u32x4 ah[MB], al[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah[mb] = read_operand2(A0 + (wn * MB + mb) * 1024, A1 + (wn * MB + mb) * 1024, lane8);
                al[mb] = read_operand2(A0 + (wn * MB + mb) * 1024 + 512, A1 + (wn * MB + mb) * 1024 + 512, lane8);
                if (y_half) al[mb] = u32x4{0u, 0u, 0u, 0u};      // (what the LDS holds there is whatever an earlier job left)
            }
            // the X operands one column block ahead of the MFMAs that consume them, requested after the first row block's MFMAs
            // of the block before: their LDS round trip runs under the other 3 (MB - 1) MFMAs.  (The compiler's own schedule read
            // two blocks, waited for all of them, multiplied; and it waits with lgkmcnt(0) for these transposing reads, so the
            // request must not be the last thing before the wait: hence the scheduling barriers.)
            u32x4 bhq[2], blq[2];
            bhq[0] = read_operand2(X0 + (wk * KB) * 1024, X1 + (wk * KB) * 1024, lane8);
            blq[0] = read_operand2(X0 + (wk * KB) * 1024 + 512, X1 + (wk * KB) * 1024 + 512, lane8);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if constexpr (PACED) {
#pragma unroll
                    for (int idx = kb * 2 * Q / KB; idx < (kb + 1) * 2 * Q / KB; ++idx) request(idx);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const u32x4 bh = bhq[kb & 1], bl = blq[kb & 1];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    f32x4 c = acc[mb][kb];
                    c = mfma_h(al[mb], bh, c);      // lo . hi
                    c = mfma_h(ah[mb], bl, c);      // hi . lo
                    c = mfma_h(ah[mb], bh, c);      // hi . hi
                    acc[mb][kb] = c;
                    if (mb == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kb + 1 < KB) {
                            bhq[(kb + 1) & 1] = read_operand2(X0 + (wk * KB + kb + 1) * 1024, X1 + (wk * KB + kb + 1) * 1024, lane8);
                            blq[(kb + 1) & 1] = read_operand2(X0 + (wk * KB + kb + 1) * 1024 + 512, X1 + (wk * KB + kb + 1) * 1024 + 512, lane8);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (bias && wk == 0) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    bacc[mb] = mfma_h(al[mb], ones, bacc[mb]);
                    bacc[mb] = mfma_h(ah[mb], ones, bacc[mb]);
                }
            }
        }
        cur = cur1 + 1 == D ? 0 : cur1 + 1;
    }
    if (has_a) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                ((f32x4*)slot)[((wn * MB + mb) * NX + (wk * KB + kb)) * 64 + lane] = acc[mb][kb];
        if (bias && wk == 0 && (lane & 15) == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) ((f32x4*)(slot + 256 * 256))[(wn * MB + mb) * 4 + (lane >> 4)] = bacc[mb];
        }
    }
}

// A 32-bit word of the device-made plan through the SCALAR cache.  The plan is read between the stores of the partial tiles, where
// hipcc cannot prove it unclobbered and falls back to vector loads -- VMEM operations whose s_waitcnt vmcnt(0) sits in front of
// every one of a workgroup's ~34 (job, evaluation) pairs: 35 us per launch in the first build of the DEV kernels (rocprofv3: 636 vs
// 601 us).  The address is wave-uniform by construction (kernel argument + loop counters).
__device__ __forceinline__ int plan_word(const void* base, int byte_off) {
    int v;
    const char* q = (const char*)base + byte_off;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
    return v;
}

// DEV: the plan was made on the device (DwpDevPlan, live row counts); the launch has one workgroup per CU
template <bool HALF, bool DEV = false>
__global__ __launch_bounds__(DWP_WG_THREADS, 1) void dwp_gemm_kernel(DwpArgs A) {
    const int g = blockIdx.x;
    DwpPlan hd;      // n_eval, G, T: what the interval needs (the per-evaluation entries are read where they are used)
    if (DEV) {
        hd.n_eval = plan_word(&A.dev->plan.n_eval, 0);
        hd.G = plan_word(&A.dev->plan.G, 0);
        hd.T = (long long)(((unsigned long long)(unsigned)plan_word(&A.dev->plan.T, 4) << 32) | (unsigned)plan_word(&A.dev->plan.T, 0));
        if (g >= hd.G || hd.T == 0) return;
    } else {
        hd.n_eval = A.plan.n_eval; hd.G = A.plan.G; hd.T = A.plan.T;
    }
    long long P = 0, c0, c1;
    dwp_interval(hd, g, c0, c1);
    for (int j = 0; j < DWP_JOBS; ++j) {
        const int w = dwp_weight(j);
        for (int e = 0; e < hd.n_eval; ++e) {
            const int kind = DEV ? plan_word(&A.dev->plan.kind[0], 4 * e) : A.plan.kind[e];
            const int n_sb = DEV ? plan_word(&A.dev->plan.n_sb[0], 4 * e) : A.plan.n_sb[e];
            const int n = dwp_has(kind, j) ? n_sb : 0;      // dwp_stages()
            int s_lo, s_hi;
            dwp_segment(c0, c1, P, w, n, s_lo, s_hi);
            P += (long long)n * w;
            if (s_hi <= s_lo) continue;
            const DwpJob jb = dwp_job_of(kind, j);
            const long long ysb = dwp_y_stride(kind), xsb = dwp_x_stride(kind);
            const char* Yb = A.ev[e].Y + (long long)jb.ya * PL_FB_BYTES;
            const char* Xb = A.ev[e].X + (long long)jb.xa * PL_FB_BYTES;
            float* slot = A.part + (long long)(g + j * hd.n_eval + e) * DWP_SLOT_FLOATS;
            const bool y_half = kind == 0 && A.ev[e].y_half != 0;      // (the second-order planes stay hi / lo)
            if constexpr (HALF) {
                switch (jb.shape) {
                case 0: dwp_segment_run_h<4, 8, 16, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                case 1: dwp_segment_run_h<4, 2, 16, 4>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                case 2: dwp_segment_run_h<2, 8, 8, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                case 3: dwp_segment_run_h<2, 1, 8, 2>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                case 4: dwp_segment_run_h<1, 8, 1, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                default: dwp_segment_run_h<1, 4, 1, 8>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot, y_half); break;
                }
            } else {
                switch (jb.shape) {
                case 0: dwp_segment_run<4, 8, 16, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                case 1: dwp_segment_run<4, 2, 16, 4>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                case 2: dwp_segment_run<2, 8, 8, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                case 3: dwp_segment_run<2, 1, 8, 2>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                case 4: dwp_segment_run<1, 8, 1, 16>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                default: dwp_segment_run<1, 4, 1, 8>(Yb, Xb, ysb, xsb, s_lo, s_hi, jb.bias, slot); break;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA of this workgroup is still on its way
}

// ---------------------------------------------------------------------------------------------------------- finish
// A parameter is assembled from up to two column ranges, each a window of one job's tile.
struct DwpSource {
    short job;       // -1: none
    short row0;      // first row of the job's tile that belongs to this Linear
    short col0;      // first column of the job's tile
    short kind;      // 0: column c of the Linear = tile column col0 + c;  1: xyz-encoding pair order (encpos)
};
struct DwpLayer {
    DwpSource src[2];    // columns [0, split_col) from src[0], the rest from src[1]
    short split_col;
    short out_f, in_f;
    short bias_job, bias_row0;
    float* d_w;
    float* d_b;
};
struct DwpFinishArgs {
    DwpLayer layer[16];
    const float* part;
    const unsigned* seedmax[DWP_MAX_EVAL];
    int kind[DWP_MAX_EVAL];
    int seed_log2[DWP_MAX_EVAL];      // kind 0: FieldBwdArgs::seed_log2 of the evaluation's backward launch
    short g_lo[DWP_JOBS * DWP_MAX_EVAL], g_hi[DWP_JOBS * DWP_MAX_EVAL];     // owners of virtual job v (g_lo > g_hi: none)
    int n_eval;
    int accumulate;
    short encpos[64];
    const DwpDevPlan* dev;     // non-null: the owners come from the device plan instead of g_lo / g_hi
};

// scale exponent of an evaluation's dY planes: K puts the largest seed magnitude into [2^6, 2^7) (the backward kernel uses
// the same K) plus the boost of mnrf_dwp.h
__device__ __forceinline__ int dwp_scale_log2(const unsigned* seedmax, int kind, int seed_log2) {
    const int e = (int)((*seedmax >> 23) & 0xffu);
    // kind 1: *seedmax holds the largest |J^| of the launch and K2 puts it into [1, 2) (field_split_bwd2_kernel uses the same K2)
    const int K = (e == 0 || e == 255) ? 0 : (kind ? 0 : seed_log2) - (e - 127);
    return K + PL_BOOST_LOG2;
}

// One thread per FOUR rows of one column of a Linear (round 4; it was one thread per element): rows 4 q .. 4 q + 3 of a column
// are one float4 of the partial tile's fragment order, and consecutive columns are consecutive float4 -- the reads of the
// partial tiles (20 per element for the large jobs) are coalesced 16-byte loads instead of 4-byte loads at a 16-byte stride.
// After the weights of a layer come its bias sums, one thread each.
__global__ void dwp_finish_kernel(DwpFinishArgs F) {
    const DwpLayer& ly = F.layer[blockIdx.y];
    const int n4 = (ly.out_f + 3) / 4;
    const int nw4 = n4 * ly.in_f;
    const int el = blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= nw4 + ly.out_f) return;
    const bool is_bias = el >= nw4;
    int job, row, col, n0 = 0, c = 0;
    if (!is_bias) {
        n0 = 4 * (el / ly.in_f);
        c = el % ly.in_f;
        const bool first = c < ly.split_col;
        const DwpSource s = first ? ly.src[0] : ly.src[1];
        job = s.job;
        row = s.row0 + n0;
        int cc = first ? c : c - ly.split_col;
        if (s.kind == 1) cc = F.encpos[cc];
        col = s.col0 + cc;
    } else {
        job = ly.bias_job;
        row = ly.bias_row0 + (el - nw4);
        col = 0;
    }
    if (job < 0) return;
    const DwpJob jb = dwp_job(job);
    // position inside a partial slot (fragment order of dwp_segment_run): float4 = rows 4 (row >> 2) .. + 3 of column col
    long long idx;
    if (!is_bias) idx = ((long long)((row >> 4) * jb.nx + (col >> 4)) * 64 + ((row & 15) >> 2) * 16 + (col & 15)) * 4;
    else idx = 256 * 256 + row;
    f32x4 total = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < F.n_eval; ++e) {
        const int v = job * F.n_eval + e;
        if (is_bias && F.kind[e]) continue;      // the second-order term has no bias gradient (its jobs write no bias sums)
        const int g_lo = F.dev ? F.dev->g_lo[v] : F.g_lo[v], g_hi = F.dev ? F.dev->g_hi[v] : F.g_hi[v];
        if (g_lo > g_hi) continue;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g = g_lo; g <= g_hi; ++g) {
            const float* src = F.part + (long long)(g + v) * DWP_SLOT_FLOATS + idx;
            if (is_bias) sum[0] += *src;
            else sum += *(const f32x4*)src;
        }
        const int k = -dwp_scale_log2(F.seedmax[e], F.kind[e], F.seed_log2[e]);      // (ldexp: K can exceed 126)
#pragma unroll
        for (int r = 0; r < 4; ++r) total[r] += ldexpf(sum[r], k);
    }
    if (is_bias) {
        float* dst = ly.d_b + (el - nw4);
        *dst = F.accumulate ? *dst + total[0] : total[0];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + r < ly.out_f) {
                float* dst = ly.d_w + (long long)(n0 + r) * ly.in_f + c;
                *dst = F.accumulate ? *dst + total[r] : total[r];
            }
    }
}

// Tail of a grid-wide maximum of non-negative floats (ordered like their bit patterns).  The workgroup folds its waves' maxima in
// LDS first (every atomic of the grid goes to the same two words and they serialise at ~10 ns each: one per workgroup, not one per
// wave).  pair == null: atomicMax into *out (zeroed by a launch in front).  Else `pair` = {value, done}, zero between launches: the
// last workgroup out stores the result to *out with a plain store and resets both words -- one launch, nothing to zero.
constexpr int MAX_THREADS = 1024;      // workgroup size of the two maximum kernels: one sample per thread up to 256 K samples
__device__ __forceinline__ void grid_max_tail(float mx, unsigned* __restrict__ out, unsigned* __restrict__ pair) {
    __shared__ float wave_max[MAX_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int waves = (blockDim.x + 63) >> 6;
    for (int w = 1; w < waves; ++w) mx = fmaxf(mx, wave_max[w]);
    if (!pair) {
        if (mx > 0.f) atomicMax(out, __builtin_bit_cast(unsigned, mx));
        return;
    }
    if (mx > 0.f) atomicMax(pair, __builtin_bit_cast(unsigned, mx));
    __threadfence();
    if (atomicAdd(pair + 1, 1u) == gridDim.x - 1) {
        __threadfence();
        *out = atomicExch(pair, 0u);
        atomicExch(pair + 1, 0u);
    }
}

// ---------------------------------------------------------------------------------------------------------- seed maximum
// The seeds (pre-activation gradients of the four output layers) exactly as field_split_bwd_kernel's prologue forms them --
// same expressions, same order, -ffp-contract=off in both translation units -- reduced to the largest magnitude.
__global__ __launch_bounds__(MAX_THREADS) void seed_max_kernel(const float* __restrict__ g_sigma, const float* __restrict__ g_rgb, const float* __restrict__ g_pn,
                                const float* __restrict__ g_m, const float* __restrict__ rgb, const float* __restrict__ pn,
                                const float* __restrict__ is_mirror, const float* __restrict__ save_inv, long long B,
                                unsigned* __restrict__ out, const int* __restrict__ n_live, int spr, unsigned* __restrict__ pair) {
    if (n_live) {      // live row count: B is the capacity
        long long bl = (long long)*n_live * spr;
        bl = bl < 0 ? 0 : bl;
        B = bl < B ? bl : B;
    }
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
        if (g_sigma) mx = fmaxf(mx, fabsf(g_sigma[i]));
        if (g_rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float y = rgb[i * 3 + c];
                mx = fmaxf(mx, fabsf(g_rgb[i * 3 + c] * y * (1.f - y)));
            }
        }
        if (g_pn) {
            const float inv = save_inv[i];
            const float g0 = g_pn[i * 3], g1 = g_pn[i * 3 + 1], g2 = g_pn[i * 3 + 2];
            float s0, s1, s2;
            if (inv > 0.f) {
                const float p0 = pn[i * 3], p1 = pn[i * 3 + 1], p2 = pn[i * 3 + 2];
                const float dt = p0 * g0 + p1 * g1 + p2 * g2;
                s0 = (g0 - p0 * dt) * inv; s1 = (g1 - p1 * dt) * inv; s2 = (g2 - p2 * dt) * inv;
            } else {
                s0 = -g0 * inv; s1 = -g1 * inv; s2 = -g2 * inv;
            }
            mx = fmaxf(mx, fmaxf(fabsf(s0), fmaxf(fabsf(s1), fabsf(s2))));
        }
        if (g_m) {
            const float y = is_mirror[i];
            mx = fmaxf(mx, fabsf(g_m[i] * y * (1.f - y)));
        }
    }
    // non-negative floats order like their bit patterns; NaN (fmaxf drops it) never gets here, inf gives exponent 255 -> K = 0
    grid_max_tail(mx, out, pair);
}

void launch_seed_max(const float* g_sigma, const float* g_rgb, const float* g_pn, const float* g_m, const float* rgb,
                     const float* pn, const float* is_mirror, const float* save_inv, long long B, unsigned* out, hipStream_t s,
                     const int* n_live, int spr, unsigned* pair) {
    if (!pair) zero_fill(s, out, sizeof(unsigned));
    // 1024-thread workgroups, at most 256 of them (one sample per thread up to 256 K samples: the launch is latency-bound, sixteen
    // dependent loads per sample).  Every workgroup ends in two same-address atomics (the maximum and the ticket of grid_max_tail),
    // which serialise at ~10 ns each: 2048 small workgroups measured 31 us per launch against 12.6 with 128 -- and with one atomic per
    // WAVE (as it was through round 4) 14-18 us against 7-9 now.
    long long blocks = (B + MAX_THREADS - 1) / MAX_THREADS;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(seed_max_kernel, dim3((unsigned)blocks), dim3(MAX_THREADS), 0, s, g_sigma, g_rgb, g_pn, g_m, rgb, pn, is_mirror,
                       save_inv, B, out, n_live, spr, pair);
}

// largest |J^| (the seed of the second-order pass) exactly as field_split_bwd2_kernel's prologue forms it
__global__ __launch_bounds__(MAX_THREADS) void jhat_max_kernel(const float* __restrict__ g_normal, const float* __restrict__ normal, const float* __restrict__ save_invj,
                                long long B, unsigned* __restrict__ out, const int* __restrict__ n_live, int spr, unsigned* __restrict__ pair) {
    if (n_live) {      // live row count: B is the capacity
        long long bl = (long long)*n_live * spr;
        bl = bl < 0 ? 0 : bl;
        B = bl < B ? bl : B;
    }
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long long)gridDim.x * blockDim.x) {
        const float inv = save_invj[i];
        const float g0 = g_normal[i * 3], g1 = g_normal[i * 3 + 1], g2 = g_normal[i * 3 + 2];
        float j0, j1, j2;
        if (inv > 0.f) {
            const float n0 = normal[i * 3], n1 = normal[i * 3 + 1], n2 = normal[i * 3 + 2];
            const float dt = n0 * g0 + n1 * g1 + n2 * g2;
            j0 = -(g0 - n0 * dt) * inv; j1 = -(g1 - n1 * dt) * inv; j2 = -(g2 - n2 * dt) * inv;
        } else {
            j0 = g0 * inv; j1 = g1 * inv; j2 = g2 * inv;
        }
        mx = fmaxf(mx, fmaxf(fabsf(j0), fmaxf(fabsf(j1), fabsf(j2))));
    }
    grid_max_tail(mx, out, pair);
}

void launch_jhat_max(const float* g_normal, const float* normal, const float* save_invj, long long B, unsigned* out, hipStream_t s,
                     const int* n_live, int spr, unsigned* pair) {
    if (!pair) zero_fill(s, out, sizeof(unsigned));
    long long blocks = (B + MAX_THREADS - 1) / MAX_THREADS;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(jhat_max_kernel, dim3((unsigned)blocks), dim3(MAX_THREADS), 0, s, g_normal, normal, save_invj, B, out, n_live, spr, pair);
}

// ---------------------------------------------------------------------------------------------------------- driver
static int dwp_cus() {
    static const int v = [] {
        const char* e = getenv("MNRF_DWP_G");
        if (e && atoi(e) > 0) return atoi(e);
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return v;
}

static DwpPlan dwp_make_plan(int n_eval, const int64_t* B, const int* kinds) {
    DwpPlan p;
    p.n_eval = n_eval;
    for (int e = 0; e < DWP_MAX_EVAL; ++e) { p.n_sb[e] = 0; p.kind[e] = 0; }
    for (int e = 0; e < n_eval; ++e) { p.n_sb[e] = (int)(dwp_tiles128(B[e]) * 4); p.kind[e] = kinds && (kinds[e] & 0xff) ? 1 : 0; }
    p.T = dwp_total(p);
    p.G = dwp_pick_G(p.T, dwp_cus());
    return p;
}

long long dwp_workspace_floats(int n_eval, const int64_t* B, const int* kinds) {
    const DwpPlan p = dwp_make_plan(n_eval, B, kinds);
    return (long long)(p.G + DWP_JOBS * n_eval) * DWP_SLOT_FLOATS;
}

// ---- the plan on the device (live row counts, mnrf_dwp.h DwpDevPlan)
struct DwpPlanArgs {
    int n_eval, cus;
    int cap_sb[DWP_MAX_EVAL];            // sample blocks of the capacity
    int kind[DWP_MAX_EVAL];
    int spr[DWP_MAX_EVAL];
    const int* n_live[DWP_MAX_EVAL];     // null: the capacity
    DwpDevPlan* out;
};
__global__ void dwp_plan_kernel(DwpPlanArgs A) {
    constexpr int NV = DWP_JOBS * DWP_MAX_EVAL;
    __shared__ DwpPlan sp;
    __shared__ long long start[NV + 1];      // cost at which virtual job v starts (job-major), then the total
    const int v = threadIdx.x;
    if (v < DWP_MAX_EVAL) {
        int n = 0;
        if (v < A.n_eval) {
            n = A.cap_sb[v];
            if (A.n_live[v]) {
                long long b = (long long)*A.n_live[v] * A.spr[v];
                b = b < 0 ? 0 : b;
                const long long live = dwp_sample_blocks(b);
                n = live < n ? (int)live : n;
            }
        }
        sp.n_sb[v] = n;
        sp.kind[v] = v < A.n_eval ? A.kind[v] : 0;
    }
    if (v == 0) sp.n_eval = A.n_eval;
    __syncthreads();
    const bool mine = v < DWP_JOBS * A.n_eval;
    const int j = mine ? v / A.n_eval : 0, e = mine ? v % A.n_eval : 0;
    const int n = mine ? dwp_stages(sp, j, e) : 0, w = dwp_weight(j);
    if (v < NV) start[v + 1] = (long long)n * w;
    __syncthreads();
    if (v == 0) {
        start[0] = 0;
        for (int k = 1; k <= NV; ++k) start[k] += start[k - 1];
        sp.T = start[NV];
        sp.G = dwp_pick_G(sp.T, A.cus);
        A.out->plan = sp;
    }
    __syncthreads();
    if (v >= NV) return;
    int lo = 1, hi = 0;
    if (mine && n > 0 && sp.T > 0) {
        lo = dwp_owner(sp, start[v]);
        hi = dwp_owner(sp, start[v] + (long long)(n - 1) * w);
    }
    A.out->g_lo[v] = (short)lo;
    A.out->g_hi[v] = (short)hi;
}

long long dwp_workspace_floats_n(int n_eval) {
    return DWP_DEVPLAN_FLOATS + (long long)(dwp_cus() + DWP_JOBS * n_eval) * DWP_SLOT_FLOATS;
}

// both plans: `plan` made on the host (dev == null) or on the device (dev: where dwp_plan_kernel left it; the launch then has
// one workgroup per CU and `plan` carries n_eval / kinds only)
static int dwp_run(const DwpPlan& plan, const DwpDevPlan* dev, int n_eval, const void* const* x_planes, const void* const* dy_planes,
                   const unsigned* const* seedmax, const int* kinds, float* part, float* const* d_params, int accumulate, hipStream_t s) {
    DwpArgs A;
    A.plan = plan;
    A.dev = dev;
    bool any_half = false;
    for (int e = 0; e < n_eval; ++e) {
        A.ev[e] = DwpEval{(const char*)x_planes[e], (const char*)dy_planes[e], seedmax[e], (kinds && (kinds[e] & 0x1000)) ? 1 : 0};
        any_half |= A.ev[e].y_half != 0;
    }
    for (int e = n_eval; e < DWP_MAX_EVAL; ++e) A.ev[e] = DwpEval{nullptr, nullptr, nullptr, 0};
    A.part = part;
    static const bool once = [] {
        (void)hipFuncSetAttribute((const void*)dwp_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DWP_LDS);
        (void)hipFuncSetAttribute((const void*)dwp_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DWP_RING_LDS);
        (void)hipFuncSetAttribute((const void*)dwp_gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DWP_LDS);
        (void)hipFuncSetAttribute((const void*)dwp_gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DWP_RING_LDS);
        return true;
    }();
    (void)once;
    // the ring of half-stages (dwp_segment_run_h) is the default; MNRF_DWP_RING=0 (read once) selects the two-buffer version
    static const bool ring = [] { const char* e = getenv("MNRF_DWP_RING"); return !(e && atoi(e) == 0); }();
    if (any_half && !ring) return -2;      // (the two-buffer loop fetches whole stages)
    const int G = dev ? dwp_cus() : plan.G;
    if (dev) {
        if (ring) hipLaunchKernelGGL((dwp_gemm_kernel<true, true>), dim3(G), dim3(DWP_WG_THREADS), DWP_RING_LDS, s, A);
        else hipLaunchKernelGGL((dwp_gemm_kernel<false, true>), dim3(G), dim3(DWP_WG_THREADS), DWP_LDS, s, A);
    } else {
        if (ring) hipLaunchKernelGGL(dwp_gemm_kernel<true>, dim3(G), dim3(DWP_WG_THREADS), DWP_RING_LDS, s, A);
        else hipLaunchKernelGGL(dwp_gemm_kernel<false>, dim3(G), dim3(DWP_WG_THREADS), DWP_LDS, s, A);
    }

    DwpFinishArgs F;
    F.part = part;
    F.n_eval = n_eval;
    F.accumulate = accumulate;
    F.dev = dev;
    for (int e = 0; e < DWP_MAX_EVAL; ++e) {
        F.seedmax[e] = e < n_eval ? seedmax[e] : nullptr;
        F.kind[e] = plan.kind[e];
        F.seed_log2[e] = 6 - ((e < n_eval && kinds) ? (kinds[e] >> 8) & 0xf : 0);      // (bits 8-11 of a kind: mnrf.h)
    }
    // owners of every virtual job (host plan), in closed form -- tests/csrc/dwp_plan_check.cpp holds it to the stage-by-stage search
    for (int v = 0; v < DWP_JOBS * DWP_MAX_EVAL; ++v) { F.g_lo[v] = 1; F.g_hi[v] = 0; }
    if (!dev) {
        long long P = 0;
        for (int j = 0; j < DWP_JOBS; ++j) {
            const int w = dwp_weight(j);
            for (int e = 0; e < n_eval; ++e) {
                const int v = j * n_eval + e, n = dwp_stages(plan, j, e);
                if (n > 0) {
                    F.g_lo[v] = (short)dwp_owner(plan, P);
                    F.g_hi[v] = (short)dwp_owner(plan, P + (long long)(n - 1) * w);
                }
                P += (long long)n * w;
            }
        }
    }
    for (int e = 0; e < 64; ++e) F.encpos[e] = 0;
    for (int gq = 0; gq < 4; ++gq)
        for (int t = 0; t < 16; ++t) {
            const int c = enc_col(t, gq);
            if (c >= 0) F.encpos[c] = (short)(16 * (t >> 2) + 4 * gq + (t & 3));
        }
    auto layer = [&](int L, int out_f, int in_f, DwpSource s0, DwpSource s1, int split_col, int bias_job, int bias_row0) {
        DwpLayer& ly = F.layer[L];
        ly.src[0] = s0; ly.src[1] = s1; ly.split_col = (short)split_col;
        ly.out_f = (short)out_f; ly.in_f = (short)in_f; ly.bias_job = (short)bias_job; ly.bias_row0 = (short)bias_row0;
        ly.d_w = d_params[2 * L]; ly.d_b = d_params[2 * L + 1];
    };
    const DwpSource none{-1, 0, 0, 0};
    layer(0, 256, 63, DwpSource{0, 0, 0, 1}, none, 63, 0, 0);                                  // xyz_encoding_1
    for (int i = 1; i < 4; ++i) layer(i, 256, 256, DwpSource{(short)i, 0, 0, 0}, none, 256, i, 0);     // 2..4
    layer(4, 256, 319, DwpSource{5, 0, 0, 1}, DwpSource{4, 0, 0, 0}, 63, 4, 0);                // xyz_encoding_5: encoding columns first
    for (int i = 5; i < 8; ++i) layer(i, 256, 256, DwpSource{(short)(i + 1), 0, 0, 0}, none, 256, i + 1, 0);   // 6..8
    layer(8, 256, 256, DwpSource{9, 0, 0, 0}, none, 256, 9, 0);                                // xyz_encoding_final
    layer(9, 128, 283, DwpSource{11, 0, 0, 0}, DwpSource{12, 0, 0, 0}, 256, 11, 0);            // dir_encoding.0: final | view columns
    layer(10, 1, 256, DwpSource{13, 0, 0, 0}, none, 256, 13, 0);                               // sigma
    layer(11, 3, 128, DwpSource{14, 0, 0, 0}, none, 128, 14, 0);                               // rgb.0
    layer(12, 128, 256, DwpSource{10, 0, 0, 0}, none, 256, 10, 0);                             // normal_net.0
    layer(13, 3, 128, DwpSource{15, 0, 0, 0}, none, 128, 15, 0);                               // normal_net.1
    layer(14, 128, 256, DwpSource{10, 128, 0, 0}, none, 256, 10, 128);                         // is_mirror_net.0
    layer(15, 1, 128, DwpSource{16, 0, 0, 0}, none, 128, 16, 0);                               // is_mirror_net.2
    hipLaunchKernelGGL(dwp_finish_kernel, dim3((64 * 319 + 256 + 255) / 256, 16), dim3(256), 0, s, F);      // (out_f / 4) x in_f + out_f threads at most
    return 0;
}

int launch_dwp(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B,
               const unsigned* const* seedmax, const int* kinds, float* ws, float* const* d_params, int accumulate, hipStream_t s) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL) return -1;
    const DwpPlan plan = dwp_make_plan(n_eval, B, kinds);
    if (plan.T == 0) return accumulate ? 0 : -2;      // nothing to add; an overwrite of nothing is the caller's business
    return dwp_run(plan, nullptr, n_eval, x_planes, dy_planes, seedmax, kinds, ws, d_params, accumulate, s);
}

int launch_dwp_n(int n_eval, const void* const* x_planes, const void* const* dy_planes, const int64_t* B, const int32_t* const* n_live,
                 const int* spr, const unsigned* const* seedmax, const int* kinds, float* ws, float* const* d_params, int accumulate,
                 hipStream_t s) {
    if (n_eval < 1 || n_eval > DWP_MAX_EVAL) return -1;
    DwpPlan cap = dwp_make_plan(n_eval, B, kinds);      // capacities: n_eval, kinds (the device plan has the live counts)
    DwpPlanArgs PA;
    PA.n_eval = n_eval;
    PA.cus = dwp_cus();
    for (int e = 0; e < DWP_MAX_EVAL; ++e) {
        PA.cap_sb[e] = cap.n_sb[e];
        PA.kind[e] = cap.kind[e];
        PA.spr[e] = e < n_eval && spr ? spr[e] : 1;
        PA.n_live[e] = e < n_eval && n_live ? n_live[e] : nullptr;
    }
    DwpDevPlan* dev = (DwpDevPlan*)ws;
    PA.out = dev;
    hipLaunchKernelGGL(dwp_plan_kernel, dim3(1), dim3(192), 0, s, PA);
    static_assert(DWP_JOBS * DWP_MAX_EVAL <= 192, "one thread per virtual job");
    // with nothing live (T == 0) the GEMM workgroups leave at once and the finish kernel writes zeros (or adds nothing)
    return dwp_run(cap, dev, n_eval, x_planes, dy_planes, seedmax, kinds, ws + DWP_DEVPLAN_FLOATS, d_params, accumulate, s);
}

}  // namespace mnrf

// ---------------------------------------------------------------------------------------------------------- streaming-read probe
// What does the memory system deliver to the instruction dwp_gemm_kernel streams with?  One persistent 8-wave workgroup per CU
// reads its contiguous share of `buf` with global_load_lds_dwordx4 (1 KiB per wave-instruction, lane-linear), `depth`
// instructions in flight per wave, into a ring it never reads; aux = 0 (default policy) or 2 (nt).  scripts/bw_probe.py prints
// the rate next to the GEMM's (MI355X_MICROARCH.md quotes 6.4 TB/s default / 6.5-6.8 nt for this instruction chip-wide).
namespace mnrf {
template <int AUX, int DEPTH>
__global__ __launch_bounds__(512, 1) void stream_probe_kernel(const char* __restrict__ buf, long long share) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* src = buf + (long long)blockIdx.x * share + wave * 1024 + lane * 16;
    char* dst = dwp_smem + wave * (DEPTH * 1024);
    const int n = (int)(share / 8192);
    for (int k = 0; k < n; ++k) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)k * 8192),
                                         (__attribute__((address_space(3))) void*)(dst + (k % DEPTH) * 1024), 16, 0, AUX);
        if (k >= DEPTH - 1) {
            if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace mnrf

namespace mnrf {
// The same reader with the GEMM's ADDRESS PATTERN: a workgroup walks consecutive 32-sample stages and takes chunk_a bytes at
// a + s * stride_a and chunk_x bytes at x + s * stride_x per stage (chunks of 1 KiB pieces dealt to the 8 waves), 16 pieces in
// flight per wave, optionally with the GEMM's raw barrier per stage.  Separates "strided chunks instead of one contiguous run" and
// "eight waves in lock step" from everything else that distinguishes dwp_gemm_kernel from the contiguous probe above.
// HALVES: the GEMM ring's lane pattern -- a stage travels as two half-stages, an instruction takes rows 0-15 (or 16-31) of the hi
// tile (lanes 0-31) and of the lo tile (lanes 32-63) of a feature block: two 512-byte runs instead of 1 KiB contiguous
// BULK (1: the GEMM ring with D = 5, 2: a deeper one): the wave does not wait instruction by instruction for its oldest request but,
// once per stage, until all but 4 (12) of its requests have landed -- the stage is complete -- then the barrier, then the next
// stage's 8 requests in one burst
template <bool BARRIER, bool HALVES, int BULK = 0>
__global__ __launch_bounds__(512, 1) void stream_probe2_kernel(const char* __restrict__ a, const char* __restrict__ x, int n_stages,
                                                               long long stride_a, long long stride_x, int pieces_a, int pieces_x,
                                                               int windows) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // `windows` jobs, job-major like the GEMM's cost line: job w reads the w-th chunk-sized window of every stage's stride
    const long long units = (long long)windows * n_stages;
    const long long u0 = blockIdx.x * units / gridDim.x, u1 = (blockIdx.x + 1) * units / gridDim.x;
    char* dst = dwp_smem + wave * (16 * 1024);
    const int per_wave = (pieces_a + pieces_x + 7) / 8;      // pieces per wave and stage (a wave past the end re-loads piece 0)
    int k = 0;
    for (long long u = u0; u < u1; ++u) {
        const int w = (int)(u / n_stages), s = (int)(u - (long long)w * n_stages);
        const int lsrc = HALVES ? (lane >> 5) * 1024 + (lane & 31) * 16 : lane * 16;
        const char* pa = a + (long long)s * stride_a + (long long)w * pieces_a * 1024 + lsrc;
        const char* px = x + (long long)s * stride_x + (long long)w * pieces_x * 1024 + lsrc;
        for (int q = 0; q < per_wave; ++q, ++k) {
            int pc = wave + 8 * q;
            if (pc >= pieces_a + pieces_x) pc = 0;
            const char* src = pc < pieces_a ? pa : px;
            if (pc >= pieces_a) pc -= pieces_a;
            // HALVES: piece pc = half h of feature block fb of the chunk (all blocks' half 0 first: a half-stage), else KiB pc
            const int np = pc < pieces_a && src == pa ? pieces_a : pieces_x;
            src += HALVES ? (pc % (np / 2)) * 2048 + (pc / (np / 2)) * 512 : pc * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (k & 15) * 1024), 16, 0, PL_LOAD_AUX);
            if (BULK == 0 && k >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
        if (BULK == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (BULK == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace mnrf

extern "C" int mnrf_bench_stream2(const void* a, const void* x, int n_stages, int64_t stride_a, int64_t stride_x, int chunk_a, int chunk_x,
                                  int windows, int barrier, void* stream) {
    using namespace mnrf;
    if (!a || !x || n_stages < 256 || chunk_a < 1024 || chunk_x < 0 || (chunk_a & 1023) || (chunk_x & 1023) || windows < 1 ||
        (int64_t)windows * chunk_a > stride_a || (int64_t)windows * chunk_x > stride_x)
        return mnrf_fail(MNRF_ERR_ARG, "mnrf_bench_stream2: chunks in whole KiB, at least 256 stages");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipStream_t s = (hipStream_t)stream;
    const bool halves = (barrier & 2) != 0;
    if (barrier & 4) hipLaunchKernelGGL((stream_probe2_kernel<true, true, 1>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                                    (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    else if (barrier & 8) hipLaunchKernelGGL((stream_probe2_kernel<true, true, 2>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                                    (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    else if ((barrier & 1) && halves) hipLaunchKernelGGL((stream_probe2_kernel<true, true>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                                    (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    else if (barrier & 1) hipLaunchKernelGGL((stream_probe2_kernel<true, false>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                                    (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    else if (halves) hipLaunchKernelGGL((stream_probe2_kernel<false, true>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                            (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    else hipLaunchKernelGGL((stream_probe2_kernel<false, false>), dim3(cus), dim3(512), 128 * 1024, s, (const char*)a, (const char*)x, n_stages,
                            (long long)stride_a, (long long)stride_x, chunk_a / 1024, chunk_x / 1024, windows);
    return mnrf_check_launch("mnrf_bench_stream2");
}

extern "C" int mnrf_bench_stream(const void* buf, int64_t bytes, int aux, int depth, void* stream) {
    using namespace mnrf;
    if (!buf || bytes < (int64_t)256 * 8192) return mnrf_fail(MNRF_ERR_ARG, "mnrf_bench_stream: buffer too small");
    if ((aux != 0 && aux != 2) || (depth != 8 && depth != 16)) return mnrf_fail(MNRF_ERR_ARG, "mnrf_bench_stream: aux 0|2, depth 8|16");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long share = bytes / cus / 8192 * 8192;
    const size_t lds = (size_t)8 * depth * 1024;
    hipStream_t s = (hipStream_t)stream;
    if (aux == 0 && depth == 8) hipLaunchKernelGGL((stream_probe_kernel<0, 8>), dim3(cus), dim3(512), lds, s, (const char*)buf, share);
    else if (aux == 2 && depth == 8) hipLaunchKernelGGL((stream_probe_kernel<2, 8>), dim3(cus), dim3(512), lds, s, (const char*)buf, share);
    else if (aux == 0) hipLaunchKernelGGL((stream_probe_kernel<0, 16>), dim3(cus), dim3(512), lds, s, (const char*)buf, share);
    else hipLaunchKernelGGL((stream_probe_kernel<2, 16>), dim3(cus), dim3(512), lds, s, (const char*)buf, share);
    return mnrf_check_launch("mnrf_bench_stream");
}
