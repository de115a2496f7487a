// mnrf_dwp.h -- weight gradients from operand PLANES (round 3): layout, job table and work plan shared by the producer
// kernels (training forward / activation-gradient kernels of mnrf_field_split*.inc), the GEMM + finish kernels
// (mnrf_dwp.hip) and a host-side unit test of the index arithmetic (tests/test_dwp_plan_cpu.py compiles this header with g++).
//
// dW[n][k] = sum_s dY[s][n] X[s][k] contracts over SAMPLES, so both operands of its MFMAs want sample-contiguous data, while
// the field kernels hold a sample per lane with its features in registers.  Rounds 1-2 kept X and dY as fp32 rows and let the
// GEMM convert (fp32 -> 3 x bf16) and transpose while staging: 5.5 VALU per element, twice per element, and that conversion
// -- not the matrix pipe, not HBM -- bounded the GEMM.  Now the producers write what they already hold: the
// hi/lo f16 B operands of the next Linear (X) / of the transposed chain (dY), as 1-KiB TILES
//     tile(sb, fb, plane) = [32 samples of sample block sb][16 features of feature block fb] f16,  32 B per sample row,
// one wave-level global_store_dwordx2 = 512 contiguous bytes (lane (m, g) of sample group s: row 16 s + m, bytes 8 g .. 8 g + 7
// = features 4 g .. 4 g + 3 of the block).  The GEMM brings tiles into LDS with lane-linear LDS-DMA (global_load_lds_dwordx4,
// no VALU) and reads MFMA operands with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane (i, gg) column i of rows
// 4 gg .. 4 gg + 3 -- feature i, four samples -- so two reads (rows 0-15, rows 16-31 of the tile) make one
// v_mfma_f32_16x16x32_f16 operand whose k slots 8 gg .. 8 gg + 7 are samples {4 gg .. 4 gg + 3, 16 + 4 gg .. 16 + 4 gg + 3} for BOTH
// operands (the contraction order over samples is free).  Arithmetic: hi.hi + hi.lo + lo.hi, fp32 accumulation (three
// products instead of the six of the bf16 x 3 scheme; half the operand bytes of fp32 rows read twice).
//
// Scale of dY: activation gradients have no natural scale; the backward kernel carries every sample with its own power of
// two (mnrf_field_split_bwd.inc).  A contraction over samples needs ONE scale per launch: a pre-pass takes the maximum seed
// magnitude of the evaluation, the backward kernel multiplies the hi/lo halves of sample s by 2^(K - k_s) <= 1 (exact; what
// falls below f16's 2^-24 is below 2^-30 of the largest gradient of the evaluation), the finish kernel multiplies by 2^-K.
#pragma once
#include <stdint.h>

#ifndef __HIPCC__      // plain g++ (tests/test_dwp_plan_cpu.py): the qualifiers of the shared inline functions mean nothing there
#ifndef __host__
#define __host__
#define __device__
#endif
#endif

#include "mnrf_layout.h"

namespace mnrf {

// dY planes carry 2^(K + PL_BOOST_LOG2) dY: K puts the largest SEED of the evaluation into [2^6, 2^7) like the per-sample
// scales of the backward kernel do; trunk gradients are typically orders of magnitude below the head seeds, and what
// bounds the precision of a small hi/lo pair is f16's subnormal step 2^-24 -- four more bits of scale keep the weight
// gradients of the first layers within ~4e-6 of the fp32-row route (3e-5 without).  The price is headroom: a scaled
// gradient of 65504 / 16 or more would overflow a plane; the backward kernel raises MNRF_GUARD_SATURATED already there.
constexpr int PL_BOOST_LOG2 = 4;
// cache policy of the producers' plane stores (aux operand of raw_buffer_store: 0 default, 2 = non-temporal).  The planes are
// 2.7 GB per kernel that nothing reads again before the weight-gradient GEMM.
// Measured (scripts/bench_train.py, alternating libraries on one box): 5.97 ms per step with the default policy, 5.90 with
// sc0 (1), **5.47-5.55 with nt (2)**, 5.76 with 3, 5.9-6.1 with sc1 (16), 5.49 with nt + sc1 (18).  Counters
// (profiles/r03r_store_policy_pmc.txt): same fetched bytes, same L2 hits / misses; the L2's memory-side write requests stall
// 34-84 % longer with the default policy (lines allocated, written back later), and the weight stream's counted waits sit
// behind the stores.
#ifndef MNRF_EXP_STORE_AUX
#define MNRF_EXP_STORE_AUX 2
#endif
// ... and of the GEMM's LDS-DMA loads of them: nt (2) measured 1.365 against 1.40-1.43 ms for the two evaluations of a training
// step and 0.1-0.2 ms per step in alternating runs (round 4; the streaming probe reads 6.87 TB/s with nt, 6.06 without)
#ifndef MNRF_EXP_LOAD_AUX
#define MNRF_EXP_LOAD_AUX 2
#endif
constexpr int PL_LOAD_AUX = MNRF_EXP_LOAD_AUX;      // the same operand of the GEMM's LDS-DMA loads of the planes
constexpr int PL_STORE_AUX = MNRF_EXP_STORE_AUX;
constexpr int PL_TILE_BYTES = 1024;                  // one plane of one feature block of one sample block
constexpr int PL_FB_BYTES = 2 * PL_TILE_BYTES;       // [hi tile][lo tile]
constexpr int PL_SB = 32;                            // samples per sample block = one wave of the field kernels (2 groups x 16)
// X planes: the sections of mnrf_layout.h SEC_* in units of 16 features
constexpr int PLX_FB = SAVE_FLOATS / 16;             // 174
constexpr long long PLX_SB_BYTES = (long long)PLX_FB * PL_FB_BYTES;
// planes are written by whole 128-sample workgroups of the field kernels: 4 sample blocks per tile, rows past B hold zeros in dY
__host__ __device__ inline long long dwp_tiles128(long long B) { return (B + 127) / 128; }
__host__ __device__ inline long long dwp_sample_blocks(long long B) { return 4 * dwp_tiles128(B); }
// dY planes: DY_* sections, plus one block for the sigma seed (row 0 = dL/dsigma)
constexpr int PLY_SIG = DY_FLOATS / 16;              // 171
constexpr int PLY_FB = PLY_SIG + 1;                  // 172
constexpr long long PLY_SB_BYTES = (long long)PLY_FB * PL_FB_BYTES;

// ---- second-order pass (round 4): the gradient through the density-gradient normal adds  dW_i += sum_s b_i[s] (x) a'_(i-1)[s]
// to the 8 trunk layers and  dw_sigma += sum_s a'_8[s]  (mnrf_field_split_bwd.inc, field_split_bwd2_kernel).  Its operands are
// the B operands that kernel already holds for its own GEMMs: the tangents a' (X2 planes: [xyz-encoding tangent | a'_1..a'_8],
// the first 132 feature blocks of the X layout) and the density-gradient signals b (Y2 planes: [b_1..b_8 | one block whose
// row 0 is 2^boost], the first 128 feature blocks of the dY layout + 1).  In the kernel the tangents of sample s carry 2^k_s (its
// J^ normalised to [1, 2)); their planes carry 2^(K2 - k_s) <= 1 of that with K2 from the largest |J^| of the launch (a pre-pass,
// like the seeds' maximum), the b planes 2^boost, and the finish kernel multiplies by 2^-(K2 + boost).  For the GEMM
// such an evaluation is one more entry of the tape with KIND 1: same job numbers, the jobs of the heads have no stages.
static_assert(TA_ENC == SEC_ENC && TA_H == SEC_H && BS_L == TA_H + 8 * 256, "second-order sections sit where the first-order ones do");
constexpr int PL2X_FB = (TA_H + 8 * 256) / 16;       // 132
constexpr long long PL2X_SB_BYTES = (long long)PL2X_FB * PL_FB_BYTES;
constexpr int PL2Y_SIG = 8 * 256 / 16;               // 128: row 0 = 2^boost (the A operand of dw_sigma += sum a'_8)
constexpr int PL2Y_FB = PL2Y_SIG + 1;                // 129
constexpr long long PL2Y_SB_BYTES = (long long)PL2Y_FB * PL_FB_BYTES;
__host__ __device__ inline long long dwp_x_stride(int kind) { return kind ? PL2X_SB_BYTES : PLX_SB_BYTES; }
__host__ __device__ inline long long dwp_y_stride(int kind) { return kind ? PL2Y_SB_BYTES : PLY_SB_BYTES; }

// ---- the GEMMs of one evaluation of the field (first-order pass): rows = dY feature blocks, columns = X feature blocks
struct DwpJob {
    short ya, na;      // first feature block and block count of the dY section (<= 16)
    short xa, nx;      // first feature block and block count of the X section (<= 16)
    short bias;        // 1: the column sums of dY (bias gradient) ride along
    short shape;       // instantiation of the GEMM body, see dwp_shape()
};
constexpr int DWP_JOBS = 17;
constexpr int DWP_MAX_EVAL = 8;
constexpr int DWP_SLOT_FLOATS = 256 * 256 + 256;     // one partial tile + its bias sums
constexpr int DWP_WG_THREADS = 512;

__host__ __device__ inline DwpJob dwp_job(int j) {
    constexpr int H = SEC_H / 16;        // 4: h1 .. h8 at H + 16 (i - 1)
    switch (j) {
    case 0: return DwpJob{0, 16, 0, 4, 1, 1};                                     // L1: dY_1 x enc
    case 1: case 2: case 3:
        return DwpJob{(short)(16 * j), 16, (short)(H + 16 * (j - 1)), 16, 1, 0};  // L2..L4: dY_i x h_(i-1)
    case 4: return DwpJob{64, 16, (short)(H + 48), 16, 1, 0};                     // L5, hidden columns: dY_5 x h4
    case 5: return DwpJob{64, 16, 0, 4, 0, 1};                                    // L5, encoding columns
    case 6: case 7: case 8:
        return DwpJob{(short)(16 * (j - 1)), 16, (short)(H + 16 * (j - 2)), 16, 1, 0};   // L6..L8 (layer i = j): dY_i x h_(i-1)
    case 9: return DwpJob{DY_FIN / 16, 16, (short)(H + 112), 16, 1, 0};           // xyz_encoding_final x h8
    case 10: return DwpJob{DY_NRM1 / 16, 16, (short)(H + 112), 16, 1, 0};         // normal_net.0 | is_mirror_net.0 (adjacent) x h8
    case 11: return DwpJob{DY_DIR / 16, 8, SEC_FIN / 16, 16, 1, 2};               // dir_encoding, final columns
    case 12: return DwpJob{DY_DIR / 16, 8, SEC_DIRE / 16, 2, 0, 3};               // dir_encoding, view columns
    case 13: return DwpJob{PLY_SIG, 1, (short)(H + 112), 16, 1, 4};               // sigma x h8
    case 14: return DwpJob{DY_RGB / 16, 1, SEC_HD / 16, 8, 1, 5};                 // rgb x hd
    case 15: return DwpJob{DY_NRM2 / 16, 1, SEC_HN / 16, 8, 1, 5};                // normal_net.1 x hn
    default: return DwpJob{DY_MIR2 / 16, 1, SEC_HM / 16, 8, 1, 5};                // is_mirror_net.2 x hm
    }
}
static_assert(DY_NRM1 + 128 == DY_MIR1, "normal_net.0 and is_mirror_net.0 gradients are adjacent: one 256-row job");
// jobs of an evaluation of kind 1 (second-order planes): the trunk (0..8) and sigma (13), no bias sums
__host__ __device__ inline bool dwp_has(int kind, int j) { return kind == 0 || j <= 8 || j == 13; }
__host__ __device__ inline DwpJob dwp_job_of(int kind, int j) {
    DwpJob jb = dwp_job(j);
    if (kind) {
        jb.bias = 0;
        if (j == 13) jb.ya = PL2Y_SIG;
    }
    return jb;
}

// cost of one stage (32 samples) of job j: KiB of operand tiles (the GEMM is HBM-bound, work is dealt by bytes) plus what a
// stage costs whatever its size -- barrier, counted wait, request -- expressed in KiB: without it the workgroups that own the
// small jobs (18-40 KiB per stage) finish last
#ifndef MNRF_EXP_DWP_STAGE_KIB
#define MNRF_EXP_DWP_STAGE_KIB 0
#endif
constexpr int DWP_STAGE_KIB = MNRF_EXP_DWP_STAGE_KIB;
constexpr int DWP_MAX_WEIGHT = 64 + DWP_STAGE_KIB;
__host__ __device__ inline int dwp_weight(int j) {
    const DwpJob jb = dwp_job(j);
    return 2 * (jb.na + jb.nx) + DWP_STAGE_KIB;
}

// ---- work plan.  Virtual job v = j * n_eval + e (job-major); its stage s (sample block s of evaluation e) starts at cost
// P_v + s * w_j on a line of total length T.  Workgroup g of G owns the stages that START in [g T / G, (g + 1) T / G): every
// stage has exactly one owner, a workgroup's stages of one virtual job are consecutive (one partial tile per (g, v), slot
// g + v), and with G <= T / 256 every workgroup between the first and the last owner of a virtual job owns at least one of
// its stages (an interval is at least four times the heaviest stage).
struct DwpPlan {
    int n_eval;
    int n_sb[DWP_MAX_EVAL];
    int kind[DWP_MAX_EVAL];      // 0: first-order planes, 1: second-order planes
    int G;
    long long T;
};

// stages of virtual job (j, e): the evaluation's sample blocks, or none where its kind has no such job
__host__ __device__ inline int dwp_stages(const DwpPlan& p, int j, int e) { return dwp_has(p.kind[e], j) ? p.n_sb[e] : 0; }

__host__ __device__ inline long long dwp_total(const DwpPlan& p) {
    long long t = 0;
    for (int j = 0; j < DWP_JOBS; ++j)
        for (int e = 0; e < p.n_eval; ++e) t += (long long)dwp_stages(p, j, e) * dwp_weight(j);
    return t;
}

__host__ __device__ inline int dwp_pick_G(long long T, int cus) {
    long long g = T / (4 * DWP_MAX_WEIGHT);
    if (g < 1) g = 1;
    return (int)(g < cus ? g : cus);
}

// the owner interval [c0, c1) of workgroup g on the cost line (two 64-bit divisions, once per workgroup)
__host__ __device__ inline void dwp_interval(const DwpPlan& p, int g, long long& c0, long long& c1) {
    c0 = (long long)g * p.T / p.G;
    c1 = (long long)(g + 1) * p.T / p.G;
}

// the owner of the stage that STARTS at cost c: the g with floor(g T / G) <= c < floor((g + 1) T / G), in closed form
// (g = ceil((c + 1) G / T) - 1).  The owners of a virtual job are owner(P_v) .. owner(P_v + (n - 1) w), every workgroup
// in between owning at least one of its stages (see above): what the finish kernel sums over.
__host__ __device__ inline int dwp_owner(const DwpPlan& p, long long c) { return (int)(((c + 1) * p.G - 1) / p.T); }

// ---- round 5: the plan made ON THE DEVICE.  With live row counts (include/mnrf.h) the sample count of an evaluation -- the
// reflected rays of a training step -- is known to the device only; dwp_plan_kernel writes this block from the counts, the
// GEMM and the finish kernel read it (the launch is sized for the capacity, workgroups g >= plan.G leave at once).
struct DwpDevPlan {
    DwpPlan plan;
    short g_lo[DWP_JOBS * DWP_MAX_EVAL], g_hi[DWP_JOBS * DWP_MAX_EVAL];     // owners of virtual job v (g_lo > g_hi: none)
};
constexpr int DWP_DEVPLAN_FLOATS = (int)((sizeof(DwpDevPlan) + 15) / 16 * 4);     // its room at the head of the workspace

// smallest s with P + s w >= c, clamped to [0, n]   (32-bit division: 64-bit ones cost hundreds of instructions on the GPU)
__host__ __device__ inline int dwp_first_stage_at(long long c, long long P, int w, int n) {
    const long long d = c - P;
    if (d <= 0) return 0;
    if (d >= (long long)w * n) return n;
    return (int)(((unsigned)d + (unsigned)w - 1u) / (unsigned)w);       // d < w n < 2^32 (n < 2^26 sample blocks)
}

// stages [s_lo, s_hi) of a virtual job (starting at cost P, weight w, n stages) owned by the workgroup with interval [c0, c1)
__host__ __device__ inline void dwp_segment(long long c0, long long c1, long long P, int w, int n, int& s_lo, int& s_hi) {
    s_lo = dwp_first_stage_at(c0, P, w, n);
    s_hi = dwp_first_stage_at(c1, P, w, n);
}

}  // namespace mnrf
