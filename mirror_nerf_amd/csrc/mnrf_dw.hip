// mnrf_dw.hip -- weight gradients of the field MLP (training) for gfx950.
//
// The backward kernel (mnrf_field_bwd.inc) leaves, per Linear, the pre-activation gradient dY
// [samples][N] and the training forward leaves its input X [samples][K], both row-major in HBM
// (288 GB of HBM3E make keeping ~22 KB per sample affordable; recomputing them would cost another
// forward).  dW[n][k] = sum_s dY[s][n] X[s][k] is a GEMM whose contraction runs over samples:
//   * dw_gemm_kernel: 128 x TK output tile per workgroup, v_mfma_f32_16x16x4_f32 with the sample
//     axis as the MFMA k axis (A operand = 4 rows x 16 columns of dY, B operand = 4 rows x 16
//     columns of X, straight out of padded LDS tiles, bank-conflict free), split over the sample
//     range; bias gradients (column sums of dY) fall out of the A operands already in registers;
//   * dw_small_kernel: the 1- and 3-row Linears (sigma, rgb, normal_net.1, is_mirror_net.2);
//   * dw_finish_kernel: sums the split partials and writes gradients in nn.Linear (out,in) layout.
// Autograd equivalent: the .grad accumulation of loss.backward() for models/mirror_nerf.py:59-99.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/mnrf.h"
#include "mnrf_error.h"
#include "mnrf_layout.h"
#include "mnrf_dw.h"

namespace mnrf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DW_TN = 128;
constexpr int DW_CH = 32;          // samples per LDS stage
constexpr int DW_PAD = 16;         // row stride = width + 16 floats: lanes of rows r and r+1 hit different bank halves

template <int TK>
__global__ __launch_bounds__(256) void dw_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ X, int ldx,
                                                      long long B, int splits, int N, int K, float* __restrict__ Cpart,
                                                      float* __restrict__ bpart) {
    constexpr int LDA = DW_TN + DW_PAD;
    constexpr int LDX = TK + DW_PAD;
    __shared__ float As[DW_CH * LDA];
    __shared__ float Xs[DW_CH * LDX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wn = wave >> 1, wk = wave & 1;
    constexpr int MB = 4;              // 64 rows of dW per wave
    constexpr int KB = TK / 32;        // TK/2 columns per wave
    const int n0 = blockIdx.x * DW_TN, k0 = blockIdx.y * TK, split = blockIdx.z;
    const long long per = (B + splits - 1) / splits;
    const long long s_begin = split * per;
    const long long s_end = s_begin + per < B ? s_begin + per : B;

    f32x4 acc[MB][KB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[MB] = {0.f, 0.f, 0.f, 0.f};

    for (long long s0 = s_begin; s0 < s_end; s0 += DW_CH) {
        __syncthreads();
        // stage DW_CH rows of dY (128 columns) and X (TK columns); rows past the range are zero
        for (int v = tid; v < DW_CH * (DW_TN / 4); v += 256) {
            const int r = v / (DW_TN / 4), c4 = v % (DW_TN / 4);
            f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s0 + r < s_end) val = *(const f32x4*)(A + (s0 + r) * lda + n0 + c4 * 4);
            *(f32x4*)(As + r * LDA + c4 * 4) = val;
        }
        for (int v = tid; v < DW_CH * (TK / 4); v += 256) {
            const int r = v / (TK / 4), c4 = v % (TK / 4);
            f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s0 + r < s_end) val = *(const f32x4*)(X + (s0 + r) * ldx + k0 + c4 * 4);
            *(f32x4*)(Xs + r * LDX + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < DW_CH / 4; ++ks) {
            float a[MB], b[KB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = As[(ks * 4 + g) * LDA + wn * 64 + mb * 16 + i];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b[kb] = Xs[(ks * 4 + g) * LDX + wk * (TK / 2) + kb * 16 + i];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                bsum[mb] += a[mb];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    acc[mb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b[kb], acc[mb][kb], 0, 0, 0);
            }
        }
    }
    // partial tile: C[n = 4g + r][k = i] of every 16x16 block
    float* C = Cpart + (long long)split * N * K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + mb * 16 + 4 * g + r;
                const int k = k0 + wk * (TK / 2) + kb * 16 + i;
                C[(long long)n * K + k] = acc[mb][kb][r];
            }
    if (bpart && blockIdx.y == 0 && wk == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float v = bsum[mb];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) bpart[(long long)split * N + n0 + wn * 64 + mb * 16 + i] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// The same GEMM on the bf16 matrix pipe at fp32 accuracy.  A fp32 number is EXACTLY the sum of three bf16 numbers
// obtained by truncation (8 + 8 + 8 significand bits, fp32's exponent range -- gradients of 1e-10 are as safe as
// activations of 100, which is why this kernel does not use the f16 split of the field kernel):
//     v = hi + mid + lo,   hi = trunc16(v), mid = trunc16(v - hi), lo = trunc16(v - hi - mid)   (both subtractions exact)
// and dY^T X is evaluated as  hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi  (the three dropped products are
// <= 2^-24 relative) with six v_mfma_f32_16x16x32_bf16 per 32-sample step into fp32 accumulators: 6 x 16 cycles
// against 8 x 32 cycles of v_mfma_f32_16x16x4_f32 for the same step.
// Staging: each thread converts (2 samples x 4 columns) blocks and stores them TRANSPOSED, [column][sample] with two
// samples per 32-bit word, so that an MFMA operand (8 consecutive samples of one column) is one ds_read_b128; the
// 96-byte column stride (64 + 32 pad) is conflict-free for the 4 x 16 lane groups of ds_read_b128.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int DWS_CH = 32;       // samples per stage = one MFMA k-step
constexpr int DWS_STRIDE = 96;   // bytes per column in LDS

__device__ __forceinline__ void split3(float v, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
    const float r1 = v - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
}

// rows (s, s+1) x 4 columns -> three planes, word [column][s/2] = (bf16(row s) | bf16(row s+1) << 16)
__device__ __forceinline__ void stage_block(char* plane0, int plane_bytes, int col, int rp, const f32x4& r0, const f32x4& r1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(r0[c], h0, m0, l0);
        split3(r1[c], h1, m1, l1);
        char* w = plane0 + (col + c) * DWS_STRIDE + rp * 4;
        *(unsigned*)(w) = (h0 >> 16) | h1;
        *(unsigned*)(w + plane_bytes) = (m0 >> 16) | m1;
        *(unsigned*)(w + 2 * plane_bytes) = (l0 >> 16) | l1;
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// All GEMMs of one backward pass with the same TK run as ONE launch (grid.z = job): a 1024-ray training batch gives
// each GEMM only a few hundred workgroups, and launching them one after the other needed 256 sample-range splits to
// fill the chip -- 600 MB of partial tiles written and re-read per pass.  Batched, 64 splits fill it.
struct DwJob {
    const float* A; const float* X; float* C; float* bp;
    int lda, ldx, N, K;
};
struct DwJobs {
    DwJob job[12];
};

template <int TK>
__global__ __launch_bounds__(256) void dw_gemm_bf16_kernel(DwJobs J, long long B, int splits) {
    const DwJob jb = J.job[blockIdx.z];
    const float* __restrict__ A = jb.A;
    const float* __restrict__ X = jb.X;
    float* __restrict__ Cpart = jb.C;
    float* __restrict__ bpart = jb.bp;
    const int lda = jb.lda, ldx = jb.ldx, N = jb.N, K = jb.K;
    const int tiles_n = N / DW_TN;
    const int tile_n = blockIdx.x % tiles_n, tile_k = blockIdx.x / tiles_n;
    if (tile_k >= K / TK) return;
    extern __shared__ __attribute__((aligned(16))) char dws[];
    constexpr int PA = DW_TN * DWS_STRIDE;      // one plane of dY^T
    constexpr int PX = TK * DWS_STRIDE;         // one plane of X^T
    char* As = dws;                              // 3 planes
    char* Xs = dws + 3 * PA;                     // 3 planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wn = wave >> 1, wk = wave & 1;
    constexpr int MB = 4;              // 64 rows of dW per wave
    constexpr int KB = TK / 32;        // TK/2 columns per wave
    constexpr int XQ = (TK / 4 + 15) / 16;               // column-quad rounds of the X tile (16 quads per round)
    const int n0 = tile_n * DW_TN, k0 = tile_k * TK, split = blockIdx.y;
    const long long per = (B + splits - 1) / splits;
    const long long s_begin = split * per;
    const long long s_end = s_begin + per < B ? s_begin + per : B;

    f32x4 acc[MB][KB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Work split of the staging: a thread owns ROW PAIR rp = lane & 15 of every stage and column quads
    // cq = (lane >> 4) + 4 * wave + 16 * q.  Within one ds_write_b32 the 32 lanes of a bank group then cover 16 row
    // pairs x 2 quads: bank = (24 * column + rp) mod 32 differs in rp, 2-way in the quad (free for stores).  (Column
    // quads across lanes -- the coalesced choice -- put all 32 lanes on ONE bank: 22.5 ms per training step vs 20.0.)
    const int rp = lane & 15, cq0 = (lane >> 4) + 4 * wave;
    float bsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // this thread's dY columns (the same in every stage)
    f32x4 ra[2][2], rx[XQ][2];
    auto fetch = [&](long long s0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long long row = s0 + 2 * rp + r;
                ra[q][r] = row < s_end ? *(const f32x4*)(A + row * lda + n0 + (cq0 + 16 * q) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int cq = cq0 + 16 * q;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long long row = s0 + 2 * rp + r;
                rx[q][r] = (cq < TK / 4 && row < s_end) ? *(const f32x4*)(X + row * ldx + k0 + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    fetch(s_begin);
    for (long long s0 = s_begin; s0 < s_end; s0 += DWS_CH) {
        __syncthreads();      // the previous stage's operands have been read
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            stage_block(As, PA, (cq0 + 16 * q) * 4, rp, ra[q][0], ra[q][1]);
#pragma unroll
            for (int c = 0; c < 4; ++c) bsum[q][c] += ra[q][0][c] + ra[q][1][c];
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q)
            if (cq0 + 16 * q < TK / 4) stage_block(Xs, PX, (cq0 + 16 * q) * 4, rp, rx[q][0], rx[q][1]);
        __syncthreads();
        if (s0 + DWS_CH < s_end) fetch(s0 + DWS_CH);      // next stage's rows travel during the MFMAs
        u32x4 b[KB][3];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[kb][pl] = *(const u32x4*)(Xs + pl * PX + (wk * (TK / 2) + kb * 16 + i) * DWS_STRIDE + g * 16);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            u32x4 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = *(const u32x4*)(As + pl * PA + (wn * 64 + mb * 16 + i) * DWS_STRIDE + g * 16);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                f32x4 c = acc[mb][kb];
                c = mfma_bf16(a[2], b[kb][0], c);    // lo . hi
                c = mfma_bf16(a[0], b[kb][2], c);    // hi . lo
                c = mfma_bf16(a[1], b[kb][1], c);    // mid . mid
                c = mfma_bf16(a[1], b[kb][0], c);    // mid . hi
                c = mfma_bf16(a[0], b[kb][1], c);    // hi . mid
                c = mfma_bf16(a[0], b[kb][0], c);    // hi . hi
                acc[mb][kb] = c;
            }
        }
    }
    // partial tile: C[n = 4g + r][k = i] of every 16x16 block
    float* C = Cpart + (long long)split * N * K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + mb * 16 + 4 * g + r;
                const int k = k0 + wk * (TK / 2) + kb * 16 + i;
                C[(long long)n * K + k] = acc[mb][kb][r];
            }
    if (bpart && tile_k == 0) {
        // column sums of dY: the 16 lanes of a lane group hold the 16 row pairs of the same columns
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = bsum[q][c];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                if (rp == 0) bpart[(long long)split * N + n0 + (cq0 + 16 * q) * 4 + c] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined version of dw_gemm_bf16_kernel<128> (round 2).  The kernel above runs in phases -- wait for the stage's rows,
// convert and store them, barrier, MFMAs, barrier -- and leans on a second resident workgroup to fill the matrix pipe
// while one converts.  Here ONE workgroup of EIGHT waves per CU owns two LDS buffers and four register sets, and every
// wave overlaps the three activities itself:
//   stage i:  global loads of stage i+4 -> register set i % 4            (three stages of load latency hidden)
//             MFMAs on LDS buffer i % 2, and BETWEEN them the conversion of stage i+1 (register set (i+1) % 4) into
//             LDS buffer (i+1) % 2: one column task (2 samples x 1 column -> 3 words) per 6 MFMAs
//             one barrier
// Two waves per SIMD, not one: a single wave issues its ~450 non-MFMA instructions per stage one dependent instruction at
// a time (measured with 4 waves: 2400 cycles per stage with the MFMAs compiled out, whatever the prefetch depth).
// Sample ranges of the splits are multiples of 32; the one partial stage of a launch is done behind the pipeline.
// OUTCOME (profiles/r02i_dw_ab.txt): the largest launch of a 1024-ray step takes 1.23 ms against 1.12 ms of the phased
// kernel, the step 7.96 vs 7.73 ms -- no gain, so the phased kernel stays the default (MNRF_DW_PIPE=1 selects this one).
// What the exercise established: both kernels sit at ~2500-2800 cycles per 32-sample stage against 1536 cycles of MFMA
// time because the fp32 -> 3 x bf16 conversion (5.5 VALU per element, 832 VALU cycles per SIMD and stage) plus the
// MFMA issue slots fill the vector issue port to ~80 %; the XCD-aware tile placement halves the L2 misses (5.2 GB per
// step instead of 10.1, L2 hit rate 47 %) without changing the time, i.e. the GEMM was not HBM-bound either.
constexpr int DWP_WAVES = 8;
constexpr int DWP_TK = 128;
struct DwpRegs {
    f32x4 a[2], x[2];      // one column quad of dY and one of X, rows 2*rp and 2*rp + 1
};

__device__ __forceinline__ void stage_col(char* plane0, int plane_bytes, int col, int rp, float v0, float v1) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    char* w = plane0 + col * DWS_STRIDE + rp * 4;
    *(unsigned*)(w) = (h0 >> 16) | h1;
    *(unsigned*)(w + plane_bytes) = (m0 >> 16) | m1;
    *(unsigned*)(w + 2 * plane_bytes) = (l0 >> 16) | l1;
}

template <bool HINT>
__global__ __launch_bounds__(64 * DWP_WAVES, 1) void dw_gemm_bf16p_kernel(DwJobs J, long long B, int splits, int njobs) {
    constexpr int TK = DWP_TK;
    // Workgroup -> (job, split, tile).  The up-to-four tiles of one (job, split) read the same 32 rows of dY and X per
    // stage, two tiles each: at full matrix-pipe rate a 128 x 128 tile needs 32 KB of operands per 0.73 us, 11 TB/s over
    // the chip -- more than HBM delivers (the phased kernel: 4.9 TB/s of L2 misses).  Workgroups are dealt to the 8 XCDs
    // round robin (id % 8) and each XCD has its own L2: the four tiles of a pair get ids 8 apart -- same XCD, same round,
    // same pace -- so that an operand line comes from HBM once and is hit in L2 once (measured: L2 misses halved).
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int tile = slot & 3, pair = (slot >> 2) * 8 + xcd;
    if (pair >= splits * njobs) return;
    const int split = pair % splits;
    const DwJob jb = J.job[pair / splits];
    const float* __restrict__ A = jb.A;
    const float* __restrict__ X = jb.X;
    float* __restrict__ Cpart = jb.C;
    float* __restrict__ bpart = jb.bp;
    const int lda = jb.lda, ldx = jb.ldx, N = jb.N, K = jb.K;
    const int tiles_n = N / DW_TN;
    const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
    if (tile_k >= K / TK) return;
    extern __shared__ __attribute__((aligned(16))) char dws[];
    constexpr int PA = DW_TN * DWS_STRIDE;      // one plane of dY^T
    constexpr int PX = TK * DWS_STRIDE;         // one plane of X^T
    constexpr int BUF = 3 * (PA + PX);          // one stage: 3 planes of each operand
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int wn = wave >> 1, wk = wave & 1;
    constexpr int MB = 2;              // 32 rows of dW per wave
    constexpr int KB = TK / 32;        // TK/2 columns per wave
    constexpr int NTASK = 8;           // column tasks of a thread per stage: 4 of dY, 4 of X
    const int n0 = tile_n * DW_TN, k0 = tile_k * TK;
    const long long per = (((B + splits - 1) / splits) + DWS_CH - 1) & ~(long long)(DWS_CH - 1);
    const long long s_begin = split * per < B ? split * per : B;
    const long long s_end = s_begin + per < B ? s_begin + per : B;
    // whole stages go through the pipeline; the one partial stage of the launch (the split that ends at B) is done after it
    const int nstage = (int)((s_end - s_begin) / DWS_CH);
    const bool partial = (s_end - s_begin) % DWS_CH != 0;

    f32x4 acc[MB][KB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: a thread owns row pair rp of every stage and column quad cq of both operands (as in dw_gemm_bf16_kernel)
    const int rp = lane & 15, cq = (lane >> 4) + 4 * wave;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    const float* const a_row = A + n0 + cq * 4;
    const float* const x_row = X + k0 + cq * 4;
    // Rows are clamped to the last sample instead of predicated (a predicated load is a branch, and a branch splits the
    // block the conversion is scheduled in).
    auto fetch = [&](DwpRegs& R, long long s0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            long long row = s0 + 2 * rp + r;
            row = row < B ? row : B - 1;
            R.a[r] = *(const f32x4*)(a_row + row * lda);
            R.x[r] = *(const f32x4*)(x_row + row * ldx);
        }
        __builtin_amdgcn_sched_barrier(0);      // the loads leave BEFORE the MFMA block (left alone, hipcc sinks them to its end)
    };
    auto zero_surplus = [&](DwpRegs& R, long long s0) {      // rows at and past s_end contribute nothing
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bool ok = s0 + 2 * rp + r < s_end;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                R.a[r][c] = ok ? R.a[r][c] : 0.f;
                R.x[r][c] = ok ? R.x[r][c] : 0.f;
            }
        }
    };
    // column task t of a stage: t < 4 -> dY column t of the quad (also feeds the bias sums, weight bw: 0 for a stage past
    // the range); else X
    auto task = [&](const DwpRegs& R, char* buf, int t, float bw) {
        if (t < 4) {
            stage_col(buf, PA, cq * 4 + t, rp, R.a[0][t], R.a[1][t]);
            bsum[t] = fmaf(bw, R.a[0][t] + R.a[1][t], bsum[t]);
        } else {
            stage_col(buf + 3 * PA, PX, cq * 4 + (t - 4), rp, R.x[0][t - 4], R.x[1][t - 4]);
        }
    };
    // MFMAs of the stage in `cur`, conversion of `R` into `nxt` (convert_c: a next stage exists)
    auto compute = [&](auto convert_c, const char* cur, const DwpRegs& R, char* nxt, float bw) {
        constexpr bool convert = decltype(convert_c)::value;     // compile-time: a branch would split the MFMA block
        const char* As = cur;
        const char* Xs = cur + 3 * PA;
        u32x4 b[KB][3];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[kb][pl] = *(const u32x4*)(Xs + pl * PX + (wk * (TK / 2) + kb * 16 + i) * DWS_STRIDE + g * 16);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            u32x4 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = *(const u32x4*)(As + pl * PA + (wn * 32 + mb * 16 + i) * DWS_STRIDE + g * 16);
            // six products per 16 x 16 block, smallest first; two column blocks alternate so that an accumulator is
            // touched by every second MFMA only
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int kp = 0; kp < KB; kp += 2)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[mb][kp + u] = mfma_bf16(a[pa[j]], b[kp + u][pb[j]], acc[mb][kp + u]);
            if (convert) {
#pragma unroll
                for (int t = 0; t < NTASK / MB; ++t) task(R, nxt, mb * (NTASK / MB) + t, bw);
            }
            if (HINT) {
                // one MFMA, then the conversion instructions that fit its shadow; 6 * KB MFMAs per row block
#pragma unroll
                for (int m = 0; m < 6 * KB; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (m % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
        }
    };

    // Control flow: the loop body is four unconditional stages.  A skipped fetch or a skipped conversion that flows back
    // into the loop makes hipcc's waitcnt insertion merge the two paths and drain every outstanding load before the next
    // fetch (measured on the first version: 4800 cycles per stage).  Fetches and conversions past the workgroup's range
    // therefore stay in (rows clamped to valid memory, results never multiplied, bias weight 0), and the last 0..3 stages
    // are written out behind the loop.
    DwpRegs R[4];
    char* const buf0 = dws;
    char* const buf1 = dws + BUF;
    auto at = [&](int st) { return s_begin + (long long)st * DWS_CH; };
    if (nstage > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch(R[j], at(j));
#pragma unroll
        for (int t = 0; t < NTASK; ++t) task(R[0], buf0, t, 1.f);
        __syncthreads();
        int st = 0;
        for (; st + 4 <= nstage; st += 4) {
            const float bw_last = st + 4 < nstage ? 1.f : 0.f;      // the stage converted by j = 3 may lie past the range
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fetch(R[j], at(st + j + 4));
                compute(std::true_type{}, j % 2 ? buf1 : buf0, R[(j + 1) % 4], j % 2 ? buf0 : buf1, j == 3 ? bw_last : 1.f);
                __syncthreads();
            }
        }
        const int left = nstage - st;     // 0..3 stages: buf0 holds stage st, R[1..3] the rows of the following ones
        if (left >= 2) compute(std::true_type{}, buf0, R[1], buf1, 1.f);
        else if (left == 1) compute(std::false_type{}, buf0, R[1], buf1, 0.f);
        if (left >= 2) {
            __syncthreads();
            if (left == 3) compute(std::true_type{}, buf1, R[2], buf0, 1.f);
            else compute(std::false_type{}, buf1, R[2], buf0, 0.f);
        }
        if (left == 3) {
            __syncthreads();
            compute(std::false_type{}, buf0, R[3], buf1, 0.f);
        }
    }
    if (partial) {      // wave-uniform, at most one workgroup per tile: plain fetch -> convert -> MFMAs
        __syncthreads();
        fetch(R[0], at(nstage));
        zero_surplus(R[0], at(nstage));
#pragma unroll
        for (int t = 0; t < NTASK; ++t) task(R[0], buf0, t, 1.f);
        __syncthreads();
        compute(std::false_type{}, buf0, R[1], buf1, 0.f);
    }
    float* C = Cpart + (long long)split * N * K;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + mb * 16 + 4 * g + r;
                const int k = k0 + wk * (TK / 2) + kb * 16 + i;
                C[(long long)n * K + k] = acc[mb][kb][r];
            }
    if (bpart && tile_k == 0) {
        // column sums of dY: the 16 lanes of a lane group hold the 16 row pairs of the same columns
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = bsum[c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (rp == 0) bpart[(long long)split * N + n0 + cq * 4 + c] = v;
        }
    }
}

// dW for Linears with <= 3 output rows: Cpart[split][3][K], bpart[split][3]; up to four of them per launch (blockIdx.y)
struct DwSmallJob {
    const float* A; int lda, n_true;
    const float* X; int ldx, K;
    float* Cpart; float* bpart;
};
struct DwSmallJobs { DwSmallJob job[4]; };

__global__ __launch_bounds__(256) void dw_small_kernel(DwSmallJobs J, long long B, int splits) {
    const DwSmallJob& j = J.job[blockIdx.y];
    const float* __restrict__ A = j.A;
    const float* __restrict__ X = j.X;
    const int lda = j.lda, n_true = j.n_true, ldx = j.ldx, K = j.K;
    const int k = threadIdx.x;
    const int split = blockIdx.x;
    const long long per = (B + splits - 1) / splits;
    const long long s_begin = split * per;
    const long long s_end = s_begin + per < B ? s_begin + per : B;
    float acc[3] = {0.f, 0.f, 0.f}, bs[3] = {0.f, 0.f, 0.f};
#pragma unroll 8
    for (long long s = s_begin; s < s_end; ++s) {
        const float x = k < K ? X[s * ldx + k] : 0.f;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const float a = n < n_true ? A[s * lda + n] : 0.f;
            acc[n] += a * x;
            bs[n] += a;
        }
    }
    if (k < K) {
#pragma unroll
        for (int n = 0; n < 3; ++n) j.Cpart[((long long)split * 3 + n) * K + k] = acc[n];
    }
    if (k < 3) j.bpart[(long long)split * 3 + k] = bs[k];
}

// ---- assemble parameter gradients
struct DwSource {
    const float* part;   // [splits][N][K]
    int K;               // row length of the partial
    int N;               // rows of the partial (128/256, or 3 for the small kernel)
    int splits;          // sample-range splits this partial (and the layer's bias partial) was computed with
};
struct DwLayer {
    DwSource src[2];     // columns [0, split_col) from src[0], the rest from src[1]
    int split_col;       // = in_features when there is a single source
    int kind0;           // column map of src[0]: 0 identity, 1 xyz-encoding pair order
    const float* bpart;  // [splits][N]
    int out_f, in_f;
    float* d_w;
    float* d_b;
};
struct DwFinishArgs {
    DwLayer layer[16];   // out_f == 0: layer not touched by this pass
    int accumulate;      // 0: overwrite the gradient tensors, 1: add to them (second-order pass)
    int encpos[64];      // logical encoding column -> position in the saved pair-ordered encoding
};

__global__ void dw_finish_kernel(DwFinishArgs F) {
    const int L = blockIdx.y;
    const DwLayer& ly = F.layer[L];
    const int nw = ly.out_f * ly.in_f;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nw) {
        const int n = e / ly.in_f, c = e % ly.in_f;
        const bool first = c < ly.split_col;
        const DwSource& s = first ? ly.src[0] : ly.src[1];
        int col = first ? c : c - ly.split_col;
        if (first && ly.kind0 == 1) col = F.encpos[c];
        float v = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < s.splits; ++sp) v += s.part[((long long)sp * s.N + n) * s.K + col];   // unrolled: loads in flight
        ly.d_w[e] = F.accumulate ? ly.d_w[e] + v : v;
    } else if (e < nw + ly.out_f && ly.bpart) {
        const int n = e - nw;
        const int NB = ly.src[0].N;
        float v = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < ly.src[0].splits; ++sp) v += ly.bpart[(long long)sp * NB + n];
        ly.d_b[n] = F.accumulate ? ly.d_b[n] + v : v;
    }
}


}  // namespace mnrf

// ====================================================================== driver

namespace mnrf {

// MNRF_DW=fp32 selects the v_mfma_f32_16x16x4_f32 kernel (bit-for-bit fp32 fmaf chains, one launch per GEMM); default:
// bf16 x 6, batched per TK
static bool dw_fp32() {
    static const bool v = [] { const char* e = getenv("MNRF_DW"); return e && e[0] == 'f'; }();
    return v;
}

// Split counts: enough workgroups to fill 256 CUs even for a 1024-ray training batch.
int dw_splits(long long B) {          // MFMA GEMMs, batched launches: >= 512 samples per workgroup, <= 64 splits: 11.4 ms/step (48: 11.9, 32: 11.7, 96 x 256: 11.6; un-batched fp32 path: 256 / 256)
    static const int cap = [] { const char* e = getenv("MNRF_DW_SPLITS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 512 ? v : (dw_fp32() ? 256 : 64); }();
    static const int per = [] { const char* e = getenv("MNRF_DW_PER"); const int v = e ? atoi(e) : 0; return v >= 32 ? v : (dw_fp32() ? 256 : 512); }();
    long long s = (B + per - 1) / per;
    return (int)(s < 1 ? 1 : (s > cap ? cap : s));
}
// Pipelined kernels run ONE workgroup per CU, so a launch costs whole rounds of 256 workgroups: pick the split count
// (<= dw_splits(B), the workspace is sized for that) that minimises rounds x (stages per workgroup + fill/epilogue).
static int dw_splits_rounds(long long B, int tiles) {
    const int cap = dw_splits(B);
    int best = 1;
    long long best_cost = -1;
    for (int sp = 1; sp <= cap; ++sp) {
        const long long wg = (long long)sp * tiles, rounds = (wg + 255) / 256;
        const long long stages = ((B + sp - 1) / sp + DWS_CH - 1) / DWS_CH;
        const long long cost = rounds * (stages + 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}
int dw_small_splits(long long B) {    // 1-/3-row Linears: HBM-bound streaming, 128 samples per workgroup
    long long s = (B + 127) / 128;
    return (int)(s < 1 ? 1 : (s > 2048 ? 2048 : s));
}

// floats of partial results per split (see the job list in launch_dw)
constexpr long long DW_PER_SPLIT = 256LL * 64 * 2 + 256LL * 256 * 8 + 128LL * 256 * 3 + 128LL * 32 +
                                   /* bias partials */ 256LL * 9 + 128LL * 3;
constexpr long long DW_PER_SMALL_SPLIT = 3LL * 256 + 3LL * 128 * 3 + 3LL * 4;

long long dw_workspace_floats(long long B) {
    return (long long)dw_splits(B) * DW_PER_SPLIT + (long long)dw_small_splits(B) * DW_PER_SMALL_SPLIT;
}

// MNRF_DW_PIPE: 1 (default since round 4) = pipelined kernel for the 128-wide tiles, 0 = the phased kernels everywhere.  Round 2
// measured the two equal on the first-order GEMMs; those run from operand planes now (mnrf_dwp.hip), what is left on this route
// is the second-order pass of TotalLoss, and there the pipelined kernel measured 8.26-8.40 against 8.35-8.87 ms per step in
// alternating runs (half the HBM traffic: see the kernel's header);
// MNRF_DW_HINT=0 drops the scheduling-group hints of the pipelined kernel
static int dw_pipe() {
    static const int v = [] { const char* e = getenv("MNRF_DW_PIPE"); return e ? atoi(e) : 1; }();
    return v;
}
static bool dw_hint() {
    static const bool v = [] { const char* e = getenv("MNRF_DW_HINT"); return !(e && e[0] == '0'); }();
    return v;
}

struct DwBatch {
    DwJobs j128, j64, j32;
    int n128 = 0, n64 = 0, n32 = 0;
    long long B;
    int splits;          // of the phased kernels (and of the fp32 path)
    int splits_p;        // of the pipelined kernels
    hipStream_t s;
    template <int TK>
    static bool pipelined() { return !dw_fp32() && TK == DWP_TK && dw_pipe() >= 1; }
    template <int TK>
    int splits_of() const { return pipelined<TK>() ? splits_p : splits; }
    template <int TK>
    void launch(const DwJobs& J, int n) {
        if (n) hipLaunchKernelGGL((dw_gemm_bf16_kernel<TK>), dim3(4, splits, n), dim3(256), 3 * (DW_TN + TK) * DWS_STRIDE, s, J, B, splits);
    }
    void launch_pipelined(const DwJobs& J, int n) {
        if (!n) return;
        constexpr int lds = 2 * 3 * (DW_TN + DWP_TK) * DWS_STRIDE;
        static const bool once = [] {
            (void)hipFuncSetAttribute((const void*)dw_gemm_bf16p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)dw_gemm_bf16p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            return true;
        }();
        (void)once;
        const int grid = 32 * ((splits_p * n + 7) / 8);      // 8 XCDs x 4 tiles per group of 8 (job, split) pairs
        if (dw_hint()) hipLaunchKernelGGL((dw_gemm_bf16p_kernel<true>), dim3(grid), dim3(64 * DWP_WAVES), lds, s, J, B, splits_p, n);
        else hipLaunchKernelGGL((dw_gemm_bf16p_kernel<false>), dim3(grid), dim3(64 * DWP_WAVES), lds, s, J, B, splits_p, n);
    }
    void flush() {
        if (pipelined<128>()) launch_pipelined(j128, n128);
        else launch<128>(j128, n128);
        launch<64>(j64, n64);
        launch<32>(j32, n32);
        n128 = n64 = n32 = 0;
    }
};

template <int TK>
static void gemm(DwBatch& bt, const float* A, int lda, int N, const float* X, int ldx, int K, float* C, float* bp) {
    if (dw_fp32()) {
        dim3 grid(N / DW_TN, K / TK, bt.splits);
        hipLaunchKernelGGL((dw_gemm_kernel<TK>), grid, dim3(256), 0, bt.s, A, lda, X, ldx, bt.B, bt.splits, N, K, C, bp);
        return;
    }
    const DwJob jb{A, X, C, bp, lda, ldx, N, K};
    if (TK == 128) bt.j128.job[bt.n128++] = jb;
    else if (TK == 64) bt.j64.job[bt.n64++] = jb;
    else bt.j32.job[bt.n32++] = jb;
}

int launch_dw(const float* save_x, const float* dY, const float* g_sigma, long long B, float* ws, float* const* d_params,
              int accumulate, hipStream_t s) {
    const int splits = dw_splits(B);
    const int ssplits = dw_small_splits(B);
    DwBatch bt;
    bt.B = B; bt.splits = splits; bt.s = s;
    bt.splits_p = dw_splits_rounds(B, 8 * 4 + 3 * 2);      // 8 GEMMs of four 128 x 128 tiles, 3 of two
    const int s128 = bt.splits_of<128>(), s64 = bt.splits_of<64>();
    float* p = ws;
    auto take = [&](long long n) { float* r = p; p += n * splits; return r; };      // sized for the larger split count
    auto take_small = [&](long long n) { float* r = p; p += n * ssplits; return r; };
    auto X = [&](int sec) { return save_x + (long long)sec * B; };
    auto Y = [&](int sec) { return dY + (long long)sec * B; };
    DwFinishArgs F;
    F.accumulate = accumulate;
    for (int e = 0; e < 64; ++e) F.encpos[e] = 0;
    for (int gq = 0; gq < 4; ++gq)
        for (int t = 0; t < 16; ++t) {
            const int c = enc_col(t, gq);
            if (c >= 0) F.encpos[c] = 16 * (t >> 2) + 4 * gq + (t & 3);
        }
    auto layer = [&](int L, int out_f, int in_f, DwSource s0, DwSource s1, int split_col, int kind0, const float* bp) {
        DwLayer& ly = F.layer[L];
        ly.src[0] = s0; ly.src[1] = s1; ly.split_col = split_col; ly.kind0 = kind0; ly.bpart = bp;
        ly.out_f = out_f; ly.in_f = in_f; ly.d_w = d_params[2 * L]; ly.d_b = d_params[2 * L + 1];
    };
    // ---- trunk
    for (int i = 0; i < 8; ++i) {
        float* bp = take(256);
        if (i == 0) {
            float* c = take(256 * 64);
            gemm<64>(bt, Y(DY_L), 256, 256, X(SEC_ENC), 64, 64, c, bp);
            layer(0, 256, 63, DwSource{c, 64, 256, s64}, DwSource{c, 64, 256, s64}, 63, 1, bp);
        } else if (i == 4) {
            float* ce = take(256 * 64);
            float* ch = take(256 * 256);
            gemm<64>(bt, Y(DY_L + 256 * 4), 256, 256, X(SEC_ENC), 64, 64, ce, bp);
            gemm<128>(bt, Y(DY_L + 256 * 4), 256, 256, X(SEC_H + 256 * 3), 256, 256, ch, nullptr);
            layer(4, 256, 319, DwSource{ce, 64, 256, s64}, DwSource{ch, 256, 256, s128}, 63, 1, bp);
        } else {
            float* c = take(256 * 256);
            gemm<128>(bt, Y(DY_L + 256 * i), 256, 256, X(SEC_H + 256 * (i - 1)), 256, 256, c, bp);
            layer(i, 256, 256, DwSource{c, 256, 256, s128}, DwSource{c, 256, 256, s128}, 256, 0, bp);
        }
    }
    const float* h8 = X(SEC_H + 256 * 7);
    {   // xyz_encoding_final (L = 8)
        float* bp = take(256);
        float* c = take(256 * 256);
        gemm<128>(bt, Y(DY_FIN), 256, 256, h8, 256, 256, c, bp);
        layer(8, 256, 256, DwSource{c, 256, 256, s128}, DwSource{c, 256, 256, s128}, 256, 0, bp);
    }
    {   // dir_encoding.0 (L = 9): columns [0,256) from final, [256,283) from the view encoding
        float* bp = take(128);
        float* cf = take(128 * 256);
        float* cd = take(128 * 32);
        gemm<128>(bt, Y(DY_DIR), 128, 128, X(SEC_FIN), 256, 256, cf, bp);
        gemm<32>(bt, Y(DY_DIR), 128, 128, X(SEC_DIRE), 32, 32, cd, nullptr);
        layer(9, 128, 283, DwSource{cf, 256, 128, s128}, DwSource{cd, 32, 128, splits}, 256, 0, bp);
    }
    DwSmallJobs SJ;
    int n_small = 0;
    auto small = [&](int L, const float* A, int lda, int n_true, const float* Xp, int K) {
        float* bp = take_small(3);
        float* c = take_small(3 * K);
        SJ.job[n_small++] = DwSmallJob{A, lda, n_true, Xp, K, K, c, bp};       // one launch for the four of them below
        layer(L, n_true, K, DwSource{c, K, 3, ssplits}, DwSource{c, K, 3, ssplits}, K, 0, bp);
    };
    small(10, g_sigma, 1, 1, h8, 256);                          // sigma
    small(11, Y(DY_RGB), 16, 3, X(SEC_HD), 128);                // rgb.0
    {   // normal_net.0 (L = 12)
        float* bp = take(128);
        float* c = take(128 * 256);
        gemm<128>(bt, Y(DY_NRM1), 128, 128, h8, 256, 256, c, bp);
        layer(12, 128, 256, DwSource{c, 256, 128, s128}, DwSource{c, 256, 128, s128}, 256, 0, bp);
    }
    small(13, Y(DY_NRM2), 16, 3, X(SEC_HN), 128);               // normal_net.1
    {   // is_mirror_net.0 (L = 14)
        float* bp = take(128);
        float* c = take(128 * 256);
        gemm<128>(bt, Y(DY_MIR1), 128, 128, h8, 256, 256, c, bp);
        layer(14, 128, 256, DwSource{c, 256, 128, s128}, DwSource{c, 256, 128, s128}, 256, 0, bp);
    }
    small(15, Y(DY_MIR2), 16, 1, X(SEC_HM), 128);               // is_mirror_net.2
    if (p - ws > dw_workspace_floats(B)) return -1;
    hipLaunchKernelGGL(dw_small_kernel, dim3(ssplits, n_small), dim3(256), 0, s, SJ, B, ssplits);
    bt.flush();
    // largest layer: 256 x 319 + 256 elements
    hipLaunchKernelGGL(dw_finish_kernel, dim3((256 * 319 + 256 + 255) / 256, 16), dim3(256), 0, s, F);
    return 0;
}


// ---- second-order pass: dW_i += b_i^T a'_{i-1} for the 8 trunk layers, dw_sigma += sum a'_8
constexpr long long DW2_PER_SPLIT = 256LL * 64 * 2 + 256LL * 256 * 7;

long long dw2_workspace_floats(long long B) {
    return (long long)dw_splits(B) * DW2_PER_SPLIT + (long long)dw_small_splits(B) * (3LL * 256 + 3) + 16;
}

int launch_dw2(const float* so, long long B, float* ws, float* const* d_params, hipStream_t s) {
    const int splits = dw_splits(B);
    const int ssplits = dw_small_splits(B);
    DwBatch bt;
    bt.B = B; bt.splits = splits; bt.s = s;
    bt.splits_p = dw_splits_rounds(B, 7 * 4);
    const int s128 = bt.splits_of<128>(), s64 = bt.splits_of<64>();
    float* p = ws;
    auto take = [&](long long n) { float* r = p; p += n * splits; return r; };
    auto sec = [&](int off) { return so + (long long)off * B; };
    DwFinishArgs F;
    F.accumulate = 1;
    for (int L = 0; L < 16; ++L) { F.layer[L].out_f = 0; F.layer[L].in_f = 0; F.layer[L].bpart = nullptr; }
    for (int e = 0; e < 64; ++e) F.encpos[e] = 0;
    for (int gq = 0; gq < 4; ++gq)
        for (int t = 0; t < 16; ++t) {
            const int c = enc_col(t, gq);
            if (c >= 0) F.encpos[c] = 16 * (t >> 2) + 4 * gq + (t & 3);
        }
    auto layer = [&](int L, int out_f, int in_f, DwSource s0, DwSource s1, int split_col, int kind0) {
        DwLayer& ly = F.layer[L];
        ly.src[0] = s0; ly.src[1] = s1; ly.split_col = split_col; ly.kind0 = kind0; ly.bpart = nullptr;
        ly.out_f = out_f; ly.in_f = in_f; ly.d_w = d_params[2 * L]; ly.d_b = d_params[2 * L + 1];
    };
    for (int i = 0; i < 8; ++i) {
        const float* b = sec(BS_L + 256 * i);
        if (i == 0) {
            float* c = take(256 * 64);
            gemm<64>(bt, b, 256, 256, sec(TA_ENC), 64, 64, c, nullptr);
            layer(0, 256, 63, DwSource{c, 64, 256, s64}, DwSource{c, 64, 256, s64}, 63, 1);
        } else if (i == 4) {
            float* ce = take(256 * 64);
            float* ch = take(256 * 256);
            gemm<64>(bt, b, 256, 256, sec(TA_ENC), 64, 64, ce, nullptr);
            gemm<128>(bt, b, 256, 256, sec(TA_H + 256 * 3), 256, 256, ch, nullptr);
            layer(4, 256, 319, DwSource{ce, 64, 256, s64}, DwSource{ch, 256, 256, s128}, 63, 1);
        } else {
            float* c = take(256 * 256);
            gemm<128>(bt, b, 256, 256, sec(TA_H + 256 * (i - 1)), 256, 256, c, nullptr);
            layer(i, 256, 256, DwSource{c, 256, 256, s128}, DwSource{c, 256, 256, s128}, 256, 0);
        }
    }
    {   // sigma.weight: column sums of the masked tangent of h8 (A = a single 1.0 read with stride 0)
        float* one = p; p += 16;
        float* bp = p; p += 3LL * ssplits;
        float* c = p; p += 3LL * 256 * ssplits;
        static const float h1 = 1.f;
        (void)hipMemcpyAsync(one, &h1, sizeof(float), hipMemcpyHostToDevice, s);
        DwSmallJobs SJ;
        SJ.job[0] = DwSmallJob{one, 0, 1, sec(TA_H + 256 * 7), 256, 256, c, bp};
        hipLaunchKernelGGL(dw_small_kernel, dim3(ssplits, 1), dim3(256), 0, s, SJ, B, ssplits);
        layer(10, 1, 256, DwSource{c, 256, 3, ssplits}, DwSource{c, 256, 3, ssplits}, 256, 0);
    }
    if (p - ws > dw2_workspace_floats(B)) return -1;
    bt.flush();
    hipLaunchKernelGGL(dw_finish_kernel, dim3((256 * 319 + 256 + 255) / 256, 16), dim3(256), 0, s, F);
    return 0;
}

}  // namespace mnrf
