// mnrf_layout.h -- the packed weight image shared by the packer and the field kernel.
//
// The field kernel evaluates every Linear of MirrorNeRF (models/mirror_nerf.py:59-99) in
// the TRANSPOSED form  Out^T[N x M] = W[N x K] . In^T[K x M]  with v_mfma_f32_16x16x4_f32:
//   A operand (one VGPR): lane l holds W[n = 16*nb + (l&15)][k-slot (l>>4)]      (weights)
//   B operand (one VGPR): lane l holds In^T[k-slot (l>>4)][sample (l&15)]        (activations)
//   C/D (4 VGPRs):        lane l, reg r holds Out^T[n = 16*nb + 4*(l>>4) + r][sample (l&15)]
// The contraction order is free, so k-step t of a 256-wide input uses, for lane group
// g = l>>4, the logical column  kmap(t,g) = 16*(t>>2) + 4*g + (t&3).  With that choice
// accumulator register r of block nb of one layer IS the B operand of k-step 4*nb+r of the
// next layer: activations never leave their lane between layers (no LDS round trip, no
// shuffles), and only the weights stream.
//
// A "tile" is the A operand of 4 consecutive k-steps for one 16-row block: 64 lanes x 4
// floats = 1 KiB, read with one conflict-free ds_read_b128 per lane.  Float j of lane l of
// tile (tq, nb) is  W[16*nb + (l&15)][col(4*tq + j, l>>4)].  Tiles are stored in exactly the
// order the kernel consumes them (stage by stage, k-block by k-block, row block by row
// block), so the global->LDS stream is one linear, fully coalesced read.
#pragma once
#include <stdint.h>

namespace mnrf {

constexpr int W = 256;          // hidden width (mirror_nerf.py:43)
constexpr int ENC_XYZ = 63;     // 3 + 3*2*10
constexpr int ENC_DIR = 27;     // 3 + 3*2*4
constexpr int NFREQ_XYZ = 10;
constexpr int NFREQ_DIR = 4;

constexpr int TILE_FLOATS = 256;
constexpr int TILE_BYTES = 1024;
constexpr int PAD_TILES = 16;    // every part of a weight stream is padded to a multiple of this many tiles (the LDS
                                 // staging chunk of a kernel tuning must divide it; 32-tile chunks measured 1 % SLOWER)

// how the 4*ntq k-steps of a part map to columns of the nn.Linear weight
enum PartKind : int {
    KIND_H = 0,    // 16*(t>>2) + 4*g + (t&3)                       (hidden activations)
    KIND_ENC = 1,  // xyz encoding in (sin,cos) pairs: P = 8*g + (t>>1), s = t&1   (see enc_col)
    KIND_DIR = 2,  // dir encoding padded to 32: e = 16*(t>>2) + 4*g + (t&3), e >= 27 -> zero
};

// Column of the 63-wide xyz encoding held by lane group g at k-step t of an ENC part.
// Pair P < 30 is frequency f = P/3, axis a = P%3: sin at 3+6f+a, cos at 6+6f+a
// (channel order of Embedding.forward, mirror_nerf.py:31-38); P = 30 -> (x, y); P = 31 -> (z, pad).
__host__ __device__ inline int enc_col(int t, int g) {
    const int P = 8 * g + (t >> 1);
    const int s = t & 1;
    if (P < 30) return 3 + 6 * (P / 3) + 3 * s + (P % 3);
    if (P == 30) return s;
    return s == 0 ? 2 : -1;
}

struct Part {
    int param;      // index into the state_dict-ordered parameter list (weight tensor)
    int n_true;     // rows of the nn.Linear weight (out features)
    int ld;         // in features (row stride of the weight)
    int ntq;        // k-blocks of 16 in this part
    int nb;         // 16-row blocks of the stage output
    int col_off;    // first weight column of this part
    int kind;       // PartKind
    int tile0;      // first tile of the part inside its stream
};

__host__ __device__ constexpr int padded_tiles(int n) { return (n + PAD_TILES - 1) / PAD_TILES * PAD_TILES; }

// ---- forward stream -------------------------------------------------------------------
// stage:       L1   L2..L4  L5(enc,h)  L6..L8  SIG | NRM1 NRM2 MIR1 MIR2 FIN  DIR(h,dir)  RGB
// (the heads that only read geo_feat come first so that geo_feat dies before the colour branch)
constexpr int N_FWD_PARTS = 18;
constexpr int N_BWD_PARTS = 9;
constexpr int FWD_TILES_SIGMA = 64 + 3 * 256 + 320 + 3 * 256 + 16;               // 1936
constexpr int FWD_TILES = FWD_TILES_SIGMA + 128 + 16 + 128 + 16 + 256 + 144 + 16;   // 2640 (3-row heads: 8 tiles + 8 pad)
// ---- backward (density-gradient) stream: A = W_i^T for i = 8..1 ------------------------
constexpr int BWD_TILES = 3 * 256 + 320 + 3 * 256 + 64;                           // 1920

// bias block (floats): per stage, padded to the stage's 16*nb rows; then w_sigma (256)
constexpr int BIAS_L = 0;             // 8 x 256
constexpr int BIAS_SIG = 2048;        // 16
constexpr int BIAS_FIN = 2064;        // 256
constexpr int BIAS_DIR = 2320;        // 128
constexpr int BIAS_RGB = 2448;        // 16
constexpr int BIAS_NRM1 = 2464;       // 128
constexpr int BIAS_NRM2 = 2592;       // 16
constexpr int BIAS_MIR1 = 2608;       // 128
constexpr int BIAS_MIR2 = 2736;       // 16
constexpr int BIAS_WSIG = 2752;       // 256: sigma.weight, seed of the density gradient
constexpr int BIAS_FLOATS = 3072;     // padded to 12 KiB

// ---- head-backward stream (training): A = W^T of rgb, dir_encoding (final | view columns),
//      xyz_encoding_final, normal_net.1, normal_net.0, is_mirror_net.2, is_mirror_net.0
constexpr int N_HBWD_PARTS = 9;   // 8 used (+1 spare slot)
constexpr int HBWD_TILES = 16 + 128 + 16 + 256 + 16 + 128 + 16 + 128;                 // 704

constexpr int64_t OFF_FWD = 0;
constexpr int64_t OFF_BIAS = (int64_t)FWD_TILES * TILE_FLOATS;
constexpr int64_t OFF_BWD = OFF_BIAS + BIAS_FLOATS;
constexpr int64_t OFF_HBWD = OFF_BWD + (int64_t)BWD_TILES * TILE_FLOATS;
constexpr int64_t PACKED_F32_FLOATS = OFF_HBWD + (int64_t)HBWD_TILES * TILE_FLOATS;

// ---- split-f16 streams (mnrf_field_split.inc): every fp32 weight w is stored as the f16 pair
//      hi = f16(w), lo = f16(w - hi) (22 significand bits together; f16 subnormals are honoured by the
//      gfx950 MFMA, measured in scripts/exp_f16split.hip) and a Linear is evaluated as
//      hi.hi + lo.hi + hi.lo on v_mfma_f32_16x16x32_f16 with fp32 accumulation.
//      One MFMA contracts 32 columns = two fp32 k-blocks: half j of lane l of "pair" (T, nb) is float
//      (j&3) of lane l of the fp32 tile (2T + (j>>2), nb) -- the fp32 B operands 8T..8T+7 of a lane,
//      in order, are the 8 halves of its f16 B operand, so activations still never leave their lane.
//      A pair = [hi tile 1 KiB][lo tile 1 KiB].  Parts follow each other WITHOUT padding (the kernels' LDS-DMA schedule
//      is one piece per GEMM unit over one continuous stream, whatever the chunk size; mnrf_field_split.inc); only
//      the end of a stream is padded to SPLIT_END_PAD pairs.  Positions of the parts (in pairs):
//      forward  L1 0|16, L2 32|96, L3 160|224, L4 288|352, L5 416 (enc) 432 (h) | 496 (enc) 512 (h), L6 576|640, L7 704|768,
//               L8 832|896 (each trunk layer in two halves of 8 row blocks), sigma 960, normal_net.0 968, normal_net.1 1032,
//               is_mirror_net.0 1036, is_mirror_net.2 1100, xyz_encoding_final 1104, dir_encoding 1232 (final) 1296 (view),
//               rgb 1304, end 1308
//      trunk^T  L8 0, L7 128, L6 256, L5 384 (encoding rows) 416 (hidden rows), L4 544, L3 672, L2 800, L1 928, end 960
//      heads^T  rgb 0, dir (final columns) 8, dir (view columns) 72, final 80, normal_net.1 208, normal_net.0 216,
//               is_mirror_net.2 280, is_mirror_net.0 288, end 352
constexpr int PAIR_BYTES = 2048;
constexpr int SPLIT_END_PAD = 16;
__host__ __device__ constexpr int padded_pairs(int n) { return (n + SPLIT_END_PAD - 1) / SPLIT_END_PAD * SPLIT_END_PAD; }
constexpr int SPLIT_FWD_USED_SIGMA = 960 + 8;
constexpr int SPLIT_FWD_USED = SPLIT_FWD_USED_SIGMA + 64 + 4 + 64 + 4 + 128 + 64 + 8 + 4;            // 1308
constexpr int SPLIT_FWD_PAIRS = padded_pairs(SPLIT_FWD_USED);                                        // 1312
constexpr int SPLIT_BWD_PAIRS = 3 * 128 + 32 + 128 + 3 * 128 + 32;                                  // 960
// head-backward parts (training): a contraction over 16 rows (ntq = 1) fills half of a 32-wide MFMA step, the rest is zero
constexpr int SPLIT_HBWD_PAIRS = padded_pairs(8 + 64 + 8 + 128 + 8 + 64 + 8 + 64);                  // 352
constexpr int64_t OFF_SPLIT_FWD = PACKED_F32_FLOATS;
constexpr int64_t OFF_SPLIT_BWD = OFF_SPLIT_FWD + (int64_t)SPLIT_FWD_PAIRS * (PAIR_BYTES / 4);
constexpr int64_t OFF_SPLIT_HBWD = OFF_SPLIT_BWD + (int64_t)SPLIT_BWD_PAIRS * (PAIR_BYTES / 4);
// ---- forward stream of the 32x32x16 tuning (mnrf_field_split32.inc; inference only).  One MFMA there contracts 16 columns
//      for a block of 32 rows: a pair = [hi tile][lo tile] of a 32-row x 16-column block, half e (0..7) of lane l =
//      W[32*nb + (l&31)][col32(T, l>>5, e)].  The accumulator of a 32x32 MFMA holds, in register r of lane half h = l>>5,
//      row 8*(r>>2) + 4*h + (r&3) of its block, so registers 8c..8c+7 of block nb ARE the lane's 8 halves of k-step
//      T = 2*nb + c of the next layer -- col32_h(T, h, e) = 16*T + 8*(e>>2) + 4*h + (e&3) -- and activations stay in
//      their lanes exactly as in the 16x16 layout.  Same part sequence as the 16x16 stream (trunk layers in two halves of
//      four 32-row blocks); 1- and 3-row heads are padded to 32 rows.  Positions (pairs): trunk as above (L1 0|16 ...
//      L8 832|896), sigma 960, normal_net.0 976, normal_net.1 1040, is_mirror_net.0 1048, is_mirror_net.2 1112,
//      xyz_encoding_final 1120, dir_encoding 1248 (final) 1312 (view), rgb 1320, end 1328.
constexpr int SPLIT32_FWD_USED_SIGMA = 960 + 16;
constexpr int SPLIT32_FWD_USED = SPLIT32_FWD_USED_SIGMA + 64 + 8 + 64 + 8 + 128 + 64 + 8 + 8;      // 1328
constexpr int SPLIT32_FWD_PAIRS = padded_pairs(SPLIT32_FWD_USED);
constexpr int64_t OFF_SPLIT32_FWD = OFF_SPLIT_HBWD + (int64_t)SPLIT_HBWD_PAIRS * (PAIR_BYTES / 4);
// xyz encoding column held by lane half h at k-step T (0..3), half e of a 32x32x16 ENC part: pair P = 16*h + 4*T + (e>>1)
__host__ __device__ inline int enc_col32(int T, int h, int e) {
    const int P = 16 * h + 4 * T + (e >> 1);
    const int s = e & 1;
    if (P < 30) return 3 + 6 * (P / 3) + 3 * s + (P % 3);
    if (P == 30) return s;
    return s == 0 ? 2 : -1;
}
// tail pad: the split kernel's static LDS-DMA schedule reads two chunks (of up to 32 KiB) past the end of a stream
constexpr int64_t SPLIT_TAIL_FLOATS = 2 * 16 * (PAIR_BYTES / 4);
constexpr int64_t PACKED_FLOATS = OFF_SPLIT32_FWD + (int64_t)SPLIT32_FWD_PAIRS * (PAIR_BYTES / 4) + SPLIT_TAIL_FLOATS;
// The last words of the image (inside the tail pad, whose contents no kernel consumes) are CALLER-OWNED device state of the
// launches that use this image: [PACKED_FLOATS - 1] the range-guard word (mnrf.h), before it TQ_PAIRS {next, done} counter
// pairs of the dynamic tile queue (mnrf_field_split3.hip).  mnrf_pack_weights zeroes all of them.
constexpr int TQ_PAIRS = 8;
constexpr int64_t OFF_TILE_QUEUE = PACKED_FLOATS - 1 - 2 * TQ_PAIRS;
// ... and before those one {value, done} pair for single-launch grid reductions over an evaluation of this model (round 5: the
// seed / tangent maxima of the training backward, mnrf_dwp.hip): zero between launches like the tile-queue pairs -- the last
// workgroup out hands the value over and resets both words -- so a reduction needs no zero-fill launch in front of it
constexpr int64_t OFF_REDUCE_PAIR = OFF_TILE_QUEUE - 2;
constexpr int DEVICE_STATE_WORDS = 2 + 2 * TQ_PAIRS + 1;      // what mnrf_pack_weights zeroes, from OFF_REDUCE_PAIR on

// ---- activations saved by the training forward, [section][sample][width], B-form column order
constexpr int SEC_ENC = 0;            // 64   xyz encoding in (sin,cos)-pair order (enc_col)
constexpr int SEC_H = 64;             // 8 x 256  h1..h8 (post-ReLU)
constexpr int SEC_FIN = 2112;         // 256  xyz_encoding_final output
constexpr int SEC_DIRE = 2368;        // 32   view encoding padded
constexpr int SEC_HD = 2400;          // 128  dir_encoding output (post-ReLU)
constexpr int SEC_HN = 2528;          // 128  normal_net.0 output
constexpr int SEC_HM = 2656;          // 128  is_mirror_net.0 output (post-LeakyReLU)
constexpr int SAVE_FLOATS = 2784;     // per sample
constexpr int N_MASKS = 10;           // relu masks of L1..L8, dir_encoding, sign mask of is_mirror_net.0
// ---- pre-activation gradients written by the backward kernel, same convention
constexpr int DY_L = 0;               // 8 x 256
constexpr int DY_FIN = 2048;          // 256
constexpr int DY_DIR = 2304;          // 128
constexpr int DY_NRM1 = 2432;         // 128
constexpr int DY_MIR1 = 2560;         // 128
constexpr int DY_RGB = 2688;          // 16 (3 used)
constexpr int DY_NRM2 = 2704;         // 16 (3 used)
constexpr int DY_MIR2 = 2720;         // 16 (1 used)
constexpr int DY_FLOATS = 2736;       // per sample
// ---- second-order pass (gradient through the density-gradient normal): tangents and masked
//      density-gradient signals, [section][sample][width]
constexpr int TA_ENC = 0;             // 64   tangent of the xyz encoding in direction J^
constexpr int TA_H = 64;              // 8 x 256  masked tangents of h1..h8
constexpr int BS_L = 2112;            // 8 x 256  b_i = (dsigma/dh_i) * relu'_i  of layers 1..8
constexpr int SO_FLOATS = 4160;       // per sample
constexpr int TRUNK_FWD_TILES = 64 + 3 * 256 + 320 + 3 * 256;   // forward stream prefix L1..L8

static_assert(FWD_TILES % PAD_TILES == 0 && FWD_TILES_SIGMA % PAD_TILES == 0, "chunking");
static_assert(BWD_TILES % PAD_TILES == 0 && HBWD_TILES % PAD_TILES == 0, "chunking");

// ---- part tables of the three fp32 streams (tile0 = first tile of the part inside its stream)
struct PartTable {
    Part fwd[N_FWD_PARTS];
    Part bwd[N_BWD_PARTS];
    Part hbwd[N_HBWD_PARTS];
};

__host__ __device__ inline void build_parts(PartTable& T) {
    int n = 0, tile = 0;
    auto add = [&](Part* arr, int& cnt, int param, int n_true, int ld, int ntq, int nb, int col_off, int kind) {
        arr[cnt] = Part{param, n_true, ld, ntq, nb, col_off, kind, tile};
        tile += padded_tiles(ntq * nb);
        cnt++;
    };
    add(T.fwd, n, 0, 256, 63, 4, 16, 0, KIND_ENC);                             // L1
    for (int i = 1; i < 4; ++i) add(T.fwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // L2..L4
    add(T.fwd, n, 8, 256, 319, 4, 16, 0, KIND_ENC);                            // L5 encoding columns
    add(T.fwd, n, 8, 256, 319, 16, 16, 63, KIND_H);                            // L5 hidden columns
    for (int i = 5; i < 8; ++i) add(T.fwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // L6..L8
    add(T.fwd, n, 20, 1, 256, 16, 1, 0, KIND_H);                               // sigma
    add(T.fwd, n, 24, 128, 256, 16, 8, 0, KIND_H);                             // normal_net.0
    add(T.fwd, n, 26, 3, 128, 8, 1, 0, KIND_H);                                // normal_net.1
    add(T.fwd, n, 28, 128, 256, 16, 8, 0, KIND_H);                             // is_mirror_net.0
    add(T.fwd, n, 30, 1, 128, 8, 1, 0, KIND_H);                                // is_mirror_net.2
    add(T.fwd, n, 16, 256, 256, 16, 16, 0, KIND_H);                            // xyz_encoding_final
    add(T.fwd, n, 18, 128, 283, 16, 8, 0, KIND_H);                             // dir_encoding: final part
    add(T.fwd, n, 18, 128, 283, 2, 8, 256, KIND_DIR);                          // dir_encoding: view part
    add(T.fwd, n, 22, 3, 128, 8, 1, 0, KIND_H);                                // rgb
    // backward: A = W_i^T, rows = input columns of layer i, contraction over its 256 outputs
    n = 0;
    tile = 0;
    for (int i = 7; i >= 5; --i) add(T.bwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // layers 8,7,6
    add(T.bwd, n, 8, 256, 319, 16, 4, 0, KIND_ENC);                             // layer 5: 4 encoding row blocks
    add(T.bwd, n, 8, 256, 319, 16, 16, ENC_XYZ, KIND_H);                        // layer 5: 16 hidden row blocks
    for (int i = 3; i >= 1; --i) add(T.bwd, n, 2 * i, 256, 256, 16, 16, 0, KIND_H);  // layers 4,3,2
    add(T.bwd, n, 0, 256, 63, 16, 4, 0, KIND_ENC);                              // layer 1: 4 encoding row blocks
    // head backward (training): n_true = rows of W = contraction length, nb = 16-blocks of W's columns
    n = 0;
    tile = 0;
    add(T.hbwd, n, 22, 3, 128, 1, 8, 0, KIND_H);        // rgb^T
    add(T.hbwd, n, 18, 128, 283, 8, 16, 0, KIND_H);     // dir_encoding^T, xyz_encoding_final columns
    add(T.hbwd, n, 18, 128, 283, 8, 2, 256, KIND_DIR);  // dir_encoding^T, view-encoding columns
    add(T.hbwd, n, 16, 256, 256, 16, 16, 0, KIND_H);    // xyz_encoding_final^T
    add(T.hbwd, n, 26, 3, 128, 1, 8, 0, KIND_H);        // normal_net.1^T
    add(T.hbwd, n, 24, 128, 256, 8, 16, 0, KIND_H);     // normal_net.0^T
    add(T.hbwd, n, 30, 1, 128, 1, 8, 0, KIND_H);        // is_mirror_net.2^T
    add(T.hbwd, n, 28, 128, 256, 8, 16, 0, KIND_H);     // is_mirror_net.0^T
}

}  // namespace mnrf
