// bf16 MFMA GEMM for gfx950 (CDNA4) -- interface.  Mixed-precision mode of the DNN hot path (BASELINE cfg3 / cfg4):
// the three contractions of gemm_f32.h with bf16 operands, fp32 accumulation and fp32 results.
//
//   NN  C[M,N]  = A[M,K] . B[K,N] (+bias)      A: bf16 activations (k-contiguous), B: bf16 weight shadow [d_in,d_out]
//   NT  C[M,N]  = A[M,K] . B[N,K]^T            A: bf16 dZ, B: the same weight shadow read as [N,K]
//   TN  C[M,N] (+)= A[K,M]^T . B[K,N]          A: bf16 layer input, B: bf16 dZ -- both k-STRIDED in memory
//
// bf16 matrices are row-major with a leading dimension that is a multiple of 8 elements (16-byte rows) and
// zero-filled padding columns; C, bias and the epilogue operands are the fp32 buffers of gemm_f32.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"

namespace tfk {

typedef uint16_t bf16_t;  // storage type at the interfaces (bit pattern of a bfloat16)

struct GemmArgsB {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  const float* bias;      // EPI_BIAS
  float* stats;           // EPI_COLSTATS: [2, tiles_m, ldc];  EPI_DACT: [2, stats_stride, ldc]
  const float* act_a;     // EPI_DACT operands (see gemm_f32.h)
  const float* act_z;
  const float* act_mean;  // EPI_DACT: batch mean;  EPI_EVAL_ACT: moving mean (nullptr: no batch norm)
  const float* act_rstd;  // EPI_DACT: batch rstd;  EPI_EVAL_ACT: moving VARIANCE
  int act_nonlin;
  int stats_stride;
  int M, N, K;
  int lda, ldb, ldc;      // lda / ldb in bf16 elements, ldc in floats
  int epi;                // EPI_* of gemm_f32.h: 0, BIAS, BIAS|COLSTATS, BIAS|EVAL_ACT (NN); 0, DACT (NT); 0, ACCUM (TN)
  const float* act_beta;  // EPI_EVAL_ACT
  float bn_eps;
  bf16_t* C_twin;         // EPI_EVAL_ACT: bf16 copy of the result (the next layer's operand); nullable
  int ldct;               // its leading dimension in elements (multiple of 8)
  float act_scale;        // EPI_DACT: 1 / keep_prob for a ReLU + dropout chain (see gemm_f32.h); 0 is read as 1
  float act_keep;         // EPI_DACT: keep_prob (0 is read as 1); with act_beta set, a ReLU chain does not read act_z
  const int* row_vend;    // EPI_COLSTATS of a stacked pass: see gemm_f32.h (nullptr: one segment)
  // gemm_bf16x3: A, B (and C_twin when ct_x3 is set: EPI_EVAL_ACT then writes three planes of its result) are THREE bf16 planes
  // interleaved per 32 elements of the flat index row * ld + col (x3_layout.h): 3 * rows * ld elements each
  int ct_x3;
  // gemm_bf16x3, optional: workspace of the two-way split-K form (see gemm_bf16x3_splitk_floats); ZERO before its first use,
  // left zero in its flag words by every launch.  nullptr: never split
  float* splitk_ws;
  size_t splitk_ws_floats;
  // gemm_bf16x3, optional: a word the HOST can read (mapped pinned memory) that a block sets to 1 when its split-K partner's
  // partial sums did not arrive within ~1 s (a workspace that was not zero, a partner that died): the launch's results are
  // then wrong and the owner of the word must fail the call that waits for them.  nullptr: the timeout goes unreported
  unsigned* err;
  // gemm_bf16x3 split-K hand-over: 1 = a block that KNOWS its partner runs on the same XCD (both publish HW_REG_XCC_ID when they
  // start) leaves its partial sums in that XCD's L2 (plain stores) instead of writing them through to memory; 0 = always through
  // memory.  Set by the launcher from env TFK_X3_HANDOVER (l2 | mem); correctness never depends on the placement
  int splitk_local;
  // gemm_bf16x3 split-K: ring tiles (32 k each) moved from the FIRST block's share of K to the second's (0 = equal halves).  The
  // block with the shorter share -- dispatched first as well -- then reaches the hand-over ahead of its partner, whose wait for
  // the partial sums shrinks by what it spends on the extra tiles.  Set by the launcher from env TFK_X3_KSKEW (experiments)
  int splitk_skew;
};

// Tile configurations.  0-2: register-staged ring of round 1 (64x64 / 128x64 / 128x128 per 4-wave block).
// 3-6: LDS-DMA staged, 64x64 (or 64x32) wave tiles:
//   3: 128x64  block, 4 waves (64x32 each), 5-slot ring  (one block per CU: the 1024-frame shapes)
//   4: 128x128 block, 4 waves (64x64 each), 4-slot ring
//   5: 256x128 block, 8 waves (64x64 each), 3-slot ring
//   6: 128x64  block, 4 waves (64x32 each), 3-slot ring  (two blocks per CU)
//   7: 256x128 block, 4 waves (128x64 each), 3-slot ring (round-3 experiment: a quarter fewer fragment reads per flop, one
//      wave per SIMD)
//   8: 256x256 block, 8 waves (64x128 each), 32 k per slot, 4-slot ring, PING-PONG schedule (the two waves of a SIMD run half a
//      slot apart: one multiplies out of registers, the other fetches fragments and issues its LDS-DMA pieces) -- the TN layout
//      (weight gradient) only: half the staged bytes per flop of 256x128; other layouts run 5 instead
constexpr int kNumGemmBf16Configs = 9;

// Returns hipError_t as int.
int gemm_bf16(GemmLayout layout, const GemmArgsB& args, hipStream_t stream);

// An NT contraction (epi 0 / EPI_DACT) and a TN contraction (epi 0 / EPI_ACCUM) that do not depend on each other, in ONE
// launch (gemm_bf16_dual_kernel).  Returns -1 when the pair is not eligible: the caller launches them one after the other.
int gemm_bf16_dual(const GemmArgsB& nt, const GemmArgsB& tn, hipStream_t stream);
// block geometry gemm_bf16_dual picks for an [M_nt, N_nt] and an [M_tn, N_tn] result (0: not eligible) and its tile rows
// (= rows per EPI_DACT chunk of the NT half)
int gemm_bf16_dual_config(int M_nt, int N_nt, int M_tn, int N_tn);
int gemm_bf16_dual_tile_rows(int cfg);

// An fp32 contraction EMULATED on the bf16 matrix pipe ("bf16x3").  Every operand is given as THREE bf16 planes p1, p2, p3
// with x = p1 + p2 + p3 exactly (round-to-nearest split, |p2| <= 2^-8 |x|, |p3| <= 2^-16 |x|: twin_split3 in kernels.h), interleaved
// per 32 elements (x3_layout.h: a ring slot's row segment is 192 contiguous bytes); the
// kernel accumulates the six plane products of order <= 2^-16 -- a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 -- in fp32.  Products
// of bf16 values are exact in fp32 and the three dropped pairs sum to at most 2^-23 of a product, so the result differs from the exact
// fp32 dot product only by fp32 accumulation error: measured against float64 it is CLOSER than the fp32 MFMA chain of
// gemm_f32 (tools/bf16x3_probe.py), at 6/16 of its matrix-pipe time.  Same layouts, leading dimensions and epilogues as
// gemm_bf16; blocks of 128 rows (= rows per EPI_COLSTATS / EPI_DACT chunk).
int gemm_bf16x3(GemmLayout layout, const GemmArgsB& args, hipStream_t stream);
constexpr int kGemmBf16x3TileRows = 128;
// Split-K.  A [1024, 2048] result has 128 tiles of 128x128 for 256 CUs; 128x64 blocks fill every CU but stage 4/3 of the bytes
// per flop, and at 1024 frames the contraction is bound by exactly that (L2 -> LDS fill).  With a workspace the NN / NT
// layouts run 128x128 blocks on HALF of K each, two blocks per tile: the first to finish leaves its partial sums in the
// workspace, the second adds them to its own (a + b = b + a: the result does not depend on who was first) and runs the
// epilogue.  The TN layout splits where even 128x64 blocks leave half the chip idle (the weight gradient of a narrow layer:
// 440 x 2048 over 1024 frames).  Floats of workspace an [M, N] x K contraction needs for that (0: it would not split).
size_t gemm_bf16x3_splitk_floats(GemmLayout layout, int M, int N, int K);
// The NT (epi 0 / EPI_DACT) and TN (epi 0 / EPI_ACCUM) contractions of a layer's backward pass in ONE launch of 128x128 blocks
// (as gemm_bf16_dual; -1: not eligible -- fewer than a tile per CU between them).  Rows per EPI_DACT chunk: kGemmBf16x3TileRows.
int gemm_bf16x3_dual(const GemmArgsB& nt, const GemmArgsB& tn, hipStream_t stream);

// rows of the block tile gemm_bf16 uses for an [M, N] result (= rows per EPI_COLSTATS / EPI_DACT chunk)
int gemm_bf16_tile_rows(int M, int N);

// Override the tile heuristic (tools / tests; env TFK_BF16_CFG does the same); -1 restores it.
void gemm_bf16_force_config(int cfg);
int gemm_bf16_pick_config(int M, int N);

}  // namespace tfk
