// fp32 MFMA GEMM for gfx950 (CDNA4) -- interface.
//
// One templated kernel family covers the three contractions of the DNN hot path
// (reference: neuralNetworks/classifiers/layer.py:52 `tf.matmul(inputs, weights) + biases`
// and its tf.gradients, neuralNetworks/trainer.py:155):
//
//   NN  C[M,N]  = A[M,K] . B[K,N] (+bias)      forward affine      (A: activations, B: W[d_in,d_out])
//   NT  C[M,N]  = A[M,K] . B[N,K]^T            dA = dZ . W^T       (B: W[d_in,d_out] read as [N,K])
//   TN  C[M,N] (+)= A[K,M]^T . B[K,N]          dW (+)= A^T . dZ    (accumulates into the gradient sum G)
//
// All matrices are row-major fp32 with a leading dimension that is a multiple of 4 floats and
// zero-filled padding columns (16-byte rows => every global access is a dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfk {

enum GemmLayout : int { GEMM_NN = 0, GEMM_NT = 1, GEMM_TN = 2 };

// Epilogue flags
enum : int {
  EPI_BIAS = 1,        // C += bias[col]
  EPI_ACCUM = 2,       // C = C_old + result
  EPI_RELU = 4,        // C = max(C, 0)            (non-BN forward with relu)
  EPI_COLSTATS = 8,    // also emit, per BM-row tile, each column's mean and sum of squared deviations of the
                       // stored values (the batch-norm statistics of the layer, Chan-mergeable):
                       // stats[(0 * tiles_m + tile_m) * ldc + col] = mean, stats[(1 * tiles_m + tile_m) * ldc + col] = M2
  EPI_EVAL_ACT = 32,   // evaluation-mode hidden layer in ONE launch (decoder.py:36-44, trainer.py:77-79 with
                       // is_training=False): C = f(((result + bias) - moving_mean) * rsqrt(moving_var + eps) + beta)
                       // -- the same operations, in the same order, as EPI_BIAS followed by bn_stats_eval and
                       // act_forward.  act_mean = moving mean (nullptr: no batch norm), act_rstd = moving VARIANCE,
                       // act_beta = beta, bn_eps, act_nonlin.  Requires EPI_BIAS.
  EPI_DACT = 16,       // C = du = result * f'(act_a) (derivative through the layer output act_a = f(u)), and per
                       // BM-row tile the column sums batch-norm's backward needs:
                       // stats[(0 * stats_stride + tile_m) * ldc + col] = sum du,
                       // stats[(1 * stats_stride + tile_m) * ldc + col] = sum du * (act_z - mean) * rstd
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // [N] when EPI_BIAS
  float* stats;       // [2, tiles_m, ldc] when EPI_COLSTATS; [2, stats_stride, ldc] when EPI_DACT
  // EPI_DACT: the layer whose output gradient this GEMM produces (all [M, ldc] / [N])
  const float* act_a;     // layer output a = f(u)
  const float* act_z;     // pre-batch-norm affine output
  const float* act_mean;  // batch mean / rstd of act_z's columns
  const float* act_rstd;
  int act_nonlin;         // TFK_NONLIN_*
  int stats_stride;
  int M, N, K;
  int lda, ldb, ldc;
  int epi;
  const float* act_beta = nullptr;  // EPI_EVAL_ACT
  float bn_eps = 0.f;
  // EPI_DACT with dropout behind a ReLU: a = relu(u) * keep_mask / keep, so d a / d u = (a > 0) / keep -- the mask
  // need not be regenerated.  act_scale = 1 / keep (1 without dropout); only valid with act_nonlin == relu.
  float act_scale = 1.f;
  float act_keep = 1.f;   // EPI_DACT: keep_prob (1 without dropout); with act_beta set and a ReLU chain the epilogue takes
                          // the normalised pre-activation from the layer output (a * keep - beta) and does not read act_z
  // Split-K (optional; epi 0 / EPI_ACCUM only): when the output has too few tiles to fill the chip and K is long
  // -- the weight gradient of a narrow layer over many frames -- the contraction is cut into chunks that run as
  // extra blocks into partial results in `splitk_ws`, summed in chunk order by a second kernel (deterministic).
  float* splitk_ws = nullptr;
  size_t splitk_ws_floats = 0;
  // EPI_COLSTATS of a STACKED pass (several micro-batches behind each other, each padded to a multiple of every tile
  // height): row_vend[m / 64] = the row where the valid rows of the segment that holds row m END -- a tile's statistics
  // count rows < row_vend[m0 / 64] only (nullptr: rows < M, one segment).
  const int* row_vend = nullptr;
  // filled by gemm_f32() for the kernel
  int nsplit = 1, ksplit = 0;
  size_t split_stride = 0;
};

// Tile configurations (compile-time instantiated); index = config id.
// (sN = LDS ring slots, pN = global-load prefetch distance in tiles)
//   0: 128x128 block, 4 waves of 64x64, s2 p1     5: 256x128 block, 8 waves of 64x64, s2 p1
//   1: 128x64  block, 4 waves of 64x32, s3 p2     6:  64x128 block, 8 waves of 32x32, s3 p2
//   2:  64x128 block, 4 waves of 32x64, s3 p2     7: 128x64  block, 8 waves of 32x32, s3 p2
//   3:  64x64  block, 4 waves of 32x32, s3 p2     8:  64x64  block, 4 waves of 32x32, s3 p1
//   4: 128x128 block, 8 waves of 64x32, s3 p1     9:  64x64  block, 4 waves of 32x32, 4-slot ring filled by LDS-DMA
// The heuristic uses 0 (>= 512 tiles of 128x128) and 9 (everything smaller; 3, its register-staged twin, with
// TFK_GEMM_DMA=0); the rest are kept for the sweep tool (tools/gemm_sweep.py) that produced
// profiles/r01_gemm_sweep_*.txt.
//  10: 128x64 block, 4 waves of 64x32 / 11: 64x128, 4 waves of 32x64 / 12: 128x128, 4 waves of 64x64 -- all with
//      the 4-slot LDS-DMA ring (round-2 experiment: larger wave tiles on the DMA path)
constexpr int kNumGemmConfigs = 13;

// cfg < 0 => heuristic choice. Returns hipError_t as int.
int gemm_f32(GemmLayout layout, const GemmArgs& args, int cfg, hipStream_t stream);

// An NT contraction (epi 0 / EPI_DACT) and a TN contraction (epi 0 / EPI_ACCUM) that do not depend on each other,
// in ONE launch (see gemm_f32_dual_kernel).  Returns -1 when the pair is not eligible (other tile configuration,
// split-K): the caller then launches them one after the other.
int gemm_f32_dual(const GemmArgs& nt, const GemmArgs& tn, hipStream_t stream);

int gemm_f32_splitk_min_k();  // shortest contraction the split-K path cuts (env TFK_SPLITK_MIN_K, default 2048)

// Heuristic used when cfg < 0 (exposed for tests / the sweep tool).
int gemm_f32_pick_config(GemmLayout layout, int M, int N, int K);

// Override the heuristic globally (env TFK_GEMM_CFG or the sweep tool); -1 restores it.
void gemm_f32_force_config(int cfg);

const char* gemm_f32_config_name(int cfg);
int gemm_f32_config_bm(int cfg);  // rows of a block tile (= rows per EPI_COLSTATS chunk)

}  // namespace tfk
