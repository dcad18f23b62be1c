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
  const float* act_mean;
  const float* act_rstd;
  int act_nonlin;
  int stats_stride;
  int M, N, K;
  int lda, ldb, ldc;      // lda / ldb in bf16 elements, ldc in floats
  int epi;                // EPI_* of gemm_f32.h: 0, BIAS, BIAS|COLSTATS (NN); 0, DACT (NT); 0, ACCUM (TN)
};

// Returns hipError_t as int.
int gemm_bf16(GemmLayout layout, const GemmArgsB& args, hipStream_t stream);

// rows of the block tile gemm_bf16 uses for an [M, N] result (= rows per EPI_COLSTATS / EPI_DACT chunk)
int gemm_bf16_tile_rows(int M, int N);

}  // namespace tfk
