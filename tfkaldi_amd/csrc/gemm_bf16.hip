// bf16 MFMA GEMM for gfx950 (CDNA4): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Replaces the same reference lines as gemm_f32.hip (neuralNetworks/classifiers/layer.py:52 and its tf.gradients,
// neuralNetworks/trainer.py:155) when the engine runs in mixed precision.
//
// One 64x64, 128x64 or 128x128 output tile per 256-thread block (4 waves as 2 x 2), K in steps of 64:
//   * global -> registers -> LDS through buffer resources (out-of-range chunks come back as zeros, no branches);
//     three LDS stages fed from a register ring that keeps PF tiles of loads in flight;
//   * a k-contiguous operand ([ext][k] in memory) is kept as rows of 64 + 8 bf16 (144 B: conflict-free
//     ds_read_b128) and a lane's MFMA operand -- 8 consecutive k of one row -- is ONE ds_read_b128;
//   * a k-strided operand ([k][ext] in memory: the weight matrix in the forward GEMM, BOTH operands of the
//     weight-gradient GEMM) is kept as it lies, rows of 64 + 32 bf16 (192 B), and transposed on the way out of LDS by
//     ds_read_b64_tr_b16: each 16-lane group reads a [4 k][16 ext] block and every lane receives the 4 k-values of
//     its own column, two reads per operand.  No transposed copy of weights or activations exists anywhere.
// At the sizes of this path (1024 rows per GPU) the bf16 contractions are not matrix-bound (8.6 GFLOP = 3.4 us at
// peak against ~20 MB of operands and a launch): what matters is load latency, hence the deep register ring.
#include "gemm_bf16.h"

#include <stdlib.h>

namespace tfk {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64, NT = 256, PF = 4;  // PF: tiles of global loads in flight per block (even)
constexpr int KC_LD = BK + 8;  // elements per LDS row of a k-contiguous operand (144 B)
constexpr int kOOB = (int)0x80000000;
constexpr int NUM_XCD = 8;

// EXT = tile extent along m (64 or 128) or n (64)
template <bool KC, int EXT>
struct Operand {
  static constexpr int LD = KC ? KC_LD : EXT + 32;   // k-strided: one k per row of EXT + 32 elements (192 / 320 B)
  static constexpr int SZ = (KC ? EXT : BK) * LD;    // elements per stage
  static constexpr int NCH = EXT * BK / 8 / NT;      // 16-byte chunks per thread per tile
};

template <bool KC, int EXT>
struct Loader {
  static constexpr int NCH = Operand<KC, EXT>::NCH;
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[NCH];  // byte offset inside the matrix without the k-tile term; kOOB if outside along ext
  int kidx[NCH];  // first k of the chunk inside a tile
  int kstride;    // bytes per unit of k
  int k_lim;

  // chunk c of the tile: (row, 8-element column) in the memory order of the operand
  static __device__ __forceinline__ void coords(int c, int& r, int& q) {
    if (KC) { r = c >> 3; q = (c & 7) << 3; }                       // [ext][k]: 8 chunks per 64-k row
    else { r = c / (EXT / 8); q = (c % (EXT / 8)) << 3; }           // [k][ext]: EXT / 8 chunks per k row
  }
  __device__ __forceinline__ void init(const bf16_t* base, int ld, int rows, int ext0, int ext_lim, int k_lim_,
                                       int tid) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * ld * 2, 0x00020000);
    k_lim = k_lim_;
    kstride = KC ? 2 : ld * 2;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int r, q;
      coords(tid + j * NT, r, q);
      if (KC) {
        voff[j] = (ext0 + r < ext_lim) ? ((ext0 + r) * ld + q) * 2 : kOOB;
        kidx[j] = q;
      } else {
        voff[j] = (ext0 + q < ext_lim) ? (r * ld + ext0 + q) * 2 : kOOB;
        kidx[j] = r;
      }
    }
  }
  __device__ __forceinline__ void load(u32x4 (&v)[NCH], int k0) const {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int off = (k0 + kidx[j] < k_lim) ? voff[j] : kOOB;
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, k0 * kstride, 0);
    }
  }
  __device__ __forceinline__ void store(const u32x4 (&v)[NCH], bf16_t* s, int tid) const {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int r, q;
      coords(tid + j * NT, r, q);
      *reinterpret_cast<u32x4*>(s + r * Operand<KC, EXT>::LD + q) = v[j];
    }
  }
};

// MFMA operand of k-step ks for the 32 rows (or columns) starting at ext_base; LD = LDS row length in elements.
template <bool KC, int LD>
__device__ __forceinline__ bf16x8 fragment(const bf16_t* s, int ext_base, int ks, int lane) {
  if (KC) {
    const int i = lane & 31, kb = lane >> 5;
    return *reinterpret_cast<const bf16x8*>(s + (ext_base + i) * LD + 16 * ks + 8 * kb);
  } else {
    const int kb = lane >> 5, half = (lane >> 4) & 1, j = (lane >> 2) & 3, q = lane & 3;
    const bf16_t* p = s + (16 * ks + 8 * kb + j) * LD + ext_base + 16 * half + 4 * q;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// TFK_ABLB (tools/gemm_bf16_ablate.hip only): timing-only variants with pieces of the K loop removed --
// 1 global loads, 2 LDS writes, 4 fragment reads, 8 barrier.  Results are wrong by construction.
#ifndef TFK_ABLB
#define TFK_ABLB 0
#endif

// FM / FN: 32-row / 32-column MFMA fragments per wave.  Block tile = (64 * FM) x (64 * FN), four waves as 2 x 2, wave
// tile (32 * FM) x (32 * FN).  Larger wave tiles stage fewer bytes AND fewer staging instructions per MFMA (a
// ds_write_b128 holds its wave ~13 cycles, an MFMA lasts 32): 64x64 -> 128x64 -> 128x128 as the problem allows.
template <bool A_KC, bool B_KC, int EPI, int FM, int FN>
__global__ void __launch_bounds__(NT)
gemm_bf16_kernel(GemmArgsB p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  constexpr int BM = 64 * FM, BN = 64 * FN;
  typedef Operand<A_KC, BM> OA;
  typedef Operand<B_KC, BN> OB;
  constexpr int A_SZ = OA::SZ, B_SZ = OB::SZ, STAGE = A_SZ + B_SZ;
  constexpr int KSTEPS = BK / 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware order: block b runs on XCD b % 8; each XCD takes a contiguous run of the column-major tile
  // sequence, so the tiles sharing a B panel (and neighbouring A panels) meet in one L2.
  int tm, tn;
  {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid % NUM_XCD, loc = bid / NUM_XCD;
    const int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
    const int seq = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    tm = seq % tiles_m;
    tn = seq / tiles_m;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int K8 = (p.K + 7) & ~7;
  Loader<A_KC, BM> la;
  Loader<B_KC, BN> lb;
  // k-contiguous: rows = ext, chunks valid while k < K8 (zero padding inside the row)
  // k-strided:    rows = k (valid while k < K), chunks valid while ext < ext rounded up to 8
  la.init(p.A, p.lda, A_KC ? p.M : p.K, m0, A_KC ? p.M : ((p.M + 7) & ~7), A_KC ? K8 : p.K, tid);
  lb.init(p.B, p.ldb, B_KC ? p.N : p.K, n0, B_KC ? p.N : ((p.N + 7) & ~7), B_KC ? K8 : p.K, tid);

  // a wave owns FM * FN accumulator fragments; with a single fragment the k-steps alternate between two
  // accumulators so that consecutive MFMAs never depend on each other
  constexpr int KS = (FM * FN == 1) ? 2 : 1;
  f32x16 acc[FM][FN][KS];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][q][r] = 0.f;

  // Pipeline (one barrier per K step of 64, nothing on the critical path but the MFMAs):
  //   registers: a ring of PF tiles of global loads in flight (a K step computes for a few hundred cycles per
  //              wave, a load takes over a thousand);
  //   LDS:       three stages -- while tile t is multiplied, tile t+1 (made visible by the previous barrier) is
  //              already readable and tile t+2 is written;
  //   fragments: double-buffered per 16-k step; the last step of a tile prefetches the first of the next.
  const int nk = (p.K + BK - 1) / BK;
  u32x4 ra[PF][OA::NCH], rb[PF][OB::NCH];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    la.load(ra[j], j * BK);  // tiles beyond K come back as zeros without touching memory
    lb.load(rb[j], j * BK);
  }
  bf16_t* st0 = smem;              // stage of tile t
  bf16_t* st1 = smem + STAGE;      // tile t + 1
  bf16_t* st2 = smem + 2 * STAGE;  // tile t + 2
  la.store(ra[0], st0, tid);
  lb.store(rb[0], st0 + A_SZ, tid);
  la.store(ra[1], st1, tid);
  lb.store(rb[1], st1 + A_SZ, tid);
  la.load(ra[0], PF * BK);
  lb.load(rb[0], PF * BK);
  la.load(ra[1], (PF + 1) * BK);
  lb.load(rb[1], (PF + 1) * BK);
  __syncthreads();
  bf16x8 fa[2][FM], fb[2][FN];
  auto read_frags = [&](int buf, const bf16_t* st, int ks) {
#pragma unroll
    for (int a = 0; a < FM; ++a) fa[buf][a] = fragment<A_KC, OA::LD>(st, wm * 32 * FM + a * 32, ks, lane);
#pragma unroll
    for (int b = 0; b < FN; ++b) fb[buf][b] = fragment<B_KC, OB::LD>(st + A_SZ, wn * 32 * FN + b * 32, ks, lane);
  };
  read_frags(0, st0, 0);
#pragma unroll 1
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      // No guard on kt < nk: the trip count is rounded up to a multiple of PF and the surplus steps multiply zero
      // tiles.  A guard would make the compiler assume that earlier steps of the unrolled body may not have issued
      // their loads, and it then waits for (nearly) ALL outstanding loads before each LDS write.
      const int kt = kt0 + j;
      const int s2 = (j + 2) % PF;  // register set holding tile kt + 2
      if (!(TFK_ABLB & 2)) {
        la.store(ra[s2], st2, tid);
        lb.store(rb[s2], st2 + A_SZ, tid);
      }
      if (!(TFK_ABLB & 1)) {
        la.load(ra[s2], (kt + 2 + PF) * BK);
        lb.load(rb[s2], (kt + 2 + PF) * BK);
      }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int cur = ks & 1;  // KSTEPS is even: every tile starts in buffer 0
        if (!(TFK_ABLB & 4)) {
          if (ks + 1 < KSTEPS) read_frags(cur ^ 1, st0, ks + 1);
          else read_frags(cur ^ 1, st1, 0);
        }
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
            acc[a][b][ks % KS] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][a], fb[cur][b], acc[a][b][ks % KS], 0, 0, 0);
      }
      if (!(TFK_ABLB & 8)) __syncthreads();
      bf16_t* t = st0; st0 = st1; st1 = st2; st2 = t;
    }
  }
  if constexpr (KS == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][0][r] += acc[0][0][1][r];
  }

  // ---- epilogue: D reg r of lane (i, h) is row (r&3) + 8*(r>>2) + 4*h, column i of a 32x32 fragment ----
  float* red = reinterpret_cast<float*>(smem);  // [2][2 waves along m][BN]; the K loop ended behind a barrier
  auto row_of = [&](int a, int r) { return m0 + wm * 32 * FM + a * 32 + 4 * h + (r & 3) + 8 * (r >> 2); };
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const int col = n0 + wn * 32 * FN + b * 32 + i;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : p.N - 1;
    const int cidx = wn * 32 * FN + b * 32 + i;
    if constexpr ((EPI & EPI_BIAS) != 0) {
      const float bv = p.bias[colc];
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][0][r] += bv;
    }
    if constexpr ((EPI & EPI_COLSTATS) != 0) {
      // per-tile batch-norm statistics (mean, sum of squared deviations), two-pass over the accumulators
      const int n_tile = min(BM, p.M - m0);
      float cmean = 0.f;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[a][b][0][r];
            if (row_of(a, r) < p.M) s += pass == 0 ? v : (v - cmean) * (v - cmean);
          }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wm * BN + cidx] = s;
        __syncthreads();
        const float t = red[cidx] + red[BN + cidx];
        if (pass == 0) {
          cmean = t / (float)n_tile;
        } else if (wm == 0 && h == 0 && col_ok) {
          p.stats[((size_t)0 * tiles_m + tm) * p.ldc + col] = cmean;
          p.stats[((size_t)1 * tiles_m + tm) * p.ldc + col] = t;
        }
        __syncthreads();
      }
    }
    if constexpr ((EPI & EPI_DACT) != 0) {
      // da -> du = da * f'(a) in the accumulators + the two column sums of batch-norm's backward for this tile
      const float mu = p.act_mean[colc], rsd = p.act_rstd[colc];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        float av[16], zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row_of(a, r), p.M - 1);
          av[r] = p.act_a[(size_t)row * p.ldc + colc];
          zv[r] = p.act_z[(size_t)row * p.ldc + colc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d1;
          switch (p.act_nonlin) {
            case 0: d1 = av[r] > 0.f ? 1.f : 0.f; break;
            case 1: d1 = av[r] * (1.f - av[r]); break;
            case 2: d1 = 1.f - av[r] * av[r]; break;
            default: d1 = 1.f;
          }
          const float du = acc[a][b][0][r] * d1;
          acc[a][b][0][r] = du;
          if (row_of(a, r) < p.M) {
            s1 += du;
            s2 += du * (zv[r] - mu) * rsd;
          }
        }
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0) {
        red[(0 * 2 + wm) * BN + cidx] = s1;
        red[(1 * 2 + wm) * BN + cidx] = s2;
      }
      __syncthreads();
      if (wm == 0 && h == 0 && col_ok) {
        p.stats[((size_t)0 * p.stats_stride + tm) * p.ldc + col] = red[(0 * 2 + 0) * BN + cidx] + red[(0 * 2 + 1) * BN + cidx];
        p.stats[((size_t)1 * p.stats_stride + tm) * p.ldc + col] = red[(1 * 2 + 0) * BN + cidx] + red[(1 * 2 + 1) * BN + cidx];
      }
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      float old[16];
      if constexpr ((EPI & EPI_ACCUM) != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = p.C[(size_t)min(row_of(a, r), p.M - 1) * p.ldc + colc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_of(a, r);
        float v = acc[a][b][0][r];
        if (EPI & EPI_ACCUM) v += old[r];
        if (col_ok && row < p.M) p.C[(size_t)row * p.ldc + col] = v;
      }
    }
  }
}

// tile shape: the largest of 128x128 / 128x64 / 64x64 that still gives (nearly) every CU a block
int pick_tile(int M, int N) {  // returns 10 * FM + FN
  static const int forced = [] { const char* q = getenv("TFK_BF16_TILE"); return q ? atoi(q) : 0; }();
  if (forced == 11 || forced == 21 || forced == 22) return forced;
  const int m128 = (M + 127) / 128;
  if (m128 * ((N + 127) / 128) >= 192) return 22;
  if (m128 * ((N + 63) / 64) >= 192) return 21;
  return 11;
}

template <bool A_KC, bool B_KC, int EPI, int FM, int FN>
int launch_tile(const GemmArgsB& p, hipStream_t stream) {
  constexpr int BM = 64 * FM, BN = 64 * FN;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)3 * (Operand<A_KC, BM>::SZ + Operand<B_KC, BN>::SZ) * sizeof(bf16_t);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<A_KC, B_KC, EPI, FM, FN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, EPI, FM, FN>), dim3(tiles_m * tiles_n), dim3(NT), lds, stream, p,
                     tiles_m, tiles_n);
  return (int)hipGetLastError();
}
template <bool A_KC, bool B_KC, int EPI>
int launch(const GemmArgsB& p, hipStream_t stream) {
  switch (pick_tile(p.M, p.N)) {
    case 22: return launch_tile<A_KC, B_KC, EPI, 2, 2>(p, stream);
    case 21: return launch_tile<A_KC, B_KC, EPI, 2, 1>(p, stream);
    default: return launch_tile<A_KC, B_KC, EPI, 1, 1>(p, stream);
  }
}

}  // namespace

int gemm_bf16_tile_rows(int M, int N) { return 64 * (pick_tile(M, N) / 10); }

int gemm_bf16(GemmLayout layout, const GemmArgsB& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 3)) return (int)hipErrorInvalidValue;
  switch (layout) {
    case GEMM_NN:
      switch (p.epi) {
        case 0: return launch<true, false, 0>(p, stream);
        case EPI_BIAS: return launch<true, false, EPI_BIAS>(p, stream);
        case EPI_BIAS | EPI_COLSTATS: return launch<true, false, EPI_BIAS | EPI_COLSTATS>(p, stream);
      }
      break;
    case GEMM_NT:
      if (p.epi == 0) return launch<true, true, 0>(p, stream);
      if (p.epi == EPI_DACT) return launch<true, true, EPI_DACT>(p, stream);
      break;
    case GEMM_TN:
      if (p.epi == 0) return launch<false, false, 0>(p, stream);
      if (p.epi == EPI_ACCUM) return launch<false, false, EPI_ACCUM>(p, stream);
      break;
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace tfk
