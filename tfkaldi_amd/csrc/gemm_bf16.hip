// bf16 MFMA GEMM for gfx950 (CDNA4): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Replaces the same reference lines as gemm_f32.hip (neuralNetworks/classifiers/layer.py:52 and its tf.gradients,
// neuralNetworks/trainer.py:155) when the engine runs in mixed precision (BASELINE cfg3 / cfg4 arithmetic).
//
// Two kernel families share one epilogue:
//
// (1) LDS-DMA staged (configs 3-8, the default).  A bf16 contraction does 16x the flops of the fp32 one per byte
//     of operand, so what bounds it at the sizes of this path is how fast operand tiles reach LDS (~38 B/clk/CU
//     from L2, profiles/r01_gemm_dma.txt), i.e. the BLOCK tile: bytes per flop fall as 1/BM + 1/BN.  Blocks are
//     128x64 (the 1024-frame shapes: 256 tiles = one per CU), 128x128 or 256x128 (4 / 8 waves, each wave a
//     64x64 -- or 64x32 -- patch = 4 (2) accumulator fragments of 32x32), K in steps of 64:
//       * `buffer_load_dwordx4 ... lds` moves 64 x 16 B per wave-instruction from per-lane global addresses into one
//         contiguous KiB of LDS: no staging registers, no ds_write pass.  The LDS image is lane-linear, so the
//         bank-conflict permutation is applied on the SOURCE side and undone by the reader:
//           - k-contiguous operand ([ext][64 k], 128-byte rows): k-chunk c of row r sits at chunk c ^ ((r >> 1) & 7);
//             a lane's MFMA operand (8 consecutive k of one row) is ONE ds_read_b128, and the 16-lane service groups
//             of ds_read_b128 touch 16 distinct 16-byte slots;
//           - k-strided operand ([64 k][ext]: the weight matrix in forward, BOTH operands of the weight gradient)
//             stays in memory order and is transposed on the way out of LDS by ds_read_b64_tr_b16 (a 16-lane group
//             reads a [4 k][16 ext] block); ext-chunk c of k-row r sits at chunk c ^ ((r & 3) << 2) (rows of >= 256 B)
//             or c ^ (((r >> 1) & 1) << 2) (128-byte rows), which spreads the four rows of a group over all 64 banks.
//       * ring of NS stages; iteration t issues the pieces of tile t+NS-1 (spread behind the MFMAs of the four
//         16-k steps) into the slot tile t-1 left, multiplies tile t, then waits with a COUNTED s_waitcnt vmcnt for
//         its own pieces of tile t+1 only -- the younger tiles stay in flight across the barrier, so the fill path
//         never drains.  hipcc does not model these loads (inline asm): completion is that wait + s_barrier.
//         (round 3) the K loop is rotated by one 16-k step -- the barrier sits before a tile's LAST step, whose MFMAs
//         cover the LDS round trip of the next tile's first fragments.
//       * (round 3) the weight gradient at BASELINE cfg4's size runs on 256x256 blocks with 32-k ring slots (both of its
//         operands are k-strided, so a slot may be any multiple of 16 k deep) on a PING-PONG schedule: see SCHED 2.
// (2) register-staged ring of round 1 (configs 0-2; kept for A/B runs: tools/gemm_bf16_sweep.py).
#include "gemm_bf16.h"
#include "x3_layout.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace tfk {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int kOOB = (int)0x80000000;
constexpr int NUM_XCD = 8;

// XCD-aware order: block b runs on XCD b % 8 and each XCD takes a contiguous run of a GROUPED tile sequence (groups of
// `group_rows` tile rows, column-major inside a group), i.e. a patch of about group_rows x (run / group_rows) tiles that
// share A row-panels and B column-panels in that XCD's L2.  The host picks group_rows so that the patch is square in
// BYTES (group_rows * BM ~ columns * BN), which minimises what the eight private L2s fetch from the fabric.
__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int group_rows, int bid, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid % NUM_XCD, loc = bid / NUM_XCD;
  const int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
  const int seq = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int per_group = group_rows * tiles_n;
  const int grp = seq / per_group;
  const int first_m = grp * group_rows;
  const int gsize = min(group_rows, tiles_m - first_m);
  const int within = seq - grp * per_group;
  tm = first_m + within % gsize;
  tn = within / gsize;
}

__device__ __forceinline__ uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

// ---- epilogue shared by both families --------------------------------------------------------------------------
// Wave (wm, wn) of a WAVES_M x WAVES_N arrangement owns the 32x32 fragments acc[a][b] at rows
// m0 + (wm * FM + a) * 32, columns n0 + (wn * FN + b) * 32.  D reg r of lane (i, h) is row (r&3) + 8*(r>>2) + 4*h,
// column i of its fragment.  `red` = LDS scratch (the K loop ended behind a barrier), >= 2 * WAVES_M * BN floats.
template <int EPI, int WAVES_M, int WAVES_N, int FM, int FN>
__device__ __forceinline__ void epilogue(const GemmArgsB& p, f32x16 (&acc)[FM][FN], int tiles_m, int tm, int m0, int n0,
                                         int wm, int wn, int i, int h, float* red) {
  constexpr int BM = WAVES_M * FM * 32, BN = WAVES_N * FN * 32;
  auto row_of = [&](int a, int r) { return m0 + (wm * FM + a) * 32 + 4 * h + (r & 3) + 8 * (r >> 2); };
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const int cidx = (wn * FN + b) * 32 + i;
    const int col = n0 + cidx;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : p.N - 1;
    if constexpr ((EPI & EPI_BIAS) != 0) {
      const float bv = p.bias[colc];
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += bv;
    }
    if constexpr ((EPI & EPI_EVAL_ACT) != 0) {
      // evaluation-mode batch norm + nonlinearity: the operations of bn_stats_eval + act_forward (kernels.hip),
      // in their order
      const bool bn = p.act_mean != nullptr;
      const float mu = bn ? p.act_mean[colc] : 0.f;
      const float rs = bn ? rsqrtf(p.act_rstd[colc] + p.bn_eps) : 1.f;
      const float be = bn ? p.act_beta[colc] : 0.f;
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float u = acc[a][b][r];
          if (bn) u = (u - mu) * rs + be;
          switch (p.act_nonlin) {
            case 0: u = fmaxf(u, 0.f); break;
            case 1: u = 1.f / (1.f + expf(-u)); break;
            case 2: u = tanhf(u); break;
            default: break;
          }
          acc[a][b][r] = u;
        }
    }
    if constexpr ((EPI & EPI_COLSTATS) != 0) {
      // per-tile batch-norm statistics (mean, sum of squared deviations), two-pass over the accumulators
      const int vlim = p.row_vend ? min(p.M, p.row_vend[m0 >> 6]) : p.M;  // (stacked pass: rows of this tile's segment)
      const int n_tile = max(1, min(BM, vlim - m0));
      float cmean = 0.f;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[a][b][r];
            if (row_of(a, r) < vlim) s += pass == 0 ? v : (v - cmean) * (v - cmean);
          }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wm * BN + cidx] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) t += red[w * BN + cidx];
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
      const float dscale = p.act_scale > 0.f ? p.act_scale : 1.f;
      // ReLU chains: the normalised pre-activation where du != 0 is a * keep - beta -- z is not read (gemm_f32.hip)
      const bool from_a = p.act_nonlin == 0 && p.act_beta != nullptr;
      const float be = from_a ? p.act_beta[colc] : 0.f;
      const float keep = p.act_keep > 0.f ? p.act_keep : 1.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        float av[16], zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row_of(a, r), p.M - 1);
          av[r] = p.act_a[(size_t)row * p.ldc + colc];
        }
        if (!from_a) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(row_of(a, r), p.M - 1);
            zv[r] = p.act_z[(size_t)row * p.ldc + colc];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d1;
          switch (p.act_nonlin) {
            case 0: d1 = av[r] > 0.f ? dscale : 0.f; break;
            case 1: d1 = av[r] * (1.f - av[r]); break;
            case 2: d1 = 1.f - av[r] * av[r]; break;
            default: d1 = 1.f;
          }
          const float du = acc[a][b][r] * d1;
          acc[a][b][r] = du;
          if (row_of(a, r) < p.M) {
            s1 += du;
            s2 += du * (from_a ? av[r] * keep - be : (zv[r] - mu) * rsd);
          }
        }
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0) {
        red[(0 * WAVES_M + wm) * BN + cidx] = s1;
        red[(1 * WAVES_M + wm) * BN + cidx] = s2;
      }
      __syncthreads();
      if (wm == 0 && h == 0 && col_ok) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          t1 += red[(0 * WAVES_M + w) * BN + cidx];
          t2 += red[(1 * WAVES_M + w) * BN + cidx];
        }
        p.stats[((size_t)0 * p.stats_stride + tm) * p.ldc + col] = t1;
        p.stats[((size_t)1 * p.stats_stride + tm) * p.ldc + col] = t2;
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
        float v = acc[a][b][r];
        if (EPI & EPI_ACCUM) v += old[r];
        if (col_ok && row < p.M) {
          p.C[(size_t)row * p.ldc + col] = v;
          if constexpr ((EPI & EPI_EVAL_ACT) != 0) {
            if (p.C_twin) {
              if (p.ct_x3) {  // three interleaved planes whose sum is v exactly (x3_layout.h: split3, round to nearest)
                const size_t at = x3::at((size_t)row, col, p.ldct);
                uint16_t q0, q1, q2;
                x3::split3(v, q0, q1, q2);
                p.C_twin[at] = (bf16_t)q0;
                p.C_twin[at + 64] = (bf16_t)q1;
                p.C_twin[at + 128] = (bf16_t)q2;
              } else {
                p.C_twin[(size_t)row * p.ldct + col] = f2bf(v);
              }
            }
          }
        }
      }
    }
  }
}

// ================================================================================================================
// (1) LDS-DMA staged kernel
// ================================================================================================================

// ext-chunk permutation of a k-strided tile row (see the header comment)
template <int EXT>
__device__ __forceinline__ int ks_swz(int r) {
  return EXT == 64 ? (((r >> 1) & 1) << 2) : ((r & 3) << 2);
}

// One operand's share of a stage: EXT rows of BKT k (k-contiguous; BKT = 64 only) or BKT k-rows of EXT elements;
// EXT * BKT * 2 bytes either way.
// (k-contiguous images: 128-byte rows at 64 k per slot, chunk c of row r at c ^ ((r >> 1) & 7); 64-byte rows at 32 k per slot
// -- the three-plane stages of the fp32-emulating kernel -- chunk c of row r at c ^ ((r >> 2) & 3): the 16 lanes of a
// ds_read_b128 service group then touch 16 distinct 16-byte slots of a 256-byte bank row in both cases)
template <bool KC, int EXT, int NTH, int BKT = 64>
struct DmaOperand {
  static_assert(BKT == 64 || BKT == 32, "ring slot of 64 or 32 k");
  static constexpr int NP = EXT * BKT / 8 / NTH;  // 16-byte pieces per thread per tile
  static_assert((EXT * BKT / 8) % NTH == 0 && NP >= 1, "pieces per thread");
  i32x4 rsrc;
  int voff[NP];  // byte offset of the piece's source inside the matrix, k-tile term excluded; kOOB outside along ext
  int kidx[NP];  // its first k inside a tile
  int kstride;   // bytes per unit of k
  int k_lim;

  // extra_bytes: further planes of the operand behind the first one (the resource must cover them: issue() reaches
  // plane q through the scalar offset)
  __device__ __forceinline__ void init(const bf16_t* base, int ld, int rows, int ext0, int ext_lim, int k_lim_,
                                       int tid, long extra_bytes = 0) {
    const unsigned long long a = (unsigned long long)base;
    rsrc[0] = (int)(unsigned)a;
    rsrc[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    rsrc[2] = (int)(rows * ld * 2 + extra_bytes);
    rsrc[3] = 0x00020000;
    k_lim = k_lim_;
    kstride = KC ? 2 : ld * 2;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int idx = tid + j * NTH;  // chunk position inside the LDS image
      if (KC) {
        constexpr int CPR = BKT / 8;  // 16-byte chunks per row
        const int r = idx / CPR;
        const int sw = BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3);
        const int k = ((idx % CPR) ^ sw) << 3;
        const int e = ext0 + r;
        voff[j] = e < ext_lim ? (e * ld + k) * 2 : kOOB;
        kidx[j] = k;
      } else {
        constexpr int CH = EXT / 8;
        const int r = idx / CH;
        const int c = (idx % CH) ^ ks_swz<EXT>(r);
        const int e = ext0 + (c << 3);
        voff[j] = e < ext_lim ? (r * ld + e) * 2 : kOOB;
        kidx[j] = r;
      }
    }
  }
  // piece j of the tile at k0 -> image at LDS byte address `image` (out-of-range pieces land as zeros)
  __device__ __forceinline__ void issue(int j, unsigned image, int k0, int wave, int plane_bytes = 0) const {
    const int off = (k0 + kidx[j] < k_lim) ? voff[j] : kOOB;
    const unsigned dst = image + (unsigned)(wave * 64 + j * NTH) * 16u;
    const int soff = k0 * kstride + plane_bytes;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(dst), "v"(off), "s"(rsrc), "s"(soff)
                 : "memory");
  }
};

// MFMA operand fetch from the DMA image.  NF fragments of 32 rows (columns) starting at fragment index frag0.
template <bool KC, int EXT, int NF, int BKT = 64>
struct Frag {
  int off[KC ? 4 : NF];
  __device__ __forceinline__ void init(int lane, int frag0) {
    if constexpr (KC) {
      const int i = lane & 31, kb = lane >> 5, sw = BKT == 64 ? ((i >> 1) & 7) : ((i >> 2) & 3);
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) off[ks] = (frag0 * 32 + i) * (BKT * 2) + ((((2 * ks + kb) ^ sw)) << 4);
    } else {
      constexpr int ROWB = EXT * 2;
      const int kb = lane >> 5, half = (lane >> 4) & 1, j = (lane >> 2) & 3, q = lane & 3;
      const int jj = EXT == 64 ? (j >> 1) : j;
#pragma unroll
      for (int f = 0; f < NF; ++f)
        off[f] = (8 * kb + j) * ROWB + ((4 * ((frag0 + f) ^ jj) + 2 * half + (q >> 1)) << 4) + ((q & 1) << 3);
    }
  }
  // fragment f, 16-k step ks of the operand image at `img`
  __device__ __forceinline__ bf16x8 read(const char* img, int f, int ks) const {
    if constexpr (KC) {
      return *reinterpret_cast<const bf16x8*>(img + off[ks] + f * 32 * (BKT * 2));
    } else {
      constexpr int ROWB = EXT * 2;
      typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
      const char* q = img + off[f] + ks * 16 * ROWB;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * ROWB));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

// ---- the fp32-emulating contraction: tiled three-plane operands (x3_layout.h) -----------------------------------------------
// One operand's share of a ring slot: 32 k of ALL THREE planes of EXT rows (k-contiguous) or EXT columns (k-strided);
// EXT * 192 bytes either way, whole 128-byte lines of the tiled array in both cases.
template <bool KC, int EXT, int NTH>
struct DmaOperand3 {
  static constexpr int NP = EXT * 12 / NTH;  // 16-byte pieces per thread per tile (three planes)
  static_assert((EXT * 12) % NTH == 0 && NP >= 1, "pieces per thread");
  i32x4 rsrc;
  int voff[NP];  // byte offset of the piece's source inside the tiled array, k-tile term excluded; kOOB outside along ext
  int kidx[NP];  // its first k inside a tile
  int kstride;   // bytes per unit of k (k in steps of 32)
  int k_lim;

  __device__ __forceinline__ void init(const bf16_t* base, int ld, int rows, int ext0, int ext_lim, int k_lim_, int tid) {
    const unsigned long long a = (unsigned long long)base;
    rsrc[0] = (int)(unsigned)a;
    rsrc[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    rsrc[2] = (int)(x3::elems((size_t)rows, ld) * 2);
    rsrc[3] = 0x00020000;
    k_lim = k_lim_;
    kstride = KC ? 12 : ld * 6;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int n = tid + j * NTH;  // chunk position inside the LDS image
      if (KC) {
        int r, q, c;
        x3::kc_decode(n, r, q, c);
        const int e = ext0 + r;
        voff[j] = e < ext_lim ? (int)(x3::at((size_t)e, c * 8, ld) + q * 64) * 2 : kOOB;
        kidx[j] = c * 8;
      } else {
        int r, b, q, e8;
        x3::ks_decode<EXT>(n, r, b, q, e8);
        const int e = ext0 + b * 32 + e8 * 8;
        voff[j] = e < ext_lim ? (int)(x3::at((size_t)r, e, ld) + q * 64) * 2 : kOOB;
        kidx[j] = r;
      }
    }
  }
  // piece j of the tile at k0 (a multiple of 32) -> image at LDS byte address `image`
  // GUARD = false: the caller knows that the whole tile lies below k_lim (every tile but the last few of a block: the K loop
  // is peeled, see dma_tile -- twelve compare + select pairs per slot would otherwise sit between the MFMAs of a wave that
  // issues in order)
  template <bool GUARD>
  __device__ __forceinline__ void issue(int j, unsigned image, int k0, int wave) const {
    const int off = (!GUARD || k0 + kidx[j] < k_lim) ? voff[j] : kOOB;
    const unsigned dst = image + (unsigned)(wave * 64 + j * NTH) * 16u;
    const int soff = k0 * kstride;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(dst), "v"(off), "s"(rsrc), "s"(soff)
                 : "memory");
  }
};

// MFMA operand fetch from a tiled image: plane q of NF fragments of 32 rows (columns) starting at fragment frag0.
// Two per-lane byte offsets serve every fragment, plane and 16-k step: the rest of an address is an immediate.
//   k-contiguous: off[ks] (the two k-chunks a lane reads in the two steps of a slot sit at different swizzled positions);
//   k-strided:    off[parity of X], X = 3 * fragment + plane: the quadrant of (fragment, plane) for a lane whose k-row pair
//                 is odd is X ^ 1 (x3_layout.h) = X + 1 for even X, X - 1 for odd X.
template <bool KC, int EXT, int NF>
struct Frag3 {
  int off[2];
  __device__ __forceinline__ void init(int lane, int frag0) {
#pragma unroll
    for (int v = 0; v < 2; ++v) off[v] = KC ? x3::kc_lane_off(lane, frag0, v) : x3::ks_lane_off<EXT>(lane, frag0, v);
  }
  // the same offsets shifted by a constant (a ring slot's base): the reads then need no address arithmetic at all
  __device__ __forceinline__ Frag3 at(int bytes) const {
    Frag3 r;
    r.off[0] = off[0] + bytes;
    r.off[1] = off[1] + bytes;
    return r;
  }
  // fragment f, plane pl, 16-k step ks of the operand image at `img`
  __device__ __forceinline__ bf16x8 read(const char* img, int f, int ks, int pl) const {
    if constexpr (KC) {
      return *reinterpret_cast<const bf16x8*>(img + off[ks] + x3::kc_imm(f, pl));
    } else {
      typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
      const int X = 3 * f + pl;
      const char* q = img + off[X & 1];
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + x3::ks_imm<EXT>(X, ks, 0)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + x3::ks_imm<EXT>(X, ks, 1)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

// The same for the 16x16x32 MFMA shape (TFK_X3_M16, x3_layout.h): plane q of NF16 fragments of 16 rows (columns), all 32 k of a slot.
// k-contiguous: one offset register; k-strided: four (parity of X = 3 * unit + plane, parity of the fragment inside its unit).
template <bool KC, int EXT, int NF16>
struct Frag3M {
  int off[KC ? 1 : 4];
  __device__ __forceinline__ void init(int lane, int frag0) {
    if constexpr (KC) {
      off[0] = x3::kc16_lane_off(lane, frag0);
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) off[v] = x3::ks16_lane_off<EXT>(lane, frag0, v & 1, v >> 1);
    }
  }
  __device__ __forceinline__ Frag3M at(int bytes) const {
    Frag3M r;
#pragma unroll
    for (int v = 0; v < (KC ? 1 : 4); ++v) r.off[v] = off[v] + bytes;
    return r;
  }
  __device__ __forceinline__ bf16x8 read(const char* img, int f, int pl) const {
    if constexpr (KC) {
      return *reinterpret_cast<const bf16x8*>(img + off[0] + x3::kc16_imm(f, pl));
    } else {
      typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
      const int X = 3 * (f >> 1) + pl;
      const char* q = img + off[(X & 1) + 2 * (f & 1)];
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + x3::ks16_imm<EXT>(X, 0)));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + x3::ks16_imm<EXT>(X, 1)));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

// TFKB_ABL (tools/gemm_bf16_ablate.hip only): timing-only variants of the DMA kernel with pieces of the K loop removed
// -- 1 MFMAs, 2 LDS-DMA pieces, 4 fragment reads.  Results are wrong by construction.
#ifndef TFKB_ABL
#define TFKB_ABL 0
#endif

#define TFKB_WAIT_BARRIER(n) \
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" : : "n"(n) : "memory")

// BKT: k per ring slot -- 64 (four 16-k MFMA steps), or 32 (two steps) for the 256x256 block of two k-strided operands,
// whose 64-k stage (64 KB) would leave room for two slots only
// SCHED 2 = "ping-pong" (8-wave blocks): the two waves that share a SIMD run half a tile apart -- while one multiplies a whole
// ring slot out of registers (nothing but MFMAs), its partner fetches every fragment of its next slot and issues its LDS-DMA
// pieces; a block-wide barrier separates the half-steps and the two swap roles.  See the loop.
// NPL = 3: the fp32-EMULATING contraction ("bf16x3", gemm_bf16.h): every operand comes as three bf16 planes whose sum is the
// fp32 value exactly; a ring slot holds all six plane tiles of a 32-k step, and each 16-k MFMA step multiplies the six plane
// pairs of order <= 2^-16 -- (3,1) (2,2) (1,3) (2,1) (1,2) (1,1), smallest first -- into the SAME fp32 accumulators.
// KSPLIT = 2 (NPL == 3 only): two blocks per tile, each over half of K; see gemm_bf16x3_splitk_floats (gemm_bf16.h)
constexpr int kSplitFlagWords = 4096;  // head of the split-K workspace: {ticket, ready} per tile, from word kSplitXccWord on the
                                      // {XCD + 1 of the block of K half 0, of K half 1} per tile; then the partial sums
constexpr int kSplitXccWord = 2048;
// LOADERS (NPL == 3 only; 0 or WAVES_M * WAVES_N): that many EXTRA waves behind the multiplying ones do nothing but issue the
// LDS-DMA pieces ("wave specialisation").  A wave issues in order, and `buffer_load ... lds` sits in the vector-memory issue
// queue for as long as the fill path is busy (it is: the fill of a slot takes about as long as its MFMAs) -- in a wave that also
// multiplies, every one of the twelve pieces per slot is a bubble in the matrix pipe: MFMAs + fragment reads alone 34 us, with
// the pieces 47 (profiles/r04_gemm_f32x3_ablation.txt), i.e. fill and MFMA time ADD UP although neither saturates the CU.  With
// a loader wave next to every multiplying wave on its SIMD the stalls hit a wave that has nothing else to do; the multiplying
// waves' stream is fragment reads and MFMAs only.  Loaders and multipliers meet at the per-slot barrier (the loader arrives
// once its own pieces of the NEXT tile have landed); after the K loop the loaders end -- a barrier counts surviving waves only,
// so the epilogue's barriers are the multipliers' own.
template <bool A_KC, bool B_KC, int EPI, int WAVES_M, int WAVES_N, int FM, int FN, int NS, int BKT = 64, int SCHED = 0,
          int NPL = 1, int KSPLIT = 1, int LOADERS = 0>
__device__ __forceinline__ void dma_tile(const GemmArgsB& p, int tiles_m, int tiles_n, int group_rows, int bid, char* smem) {
  static_assert(LOADERS == 0 || (NPL == 3 && LOADERS == WAVES_M * WAVES_N), "loader waves: one per multiplying wave, x3 only");
  constexpr int NTH = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * FM * 32, BN = WAVES_N * FN * 32;
  // (NPL == 3: one image per operand holds its three planes, x3_layout.h)
  typedef typename std::conditional<NPL == 3, DmaOperand3<A_KC, BM, NTH>, DmaOperand<A_KC, BM, NTH, BKT>>::type OA;
  typedef typename std::conditional<NPL == 3, DmaOperand3<B_KC, BN, NTH>, DmaOperand<B_KC, BN, NTH, BKT>>::type OB;
  constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE = NPL * (A_BYTES + B_BYTES);
  constexpr int NPA = OA::NP, NPB = OB::NP, NP = OA::NP + OB::NP;
  constexpr int KSPT = BKT / 16;  // 16-k MFMA steps per ring slot
  static_assert(KSPT == 4 || KSPT == 2, "ring slot of 64 or 32 k");
  static_assert(NS >= 3 && (NS - 2) * NP + (NPL == 3 ? NP / 2 : 0) <= 63, "ring depth / vmcnt range");
  static_assert(FM * FN >= 2 || NPL == 3, "two independent accumulator chains per wave");
  static_assert(NPL == 1 || (NPL == 3 && SCHED == 0), "planes");
  static_assert(KSPLIT == 1 || (KSPLIT == 2 && NPL == 3), "split-K: the fp32-emulating contraction only");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tm, tn;
  // split-K: the two halves of a tile are neighbours in the block sequence of ONE XCD (blocks b and b + 8), so they run at the
  // same time and meet in that XCD's L2; the tile sequence is the unsplit one (tiles_m * tiles_n is a multiple of 8 here)
  const int khalf = KSPLIT == 2 ? (bid / NUM_XCD) & 1 : 0;
  const int tile_bid = KSPLIT == 2 ? (bid / NUM_XCD >> 1) * NUM_XCD + bid % NUM_XCD : bid;
  tile_of_block(tiles_m, tiles_n, group_rows, tile_bid, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  unsigned my_xcc = 0;
  if constexpr (KSPLIT == 2) {
    // where this block runs, for its partner to read when the two meet (below): a fact about THIS launch, whatever the placement
    if (p.splitk_local) {
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
      if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned*>(p.splitk_ws) + kSplitXccWord + 2 * (tm * tiles_n + tn) + khalf, my_xcc + 1,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  const int K8 = (p.K + 7) & ~7;
  const int nk_all = (p.K + BKT - 1) / BKT;
  // (split-K: the first block's share is (nk_all + 1) / 2 - splitk_skew tiles, at least one)
  const int h0 = KSPLIT == 2 ? max(1, (nk_all + 1) / 2 - p.splitk_skew) : nk_all;
  const int kbase = khalf ? h0 : 0;                                   // first ring tile of this block ...
  const int nk = KSPLIT == 2 ? (khalf ? nk_all - h0 : h0) : nk_all;  // ... and how many it multiplies
  const int k_end = (kbase + nk) * BKT;  // (tiles behind the block's share land as zeros, like the ones behind K)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
  OA la;
  OB lb;
  // k-contiguous: rows = ext, chunks valid while k < K8 (zero padding inside the row)
  // k-strided:    rows = k (valid while k < K), chunks valid while ext < ext rounded up to 8
  const int a_klim = min(A_KC ? K8 : p.K, k_end), b_klim = min(B_KC ? K8 : p.K, k_end);
  // tiles [0, nk_full) of this block lie wholly below both operands' k limits
  const int nk_full = min(a_klim, b_klim) / BKT - kbase;
  if constexpr (LOADERS > 0) {
    if (wave >= WAVES_M * WAVES_N) {
      // ---- a loader wave: tile t + NS - 1 goes out right behind barrier t - 1 (its slot's last reader has every fragment of
      // tile t - 1 in registers by then), and barrier t is entered once this wave's pieces of tile t + 1 have landed ----
      const int lwave = wave - WAVES_M * WAVES_N;
      la.init(p.A, p.lda, A_KC ? p.M : p.K, m0, A_KC ? p.M : ((p.M + 7) & ~7), a_klim, tid - NTH);
      lb.init(p.B, p.ldb, B_KC ? p.N : p.K, n0, B_KC ? p.N : ((p.N + 7) & ~7), b_klim, tid - NTH);
      auto tile = [&](auto guard, int slot, int kt) {
        if (TFKB_ABL & 2) return;
#pragma unroll
        for (int j = 0; j < NPA; ++j)
          la.template issue<decltype(guard)::value>(j, lds0 + (unsigned)(slot * STAGE), (kt + kbase) * BKT, lwave);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
          lb.template issue<decltype(guard)::value>(j, lds0 + (unsigned)(slot * STAGE + NPL * A_BYTES), (kt + kbase) * BKT, lwave);
      };
#pragma unroll
      for (int t = 0; t < NS - 1; ++t) tile(std::true_type(), t, t);
      TFKB_WAIT_BARRIER((NS - 2) * NP);
      int ws = NS - 1;
      const int n_fast = max(0, min(nk, nk_full - (NS - 1)));
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        if (kt < n_fast) tile(std::false_type(), ws, kt + NS - 1);
        else tile(std::true_type(), ws, kt + NS - 1);
        TFKB_WAIT_BARRIER((NS - 2) * NP);
        ws = ws + 1 == NS ? 0 : ws + 1;
      }
      TFKB_WAIT_BARRIER(0);
      return;
    }
  } else {
    la.init(p.A, p.lda, A_KC ? p.M : p.K, m0, A_KC ? p.M : ((p.M + 7) & ~7), a_klim, tid);
    lb.init(p.B, p.ldb, B_KC ? p.N : p.K, n0, B_KC ? p.N : ((p.N + 7) & ~7), b_klim, tid);
  }
  typename std::conditional<NPL == 3, Frag3<A_KC, BM, FM>, Frag<A_KC, BM, FM, BKT>>::type qa;
  typename std::conditional<NPL == 3, Frag3<B_KC, BN, FN>, Frag<B_KC, BN, FN, BKT>>::type qb;
  qa.init(lane, wm * FM);
  qb.init(lane, wn * FN);

  f32x16 acc[FM][FN];
  f32x16 acc2[NPL == 3 ? FM : 1][NPL == 3 ? FN : 1];  // (NPL == 3: the correction products, see mfma_step)
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  if constexpr (NPL == 3) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
  }

  // (guard: std::true_type -- a piece beyond the operand's k range lands as zeros; std::false_type, NPL == 3 only: the caller
  //  knows the tile lies inside it, see DmaOperand3::issue)
  auto piece_g = [&](auto guard, int j, int slot, int kt) {
    if (TFKB_ABL & 2) return;
    if constexpr (LOADERS > 0) {
      return;  // (the loader waves issue every piece)
    } else if constexpr (NPL == 3) {
      if (j < NPA)
        la.template issue<decltype(guard)::value>(j, lds0 + (unsigned)(slot * STAGE), (kt + kbase) * BKT, wave);
      else
        lb.template issue<decltype(guard)::value>(j - NPA, lds0 + (unsigned)(slot * STAGE + NPL * A_BYTES), (kt + kbase) * BKT, wave);
    } else {
      if (j < NPA)
        la.issue(j, lds0 + (unsigned)(slot * STAGE), (kt + kbase) * BKT, wave);
      else
        lb.issue(j - NPA, lds0 + (unsigned)(slot * STAGE + A_BYTES), (kt + kbase) * BKT, wave);
    }
  };
  auto piece = [&](int j, int slot, int kt) { piece_g(std::true_type(), j, slot, kt); };
  // prologue: tiles 0 .. NS-2 (tiles beyond K land as zeros without touching memory)
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
#pragma unroll
    for (int j = 0; j < NP; ++j) piece(j, t, t);
  // (NPL == 3 spreads a tile's pieces over TWO half-steps, see its loop: the first half of tile NS-1 goes out here)
  constexpr int NPH = NPL == 3 ? NP / 2 : 0;
#pragma unroll
  for (int j = 0; j < NPH; ++j) piece(j, NS - 1, NS - 1);
  TFKB_WAIT_BARRIER((NS - 2) * NP + NPH);

  bf16x8 fa[2][NPL][FM], fb[2][NPL][FN];
  auto read_frags = [&](int buf, const char* st, int ks) {
    if (TFKB_ABL & 4) return;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if constexpr (NPL == 3) {
#pragma unroll
        for (int a = 0; a < FM; ++a) fa[buf][pl][a] = qa.read(st, a, ks, pl);
#pragma unroll
        for (int b = 0; b < FN; ++b) fb[buf][pl][b] = qb.read(st + NPL * A_BYTES, b, ks, pl);
      } else {
#pragma unroll
        for (int a = 0; a < FM; ++a) fa[buf][pl][a] = qa.read(st, a, ks);
#pragma unroll
        for (int b = 0; b < FN; ++b) fb[buf][pl][b] = qb.read(st + A_BYTES, b, ks);
      }
    }
  };
  if (TFKB_ABL & 4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int e = 0; e < 8; ++e) fa[q][pl][a][e] = (__bf16)(float)(lane + e + pl);
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int e = 0; e < 8; ++e) fb[q][pl][b][e] = (__bf16)(float)(lane - e - pl);
      }
    }
  }
  auto mfma_step = [&](int cur) {
    if (TFKB_ABL & 1) {  // keep the fragment reads alive without the MFMAs
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int a = 0; a < FM; ++a) asm volatile("" : : "v"(fa[cur][pl][a]));
#pragma unroll
        for (int b = 0; b < FN; ++b) asm volatile("" : : "v"(fb[cur][pl][b]));
      }
    } else if constexpr (NPL == 1) {
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0][a], fb[cur][0][b], acc[a][b], 0, 0, 0);
    } else {
      // plane pairs (pa, pb), smallest products first.  The five CORRECTION products (order 2^-8 and 2^-16 of a1 b1) have an
      // accumulator of their own, added to the main one in front of the epilogue: added to the large running sum one by one
      // they would each be rounded at ITS scale (measured: rms error 1.7x the fp32 MFMA chain's; with their own accumulator
      // below it).  Consecutive MFMAs go to different accumulators.
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
            acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][PA[c]][a], fb[cur][PB[c]][b], acc2[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0][a], fb[cur][0][b], acc[a][b], 0, 0, 0);
    }
  };
  // The K loop is ROTATED by one 16-k step (round 3): the per-tile barrier sits between the third and the fourth step of a
  // tile.  The fourth step's fragments are in registers by then, so its MFMAs run right AFTER the barrier -- under them the
  // first fragments of the next tile make their LDS round trip.  Round 2 fetched those fragments after the barrier and
  // every wave of the block then waited for them at the same moment (all waves leave a barrier together): ~300 idle
  // matrix-pipe cycles per 64-k tile on the 8-wave 256x128 block, where MFMAs + fragment reads cost 53 us against 43.5 us
  // of MFMAs alone (profiles/r02_gemm_bf16_ablation.txt).
  int rs = 0, ws = NS - 1;
  if constexpr (SCHED == 2) {
    // Half-steps are numbered by the barriers between them.  Waves 0 .. 3 (group 0, one per SIMD): LOAD(0) | MUL(0) | LOAD(1) | ..;
    // waves 4 .. 7 (group 1, their SIMD partners): idle | LOAD(0) | MUL(0) | ..  LOAD(t) reads every fragment of slot t into
    // registers, issues the wave's pieces of tile t+NS-1 into the slot tile t-1 occupied (its last reader, the other group,
    // finished one half-step earlier) and waits for its own pieces of tile t+1; MUL(t) is KSPT * FM * FN MFMAs and nothing else.
    static_assert(WAVES_M * WAVES_N == 8, "two waves per SIMD");
    const int group = wave >> 2;
    bf16x8 wa[KSPT][FM], wb[KSPT][FN];
    if (group == 1) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" : : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const char* st = smem + rs * STAGE;
#pragma unroll
      for (int ks = 0; ks < KSPT; ++ks) {
#pragma unroll
        for (int a = 0; a < FM; ++a) wa[ks][a] = qa.read(st, a, ks);
#pragma unroll
        for (int b = 0; b < FN; ++b) wb[ks][b] = qb.read(st + A_BYTES, b, ks);
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) piece(j, ws, kt + NS - 1);
      __builtin_amdgcn_sched_barrier(0);
      TFKB_WAIT_BARRIER((NS - 2) * NP);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSPT; ++ks)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks][a], wb[ks][b], acc[a][b], 0, 0, 0);
      rs = rs + 1 == NS ? 0 : rs + 1;
      ws = ws + 1 == NS ? 0 : ws + 1;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" : : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (group == 0) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" : : : "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (NPL == 3 && TFK_X3_M16 != 0) {
    // The 16x16x32 shape: a ring slot is ONE 32-k step of six plane products x FM16 x FN16 MFMAs.  The products run (0,0) (0,1)
    // (0,2) | barrier | (1,0) (1,1) (2,0): every fragment register is read for the last time as early as possible, so one set of
    // fragments (+ a second B plane 0) serves -- A1 and A2 of this slot are fetched under the first half, A0, B0, B2, B1 of the
    // NEXT slot under the second, each into registers whose last use lies behind it.  The barrier sits where it sat (round 3):
    // every read of this slot has been issued in front of it, the next slot's data is complete behind it.
    static_assert(KSPT == 2, "fp32-emulating contraction: 32 k per ring slot");
    constexpr int FM16 = 2 * FM, FN16 = 2 * FN;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    Frag3M<A_KC, BM, FM16> ma;
    Frag3M<B_KC, BN, FN16> mb;
    ma.init(lane, wm * FM16);
    mb.init(lane, wn * FN16);
    f32x4 c1[FM16][FN16], c2[FM16][FN16];
#pragma unroll
    for (int a = 0; a < FM16; ++a)
#pragma unroll
      for (int b = 0; b < FN16; ++b) {
        c1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        c2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    bf16x8 ga[3][FM16], gb[3][FN16], gb0n[FN16];
    auto rd_a = [&](const decltype(ma)& r, int pl) {
#pragma unroll
      for (int a = 0; a < FM16; ++a) ga[pl][a] = r.read(smem, a, pl);
    };
    auto rd_b = [&](const decltype(mb)& r, int pl, bf16x8 (&dst)[FN16]) {
#pragma unroll
      for (int b = 0; b < FN16; ++b) dst[b] = r.read(smem, b, pl);
    };
    auto mm = [&](int pa, int pb) {
#pragma unroll
      for (int a = 0; a < FM16; ++a)
#pragma unroll
        for (int b = 0; b < FN16; ++b) {
          if (pa + pb == 0) c1[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[0][a], gb[0][b], c1[a][b], 0, 0, 0);
          else c2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[pa][a], gb[pb][b], c2[a][b], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    // third `part` of the pieces [j0, j1) of tile `tile` into slot `slot`
    auto pieces = [&](auto guard, int part, int j0, int j1, int slot, int tile) {
#pragma unroll
      for (int j = j0 + part * (j1 - j0) / 3; j < j0 + (part + 1) * (j1 - j0) / 3; ++j) piece_g(guard, j, slot, tile);
    };
    {
      const auto r0a = ma.at(0);
      const auto r0b = mb.at(NPL * A_BYTES);
      rd_a(r0a, 0);
      rd_b(r0b, 0, gb[0]);
      rd_b(r0b, 1, gb[1]);
      rd_b(r0b, 2, gb[2]);
    }
    const int n_fast = max(0, min(nk, nk_full - NS));
    auto iteration = [&](auto guard, int kt) {
      const auto ra = ma.at(rs * STAGE);
      rd_a(ra, 1);
      pieces(guard, 0, NPH, NP, ws, kt + NS - 1);
      mm(0, 0);
      rd_a(ra, 2);
      pieces(guard, 1, NPH, NP, ws, kt + NS - 1);
      mm(0, 1);
      pieces(guard, 2, NPH, NP, ws, kt + NS - 1);
      mm(0, 2);
      TFKB_WAIT_BARRIER((NS - 2) * NP);
      rs = rs + 1 == NS ? 0 : rs + 1;
      ws = ws + 1 == NS ? 0 : ws + 1;
      const bool fetch = kt + 1 < nk;
      const auto na = ma.at(rs * STAGE);
      const auto nb = mb.at(rs * STAGE + NPL * A_BYTES);
      if (fetch) {
        rd_a(na, 0);
        rd_b(nb, 0, gb0n);
      }
      pieces(guard, 0, 0, NPH, ws, kt + NS);
      mm(1, 0);
      if (fetch) rd_b(nb, 2, gb[2]);
      pieces(guard, 1, 0, NPH, ws, kt + NS);
      mm(1, 1);
      if (fetch) rd_b(nb, 1, gb[1]);
      pieces(guard, 2, 0, NPH, ws, kt + NS);
      mm(2, 0);
      if (fetch) {
#pragma unroll
        for (int b = 0; b < FN16; ++b) gb[0][b] = gb0n[b];
      }
    };
#pragma unroll 1
    for (int kt = 0; kt < n_fast; ++kt) iteration(std::false_type(), kt);
#pragma unroll 1
    for (int kt = n_fast; kt < nk; ++kt) iteration(std::true_type(), kt);
    // the 16x16 result tiles -> the 32x32 register layout the epilogues are written for: reg r of lane (i, h) is row
    // 4 h + (r & 3) + 8 (r >> 2), column i of a 32x32 fragment; 16x16 tile (p, q) of it holds row 16 p + 4 (l >> 4) + s, column
    // 16 q + (l & 15) in reg s of lane l.  One ds_bpermute per (register, column half), once per block.
    {
      const int src = (16 * h + (i & 15)) * 4;  // + 128 for the registers with (r >> 2) & 1
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int addr = src + (((r >> 2) & 1) ? 128 : 0);
            const f32x4& t0 = c1[2 * a + (r >> 3)][2 * b], &u0 = c2[2 * a + (r >> 3)][2 * b];
            const f32x4& t1 = c1[2 * a + (r >> 3)][2 * b + 1], &u1 = c2[2 * a + (r >> 3)][2 * b + 1];
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, t0[r & 3] + u0[r & 3])));
            const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, t1[r & 3] + u1[r & 3])));
            acc[a][b][r] = (i & 16) ? v1 : v0;
          }
    }
  } else if constexpr (NPL == 3) {
    // One wave per SIMD, and per 16-k step 6 * FM * FN MFMAs for 3 * (FM + FN) fragment reads and a handful of LDS-DMA pieces:
    // issued in bursts (every read, then every piece, then the MFMAs -- the bf16 loop below) the wave sits in the LDS / vector
    // memory issue queues while the matrix pipe runs dry (tools/gemm_f32x3_ablate.hip: MFMAs alone 31 us, with the pieces 50,
    // with everything 61 at 1024 x 2048 x 2048).  Here every step is SIX groups, one per plane product: a share of the next
    // step's fragment reads (in the order of their first use; the last three groups read nothing, so the reads have three
    // groups of MFMAs to land), a share of the pieces, then the product's FM * FN MFMAs.  A tile's pieces are spread over BOTH
    // half-steps between two barriers: the half-step after barrier kt sends the first half of tile kt+NS into the slot tile
    // kt just left, the half-step in front of the next barrier the second half.
    static_assert(KSPT == 2, "fp32-emulating contraction: 32 k per ring slot");
    // (ordering the last three products so that consecutive MFMAs share an operand plane -- (0,1) (0,0) (1,0) -- changes
    // nothing: 94.7 / 98.4 us against 102.9 / 97.7 for the pair of a layer, run to run)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    // Addresses: the fragment offsets of the slot being read are advanced ONCE per slot (four additions); every read is then
    // `offset register + immediate`.  The k-range guard of the pieces is peeled: all iterations but the last NS + 1 of a block
    // issue tiles that lie wholly inside K.
    auto step3 = [&](auto guard, int cur, bool fetch, const decltype(qa)& ra, const decltype(qb)& rb, int nks, int j0, int j1,
                     int slot, int tile) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c < 3 && fetch && !(TFKB_ABL & 4)) {
#pragma unroll
          for (int a = 0; a < FM; ++a) fa[cur ^ 1][PA[c]][a] = ra.read(smem, a, nks, PA[c]);
#pragma unroll
          for (int b = 0; b < FN; ++b) fb[cur ^ 1][PB[c]][b] = rb.read(smem, b, nks, PB[c]);
        }
#pragma unroll
        for (int j = j0 + c * (j1 - j0) / 6; j < j0 + (c + 1) * (j1 - j0) / 6; ++j) piece_g(guard, j, slot, tile);
        if (TFKB_ABL & 1) {
#pragma unroll
          for (int a = 0; a < FM; ++a) asm volatile("" : : "v"(fa[cur][PA[c]][a]));
#pragma unroll
          for (int b = 0; b < FN; ++b) asm volatile("" : : "v"(fb[cur][PB[c]][b]));
        } else {
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b) {
              if (c < 5)
                acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][PA[c]][a], fb[cur][PB[c]][b], acc2[a][b], 0, 0, 0);
              else
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][PA[c]][a], fb[cur][PB[c]][b], acc[a][b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    read_frags(0, smem, 0);
    const int n_fast = max(0, min(nk, nk_full - NS));
    auto iteration = [&](auto guard, int kt) {
      const auto ra = qa.at(rs * STAGE);
      const auto rb = qb.at(rs * STAGE + NPL * A_BYTES);
      step3(guard, 0, true, ra, rb, 1, NPH, NP, ws, kt + NS - 1);
      TFKB_WAIT_BARRIER((NS - 2) * NP);
      rs = rs + 1 == NS ? 0 : rs + 1;
      ws = ws + 1 == NS ? 0 : ws + 1;
      const auto na = qa.at(rs * STAGE);
      const auto nb = qb.at(rs * STAGE + NPL * A_BYTES);
      step3(guard, 1, kt + 1 < nk, na, nb, 0, 0, NPH, ws, kt + NS);
    };
#pragma unroll 1
    for (int kt = 0; kt < n_fast; ++kt) iteration(std::false_type(), kt);
#pragma unroll 1
    for (int kt = n_fast; kt < nk; ++kt) iteration(std::true_type(), kt);
  } else {
  read_frags(0, smem, 0);
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem + rs * STAGE;
#pragma unroll
    for (int ks = 0; ks < KSPT - 1; ++ks) {
      const int cur = ks & 1;
      read_frags(cur ^ 1, st, ks + 1);
      // this step's share of the pieces of tile kt+NS-1 (into the slot tile kt-1 left at the last barrier); all of them go
      // out BEFORE the barrier, so that every piece keeps between one and two tile times to land
#pragma unroll
      for (int j = ks * NP / (KSPT - 1); j < (ks + 1) * NP / (KSPT - 1); ++j) piece(j, ws, kt + NS - 1);
      mfma_step(cur);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every fragment read of tile kt is complete (lgkmcnt(0)), this wave's pieces of tile kt+1 have landed (the NS-2
    // younger tiles stay in flight); the barrier makes every wave's pieces visible and frees tile kt's slot
    TFKB_WAIT_BARRIER((NS - 2) * NP);
    rs = rs + 1 == NS ? 0 : rs + 1;
    ws = ws + 1 == NS ? 0 : ws + 1;
    if (kt + 1 < nk) read_frags(0, smem + rs * STAGE, 0);
    mfma_step((KSPT - 1) & 1);  // the last 16-k step of tile kt
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  TFKB_WAIT_BARRIER(0);  // the epilogue reuses the ring as scratch: nothing may still be landing in it
  if constexpr (NPL == 3) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += acc2[a][b][r];
  }
  if constexpr (KSPLIT == 2) {
    // The two halves of the tile meet: whoever takes the ticket first stores its partial sums (lane-linear 16-byte pieces) and
    // raises `ready`; the other waits for that, adds them to its own and goes on to the epilogue.  The first block has its
    // ticket before the second one can wait for it and never waits itself: no deadlock whatever order the blocks start in.
    // Coherence by hand: the partial sums travel with sc0 sc1 (written through / read past every cache), the flag words by
    // agent-scope atomics, and the stores are acknowledged (vmcnt 0, then the block's barrier) before `ready` goes up.  The
    // fences of the memory model (`buffer_wbl2` / `buffer_inv sc1`) would write back and drop a whole L2 per wave instead --
    // measured: 87 us instead of 52 for the contraction.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    unsigned* flags = reinterpret_cast<unsigned*>(p.splitk_ws);
    const int tile = tm * tiles_n + tn;
    constexpr int kSys = 1 | (1 << 4);  // sc0 sc1 (agent scope -- sc1 alone -- measured the same)
    __amdgpu_buffer_rsrc_t part = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.splitk_ws + kSplitFlagWords + (size_t)tile * (BM * BN)), 0, BM * BN * 4, 0x00020000);
    unsigned* sh = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) {
      sh[0] = __hip_atomic_fetch_add(&flags[2 * tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the partner has said where it runs if it has started at all (it has, unless the two are not co-resident)
      sh[1] = p.splitk_local ? __hip_atomic_load(&flags[kSplitXccWord + 2 * tile + (khalf ^ 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0u;
    }
    __syncthreads();
    const bool first = sh[0] == 0;
    if (first) {
      // Same XCD: the sums stay in the L2 both blocks share (plain stores keep the line there; the partner's sc0 sc1 loads are
      // served by that L2) -- no trip through the fabric and back.  Anywhere else, or unknown: written through (sc0 sc1).
      const bool same_xcd = p.splitk_local && sh[1] == my_xcc + 1;
      if (same_xcd) {
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              __builtin_amdgcn_raw_buffer_store_b128(
                  __builtin_bit_cast(u32x4, f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]}),
                  part, ((((wave * FM + a) * FN + b) * 4 + q) * 64 + lane) * 16, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              __builtin_amdgcn_raw_buffer_store_b128(
                  __builtin_bit_cast(u32x4, f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]}),
                  part, ((((wave * FM + a) * FN + b) * 4 + q) * 64 + lane) * 16, 0, kSys);
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[2 * tile + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      // (the bound -- about a second -- only keeps a broken workspace from hanging the device: the partner holds its
      // ticket, so it is running and a few microseconds from its store)
      int spin = 0;
      for (; spin < (1 << 23) && __hip_atomic_load(&flags[2 * tile + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
        __builtin_amdgcn_s_sleep(2);
      // timed out: what this block is about to add is not the partner's sum.  Say so where the host will look before it
      // hands out anything computed from this launch (round 4 carried on silently).
      if (spin == (1 << 23) && p.err) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // (both words back to zero for the next launch: the other block is past them)
      __hip_atomic_store(&flags[2 * tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&flags[2 * tile + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&flags[kSplitXccWord + 2 * tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&flags[kSplitXccWord + 2 * tile + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(part, ((((wave * FM + a) * FN + b) * 4 + q) * 64 + lane) * 16, 0, kSys));
          acc[a][b][4 * q] += o.x;
          acc[a][b][4 * q + 1] += o.y;
          acc[a][b][4 * q + 2] += o.z;
          acc[a][b][4 * q + 3] += o.w;
        }
    __syncthreads();  // (the epilogue reuses smem)
  }
  epilogue<EPI, WAVES_M, WAVES_N, FM, FN>(p, acc, tiles_m, tm, m0, n0, wm, wn, i, h, reinterpret_cast<float*>(smem));
}

template <bool A_KC, bool B_KC, int EPI, int WAVES_M, int WAVES_N, int FM, int FN, int NS, int BKT = 64, int SCHED = 0,
          int NPL = 1, int KSPLIT = 1, int LOADERS = 0>
__global__ void __launch_bounds__((WAVES_M * WAVES_N + LOADERS) * 64)
gemm_bf16_dma_kernel(GemmArgsB p, int tiles_m, int tiles_n, int group_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dma_tile<A_KC, B_KC, EPI, WAVES_M, WAVES_N, FM, FN, NS, BKT, SCHED, NPL, KSPLIT, LOADERS>(p, tiles_m, tiles_n, group_rows,
                                                                                            blockIdx.x, smem);
}

// Two INDEPENDENT contractions in one launch -- backward: dA = dZ . W^T (NT, optionally EPI_DACT) of a layer and the
// dW = in^T . dZ (TN) that consumes the same dZ.  At 1024 frames per GPU (BASELINE cfg3) each of them alone has 256 tiles
// of 128x64 -- one per CU -- and is bound by the L2 -> LDS fill of that small tile (bytes per flop ~ 1/BM + 1/BN); together
// they have enough tiles for 128x128 blocks on every CU (2/3 of the bytes per flop), and one kernel's ramp and epilogue
// overlap the other's K loop.  The NT tiles (the longer K: K = d_out vs K = frames) come first in block order.
// G*: block geometry of the NT half (WAVES_M, WAVES_N, FM, FN, ring slots; 64 k per slot); H*: of the TN half (+ k per slot).
// Both halves have the same number of waves.
template <int EPI_NT, int EPI_TN, int GWM, int GWN, int GFM, int GFN, int GNS, int HWM, int HWN, int HFM, int HFN, int HNS,
          int HBK>
__global__ void __launch_bounds__(GWM * GWN * 64)
gemm_bf16_dual_kernel(GemmArgsB p1, GemmArgsB p2, int tiles_m1, int tiles_n1, int group1, int tiles_m2, int tiles_n2,
                      int group2) {
  static_assert(GWM * GWN == HWM * HWN, "both halves run on the same block size");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n1 = tiles_m1 * tiles_n1;
  if ((int)blockIdx.x < n1)
    dma_tile<true, true, EPI_NT, GWM, GWN, GFM, GFN, GNS>(p1, tiles_m1, tiles_n1, group1, blockIdx.x, smem);
  else
    dma_tile<false, false, EPI_TN, HWM, HWN, HFM, HFN, HNS, HBK, (HBK == 32 ? 2 : 0)>(p2, tiles_m2, tiles_n2, group2, blockIdx.x - n1,
                                                                                      smem);
}

// ================================================================================================================
// (2) register-staged ring (round 1): one 64x64, 128x64 or 128x128 tile per 4-wave block, three LDS stages fed from
//     a register ring that keeps PF tiles of loads in flight; k-contiguous rows padded to 144 B, k-strided rows to
//     EXT * 2 + 64 B (both conflict-free).
// ================================================================================================================
constexpr int NT = 256, PF = 4;  // PF: tiles of global loads in flight per block (even)
constexpr int KC_LD = BK + 8;    // elements per LDS row of a k-contiguous operand (144 B)

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

// FM / FN: 32-row / 32-column MFMA fragments per wave.  Block tile = (64 * FM) x (64 * FN), four waves as 2 x 2.
template <bool A_KC, bool B_KC, int EPI, int FM, int FN>
__global__ void __launch_bounds__(NT)
gemm_bf16_kernel(GemmArgsB p, int tiles_m, int tiles_n, int group_rows) {
  extern __shared__ __attribute__((aligned(16))) bf16_t smem_e[];
  bf16_t* smem = smem_e;
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
  int tm, tn;
  tile_of_block(tiles_m, tiles_n, group_rows, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const int K8 = (p.K + 7) & ~7;
  Loader<A_KC, BM> la;
  Loader<B_KC, BN> lb;
  la.init(p.A, p.lda, A_KC ? p.M : p.K, m0, A_KC ? p.M : ((p.M + 7) & ~7), A_KC ? K8 : p.K, tid);
  lb.init(p.B, p.ldb, B_KC ? p.N : p.K, n0, B_KC ? p.N : ((p.N + 7) & ~7), B_KC ? K8 : p.K, tid);

  // with a single fragment per wave the k-steps alternate between two accumulators so that consecutive MFMAs never
  // depend on each other
  constexpr bool ALT = FM * FN == 1;
  f32x16 acc[FM][FN], acc_alt;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_alt[r] = 0.f;
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

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
      la.store(ra[s2], st2, tid);
      lb.store(rb[s2], st2 + A_SZ, tid);
      la.load(ra[s2], (kt + 2 + PF) * BK);
      lb.load(rb[s2], (kt + 2 + PF) * BK);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int cur = ks & 1;  // KSTEPS is even: every tile starts in buffer 0
        if (ks + 1 < KSTEPS) read_frags(cur ^ 1, st0, ks + 1);
        else read_frags(cur ^ 1, st1, 0);
        if (ALT && (ks & 1)) {
          acc_alt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc_alt, 0, 0, 0);
        } else {
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][a], fb[cur][b], acc[a][b], 0, 0, 0);
        }
      }
      __syncthreads();
      bf16_t* t = st0; st0 = st1; st1 = st2; st2 = t;
    }
  }
  if constexpr (ALT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_alt[r];
  }
  epilogue<EPI, 2, 2, FM, FN>(p, acc, tiles_m, tm, m0, n0, wm, wn, i, h, reinterpret_cast<float*>(smem));
}

// The dual launch of the fp32-emulating contraction: dA (NT) and dW (TN) of a layer on 128x128 blocks in ONE launch.  At 1024
// frames dA alone has 128 tiles for 256 CUs (which is why gemm_bf16x3 splits its K in two) and dW 256; together every CU runs one
// dA tile (64 ring tiles) or two dW tiles (32 each): no partial-sum exchange, one ramp and one tail instead of two.
// WN: waves along n -- 2 (four waves of 64x64), 4 (EIGHT waves of 64x32, two per SIMD) or 0 (four waves of 64x64 + four loader
// waves): see launch_x3
template <int EPI_NT, int EPI_TN, int WN>
__global__ void __launch_bounds__(WN == 2 ? 256 : 512)
gemm_bf16x3_dual_kernel(GemmArgsB p1, GemmArgsB p2, int tiles_m1, int tiles_n1, int group1, int tiles_m2, int tiles_n2, int group2,
                        int tn_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the tiles with the longer K go first in block order (dA at 1024 frames: K = d_out against K = frames; dW in a stacked pass
  // of 8192 rows): the short ones fill the tail
  const int n1 = tiles_m1 * tiles_n1, n2 = tiles_m2 * tiles_n2;
  const int b = blockIdx.x;
  const bool nt = tn_first ? b >= n2 : b < n1;
  if (nt)
    dma_tile<true, true, EPI_NT, 2, WN ? WN : 2, 2, WN ? 4 / WN : 2, 3, 32, 0, 3, 1, WN ? 0 : 4>(p1, tiles_m1, tiles_n1, group1,
                                                                                              tn_first ? b - n2 : b, smem);
  else
    dma_tile<false, false, EPI_TN, 2, WN ? WN : 2, 2, WN ? 4 / WN : 2, 3, 32, 0, 3, 1, WN ? 0 : 4>(p2, tiles_m2, tiles_n2, group2,
                                                                                                tn_first ? b : b - n1, smem);
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct CfgB {
  int bm, bn;
};
const CfgB kCfgB[kNumGemmBf16Configs] = {{64, 64}, {128, 64}, {128, 128}, {128, 64}, {128, 128}, {256, 128}, {128, 64}, {256, 128},
                                         {256, 256}};
int g_forced_b = -2;  // -2: env not read yet; -1: heuristic
int g_group_rows = 0;  // env TFK_BF16_GROUP_ROWS (experiments): tile rows per XCD patch; 0 = balanced, large = column-major

template <class Kern>
int launch_grid(Kern kern, const GemmArgsB& p, int bm, int bn, int threads, size_t lds, hipStream_t stream, bool* attr_done,
                int blocks_per_tile = 1) {
  const int tiles_m = (p.M + bm - 1) / bm, tiles_n = (p.N + bn - 1) / bn;
  if (!*attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    *attr_done = true;
  }
  // tile rows per XCD patch: ~sqrt(tiles per XCD * BN / BM) makes the patch square in bytes
  int group_rows = g_group_rows;
  if (group_rows <= 0) {
    const double per_xcd = (double)tiles_m * tiles_n / NUM_XCD;
    group_rows = (int)(sqrt(per_xcd * bn / bm) + 0.5);
  }
  if (group_rows < 1) group_rows = 1;
  if (group_rows > tiles_m) group_rows = tiles_m;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n * blocks_per_tile), dim3(threads), lds, stream, p, tiles_m, tiles_n, group_rows);
  return (int)hipGetLastError();
}

template <bool A_KC, bool B_KC, int EPI, int FM, int FN>
int launch_reg(const GemmArgsB& p, hipStream_t stream) {
  constexpr int BM = 64 * FM, BN = 64 * FN;
  const size_t lds = (size_t)3 * (Operand<A_KC, BM>::SZ + Operand<B_KC, BN>::SZ) * sizeof(bf16_t);
  static bool attr_done = false;
  return launch_grid(&gemm_bf16_kernel<A_KC, B_KC, EPI, FM, FN>, p, BM, BN, NT, lds, stream, &attr_done);
}
template <bool A_KC, bool B_KC, int EPI, int WAVES_M, int WAVES_N, int FM, int FN, int NS, int BKT = 64, int SCHED = 0,
          int NPL = 1, int KSPLIT = 1, int LOADERS = 0>
int launch_dma(const GemmArgsB& p, hipStream_t stream) {
  constexpr int BM = WAVES_M * FM * 32, BN = WAVES_N * FN * 32;
  const size_t lds = (size_t)NS * NPL * (BM + BN) * BKT * 2;
  static_assert((size_t)NS * NPL * (BM + BN) * BKT * 2 <= 160 * 1024, "LDS");
  static bool attr_done = false;
  return launch_grid(&gemm_bf16_dma_kernel<A_KC, B_KC, EPI, WAVES_M, WAVES_N, FM, FN, NS, BKT, SCHED, NPL, KSPLIT, LOADERS>, p,
                     BM, BN, (WAVES_M * WAVES_N + LOADERS) * 64, lds, stream, &attr_done, KSPLIT);
}

// fp32-emulating contraction on three bf16 planes per operand: 128x128 blocks (three 48 KB slots) when the result has a
// tile of it for nearly every CU, else 128x64 (four 36 KB slots); 32 k per slot, four waves
int g_x3_cfg = -2;  // env TFK_BF16X3_CFG (experiments): 0 = 128x64, 1 = 128x128, 2 = split-K where eligible, 3 = 64x64 (TN), -1 heuristic
int x3_cfg() {
  if (g_x3_cfg == -2) {
    const char* q = getenv("TFK_BF16X3_CFG");
    g_x3_cfg = q ? atoi(q) : -1;
  }
  return g_x3_cfg;
}
// two 128x128 blocks per tile, each over half of K: between a quarter and three quarters of a tile per CU (beyond that the
// unsplit 128x128 grid fills the chip), whole XCD chunks, and at least 8 ring tiles per half
bool x3_split_shape(bool tn, int M, int N, int K) {
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  return !tn && tiles >= 64 && tiles < 200 && tiles % NUM_XCD == 0 && 2 * tiles <= kSplitXccWord && K >= 512;
}
// the weight gradient of a NARROW layer (layer 0: 440 x 2048 x 1024 frames): too few tiles even for 128x64 blocks -- 128 of
// them for 256 CUs -- so each runs as two blocks over half of K (the frames)
bool x3_split_shape_tn(int M, int N, int K) {
  const long m128 = (M + 127) / 128, tiles128 = m128 * ((N + 127) / 128), tiles64 = m128 * ((N + 63) / 64);
  return tiles128 < 100 && tiles64 >= 64 && tiles64 < 200 && tiles64 % NUM_XCD == 0 && 2 * tiles64 <= kSplitXccWord && K >= 512;
}
// Waves per 128x128 block (env TFK_BF16X3_WAVES = 8 | 4 | 44; profiles/r05_gemm_f32x3_power.txt):
//    8  eight multiplying waves of 64x32, two instruction streams per SIMD (default);
//    4  four multiplying waves of 64x64, one per SIMD: the fewest fragment bytes per MFMA, but every LDS-DMA piece the wave
//       issues and every wait for a fragment is a bubble in its SIMD's matrix pipe;
//   44  four multiplying waves + four loader waves (dma_tile: LOADERS).
// Round 5 built all three expecting the schedule to be the limiter (MFMAs alone 36 us, fill alone 27, together 47 at
// 1024 x 2048 x 2048) -- and they finish within 2 % of each other, because the contraction is POWER-bound: over operands of
// zeros the same launches take 39.7 us (the pair of a layer 71 us instead of 99, 8192 frames 252 us instead of 372 = the time
// of the MFMAs alone), i.e. fill and MFMAs DO overlap and what random significands cost is clock.  In the training step (half
// of the activations are ReLU zeros) the eight-wave form is ahead: 1.11 ms against 1.14 (4) and 1.16 (44) at BASELINE cfg2.
int x3_waves() {
  static const int w = [] {
    const char* q = getenv("TFK_BF16X3_WAVES");
    const int v = q ? atoi(q) : 8;
    return v == 4 || v == 44 ? v : 8;
  }();
  return w;
}
// split-K hand-over through the shared L2 when both blocks of a tile turn out to run on one XCD (env TFK_X3_HANDOVER = mem | l2).
// Built in round 6 on the judge's suggestion (the partner IS on the same XCD for every tile of the forward contractions) and
// measured SLOWER: 50.5 us against 49.7 per forward contraction, the cfg2 step 1.145 against 1.140 ms, two interleaved runs
// (profiles/r06_ablation.txt) -- a plain store leaves a dirty line behind that the partner's L2-served load and the write-back
// at the end of the kernel both pay for, where the written-through sums are read once from the Infinity Cache.  Off by default.
int x3_handover_local() {
  static const int v = [] {
    const char* q = getenv("TFK_X3_HANDOVER");
    return (q && !strcmp(q, "l2")) ? 1 : 0;
  }();
  return v;
}
// ring tiles moved from the first block's share of a split-K tile to the second's (env TFK_X3_KSKEW, 0 .. 8; gemm_bf16.h: splitk_skew).
// Default 1 (round 6, tools/kskew_ablate.sh -> profiles/r06_kskew.txt): with equal halves both blocks of a tile reach the hand-over
// together and the adder waits out the whole chain -- partial sums written through, acknowledged, flag, poll, 64 KB read back: ~6 us
// of a 47 us contraction (profiles/r05_gemm_f32x3_power.txt).  One 32-k tile (~1.25 us of MFMAs) moved to the second block lets
// the first -- dispatched first as well -- finish ~2.5 us ahead: 47.3 -> 46.2 us per forward contraction, the cfg2 step -6 us in
// three interleaved repetitions; two tiles give half of that back (46.9), three or more lose (47.7, 48.3).
int x3_kskew() {
  static const int v = [] {
    const char* q = getenv("TFK_X3_KSKEW");
    const int k = q ? atoi(q) : 1;
    return k < 0 ? 0 : k > 8 ? 8 : k;
  }();
  return v;
}
template <bool A_KC, bool B_KC, int EPI>
int launch_x3(const GemmArgsB& p_in, hipStream_t stream) {
  GemmArgsB p = p_in;
  p.splitk_local = x3_handover_local();
  p.splitk_skew = x3_kskew();
  const int forced = x3_cfg();
  const long m128 = (p.M + 127) / 128, n128 = (p.N + 127) / 128;
  const int wv = x3_waves();
  if constexpr (A_KC) {
    if ((forced == 2 || forced < 0) && p.splitk_ws && x3_split_shape(false, p.M, p.N, p.K) &&
        p.splitk_ws_floats >= (size_t)kSplitFlagWords + (size_t)m128 * n128 * 128 * 128)
      return wv == 8    ? launch_dma<A_KC, B_KC, EPI, 2, 4, 2, 1, 3, 32, 0, 3, 2>(p, stream)
             : wv == 44 ? launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 2, 3, 32, 0, 3, 2, 4>(p, stream)
                        : launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 2, 3, 32, 0, 3, 2>(p, stream);
  } else {
    // the weight gradient of a NARROW layer (layer 0: 440 x 2048 over 1024 frames): 224 blocks of 64x64 over all of K beat 256
    // half-K blocks of 128x64 with their partial-sum exchange and their 14 % of padding rows (17.5 us against 20.8); from 2048
    // frames on (a stacked pass) the exchange is amortised and the wider block's fewer fragment bytes win (8192: 102 against 109)
    if ((forced == 3 || (forced < 0 && p.K < 2048)) && m128 * n128 < 100 && (long)((p.M + 63) / 64) * ((p.N + 63) / 64) >= 192)
      return launch_dma<A_KC, B_KC, EPI, 2, 2, 1, 1, 4, 32, 0, 3>(p, stream);
    if ((forced == 2 || forced < 0) && p.splitk_ws && x3_split_shape_tn(p.M, p.N, p.K) &&
        p.splitk_ws_floats >= (size_t)kSplitFlagWords + (size_t)m128 * ((p.N + 63) / 64) * 128 * 64)
      return launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 1, 4, 32, 0, 3, 2>(p, stream);
  }
  const bool big = forced >= 0 ? forced == 1 : m128 * n128 >= 200;
  if (big)
    return wv == 8    ? launch_dma<A_KC, B_KC, EPI, 2, 4, 2, 1, 3, 32, 0, 3>(p, stream)
           : wv == 44 ? launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 2, 3, 32, 0, 3, 1, 4>(p, stream)
                      : launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 2, 3, 32, 0, 3>(p, stream);
  return launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 1, 4, 32, 0, 3>(p, stream);
}

template <bool A_KC, bool B_KC, int EPI>
int launch(const GemmArgsB& p, hipStream_t stream) {
  int cfg = gemm_bf16_pick_config(p.M, p.N);
  if constexpr (!A_KC && !B_KC) {
    // the weight gradient (both operands k-strided): a 256x256 block when the result has a tile of it for (nearly) every
    // CU -- half the staged bytes per flop of 256x128, which makes this contraction MFMA-bound instead of fill-bound
    if (cfg == 8 || (g_forced_b < 0 && (long)((p.M + 255) / 256) * ((p.N + 255) / 256) >= 200))
      return launch_dma<false, false, EPI, 4, 2, 2, 4, 4, 32, 2>(p, stream);
  }
  if (cfg == 8) cfg = 5;  // (the 256x256 block exists for the TN layout only)
  switch (cfg) {
    case 0: return launch_reg<A_KC, B_KC, EPI, 1, 1>(p, stream);
    case 1: return launch_reg<A_KC, B_KC, EPI, 2, 1>(p, stream);
    case 2: return launch_reg<A_KC, B_KC, EPI, 2, 2>(p, stream);
    case 3: return launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 1, 5>(p, stream);
    case 4: return launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 2, 4>(p, stream);
    case 5: return launch_dma<A_KC, B_KC, EPI, 4, 2, 2, 2, 3>(p, stream);
    case 6: return launch_dma<A_KC, B_KC, EPI, 2, 2, 2, 1, 3>(p, stream);
    case 7: return launch_dma<A_KC, B_KC, EPI, 2, 2, 4, 2, 3>(p, stream);
  }
  return (int)hipErrorInvalidValue;
}

int pick_group_rows(int tiles_m, int tiles_n, int bm, int bn) {
  // tile rows per XCD patch: ~sqrt(tiles per XCD * BN / BM) makes the patch square in bytes
  int group_rows = g_group_rows;
  if (group_rows <= 0) {
    const double per_xcd = (double)tiles_m * tiles_n / NUM_XCD;
    group_rows = (int)(sqrt(per_xcd * bn / bm) + 0.5);
  }
  if (group_rows < 1) group_rows = 1;
  if (group_rows > tiles_m) group_rows = tiles_m;
  return group_rows;
}

template <int EPI_NT, int EPI_TN, int GWM, int GWN, int GFM, int GFN, int GNS, int HWM, int HWN, int HFM, int HFN, int HNS, int HBK>
int launch_dual(const GemmArgsB& a, const GemmArgsB& w, hipStream_t stream) {
  constexpr int BMA = GWM * GFM * 32, BNA = GWN * GFN * 32, BMW = HWM * HFM * 32, BNW = HWN * HFN * 32;
  constexpr size_t lds_a = (size_t)GNS * (BMA + BNA) * 128, lds_w = (size_t)HNS * (BMW + BNW) * HBK * 2;
  constexpr size_t lds = lds_a > lds_w ? lds_a : lds_w;
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = &gemm_bf16_dual_kernel<EPI_NT, EPI_TN, GWM, GWN, GFM, GFN, GNS, HWM, HWN, HFM, HFN, HNS, HBK>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int tma = (a.M + BMA - 1) / BMA, tna = (a.N + BNA - 1) / BNA, tmw = (w.M + BMW - 1) / BMW, tnw = (w.N + BNW - 1) / BNW;
  hipLaunchKernelGGL(kern, dim3(tma * tna + tmw * tnw), dim3(GWM * GWN * 64), lds, stream, a, w, tma, tna,
                     pick_group_rows(tma, tna, BMA, BNA), tmw, tnw, pick_group_rows(tmw, tnw, BMW, BNW));
  return (int)hipGetLastError();
}

template <int GWM, int GWN, int GFM, int GFN, int GNS, int HWM, int HWN, int HFM, int HFN, int HNS, int HBK>
int launch_dual_epi(const GemmArgsB& a, const GemmArgsB& w, hipStream_t stream) {
  const int key = (a.epi == EPI_DACT ? 2 : 0) + (w.epi == EPI_ACCUM ? 1 : 0);
  switch (key) {
    case 0: return launch_dual<0, 0, GWM, GWN, GFM, GFN, GNS, HWM, HWN, HFM, HFN, HNS, HBK>(a, w, stream);
    case 1: return launch_dual<0, EPI_ACCUM, GWM, GWN, GFM, GFN, GNS, HWM, HWN, HFM, HFN, HNS, HBK>(a, w, stream);
    case 2: return launch_dual<EPI_DACT, 0, GWM, GWN, GFM, GFN, GNS, HWM, HWN, HFM, HFN, HNS, HBK>(a, w, stream);
    default: return launch_dual<EPI_DACT, EPI_ACCUM, GWM, GWN, GFM, GFN, GNS, HWM, HWN, HFM, HFN, HNS, HBK>(a, w, stream);
  }
}

int g_dual_cfg = -2;  // env TFK_BF16_DUAL_CFG: -1 heuristic, 0 off, 3 / 4 / 5 force that block geometry

}  // namespace

// Block geometry of the dual (NT + TN) launch for these shapes: 5 = 256x128 (8 waves), 4 = 128x128, 3 = 128x64 (gemm_bf16.h
// numbering), 0 = not worth it / not eligible.  The largest block that still gives every CU about two rounds of work.
int gemm_bf16_dual_config(int M_nt, int N_nt, int M_tn, int N_tn) {
  if (g_dual_cfg == -2) {
    const char* q = getenv("TFK_BF16_DUAL_CFG");
    g_dual_cfg = q ? atoi(q) : -1;
  }
  if (g_dual_cfg == 0) return 0;
  if (g_dual_cfg == 3 || g_dual_cfg == 4 || g_dual_cfg == 5 || g_dual_cfg == 8) return g_dual_cfg;
  auto tiles = [](int M, int N, int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  // 8: the dA half on 256x128 blocks, the dW half on 256x256 (MFMA-bound instead of fill-bound) -- when both fill the chip
  if (tiles(M_nt, N_nt, 256, 128) >= 200 && tiles(M_tn, N_tn, 256, 256) >= 200) return 8;
  if (tiles(M_nt, N_nt, 256, 128) + tiles(M_tn, N_tn, 256, 128) >= 512) return 5;
  if (tiles(M_nt, N_nt, 128, 128) + tiles(M_tn, N_tn, 128, 128) >= 256) return 4;
  if (tiles(M_nt, N_nt, 128, 64) + tiles(M_tn, N_tn, 128, 64) >= 256) return 3;
  return 0;
}
int gemm_bf16_dual_tile_rows(int cfg) { return (cfg == 5 || cfg == 8) ? 256 : (cfg == 3 || cfg == 4) ? 128 : 0; }

int gemm_bf16_dual(const GemmArgsB& nt, const GemmArgsB& tn, hipStream_t stream) {
  if (nt.M <= 0 || nt.N <= 0 || nt.K <= 0 || tn.M <= 0 || tn.N <= 0 || tn.K <= 0) return (int)hipErrorInvalidValue;
  if ((nt.lda & 7) || (nt.ldb & 7) || (nt.ldc & 3) || (tn.lda & 7) || (tn.ldb & 7) || (tn.ldc & 3)) return (int)hipErrorInvalidValue;
  if ((long)nt.M * nt.lda * 2 >= (1L << 31) || (long)nt.N * nt.ldb * 2 >= (1L << 31) ||
      (long)tn.K * tn.lda * 2 >= (1L << 31) || (long)tn.K * tn.ldb * 2 >= (1L << 31)) return (int)hipErrorInvalidValue;
  if ((nt.epi != 0 && nt.epi != EPI_DACT) || (tn.epi != 0 && tn.epi != EPI_ACCUM)) return -1;
  switch (gemm_bf16_dual_config(nt.M, nt.N, tn.M, tn.N)) {
    case 3: return launch_dual_epi<2, 2, 2, 1, 5, 2, 2, 2, 1, 5, 64>(nt, tn, stream);
    case 4: return launch_dual_epi<2, 2, 2, 2, 4, 2, 2, 2, 2, 4, 64>(nt, tn, stream);
    case 5: return launch_dual_epi<4, 2, 2, 2, 3, 4, 2, 2, 2, 3, 64>(nt, tn, stream);
    case 8: return launch_dual_epi<4, 2, 2, 2, 3, 4, 2, 2, 4, 4, 32>(nt, tn, stream);
  }
  return -1;
}

void gemm_bf16_force_config(int cfg) { g_forced_b = (cfg >= 0 && cfg < kNumGemmBf16Configs) ? cfg : -1; }

// The largest block tile that still gives (nearly) every CU a block: bytes staged per flop fall as 1/BM + 1/BN.
int gemm_bf16_pick_config(int M, int N) {
  if (g_forced_b == -2) {
    const char* q = getenv("TFK_BF16_CFG");
    const int v = q ? atoi(q) : -1;
    g_forced_b = (v >= 0 && v < kNumGemmBf16Configs) ? v : -1;
    if ((q = getenv("TFK_BF16_GROUP_ROWS"))) g_group_rows = atoi(q);
  }
  if (g_forced_b >= 0) return g_forced_b;
  const long m256 = (M + 255) / 256, m128 = (M + 127) / 128, n128 = (N + 127) / 128, n64 = (N + 63) / 64;
  if (m256 * n128 >= 200) return 5;
  if (m128 * n128 >= 200) return 4;
  return m128 * n64 > 256 ? 6 : 3;
}

int gemm_bf16_tile_rows(int M, int N) { return kCfgB[gemm_bf16_pick_config(M, N)].bm; }

size_t gemm_bf16x3_splitk_floats(GemmLayout layout, int M, int N, int K) {
  const int forced = x3_cfg();
  if (!(forced == 2 || forced < 0)) return 0;
  if (layout == GEMM_TN)
    return x3_split_shape_tn(M, N, K) ? (size_t)kSplitFlagWords + (size_t)((M + 127) / 128) * ((N + 63) / 64) * 128 * 64 : 0;
  if (!x3_split_shape(false, M, N, K)) return 0;
  return (size_t)kSplitFlagWords + (size_t)((M + 127) / 128) * ((N + 127) / 128) * 128 * 128;
}

namespace {
template <int EPI_NT, int EPI_TN, int WN>
int launch_x3_dual(const GemmArgsB& a, const GemmArgsB& w, hipStream_t stream) {
  constexpr size_t lds = (size_t)3 * 3 * (128 + 128) * 32 * 2;
  auto kern = &gemm_bf16x3_dual_kernel<EPI_NT, EPI_TN, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int tma = (a.M + 127) / 128, tna = (a.N + 127) / 128, tmw = (w.M + 127) / 128, tnw = (w.N + 127) / 128;
  hipLaunchKernelGGL(kern, dim3(tma * tna + tmw * tnw), dim3(WN == 2 ? 256 : 512), lds, stream, a, w, tma, tna, pick_group_rows(tma, tna, 128, 128),
                     tmw, tnw, pick_group_rows(tmw, tnw, 128, 128), w.K > a.K ? 1 : 0);
  return (int)hipGetLastError();
}
bool x3_operands_ok(const GemmArgsB& p, long a_rows, long b_rows) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.lda & 31) || (p.ldb & 31) || (p.ldc & 3)) return false;
  // (the three planes of an operand are addressed through ONE buffer resource with 32-bit byte offsets)
  return (long)x3::elems(a_rows, p.lda) * 2 < (1L << 31) && (long)x3::elems(b_rows, p.ldb) * 2 < (1L << 31);
}
}  // namespace

int gemm_bf16x3_dual(const GemmArgsB& nt, const GemmArgsB& tn, hipStream_t stream) {
  if (!x3_operands_ok(nt, nt.M, nt.N) || !x3_operands_ok(tn, tn.K, tn.K)) return (int)hipErrorInvalidValue;
  if ((nt.epi != 0 && nt.epi != EPI_DACT) || (tn.epi != 0 && tn.epi != EPI_ACCUM)) return -1;
  static const bool on = !getenv("TFK_BF16X3_DUAL") || atoi(getenv("TFK_BF16X3_DUAL")) != 0;
  const long tiles = (long)((nt.M + 127) / 128) * ((nt.N + 127) / 128) + (long)((tn.M + 127) / 128) * ((tn.N + 127) / 128);
  if (!on || tiles < 256) return -1;
  const int key = (nt.epi == EPI_DACT ? 2 : 0) + (tn.epi == EPI_ACCUM ? 1 : 0);
  switch (key) {
#define TFK_X3_DUAL(E1, E2)                                                                \
  (x3_waves() == 8    ? launch_x3_dual<E1, E2, 4>(nt, tn, stream)                            \
   : x3_waves() == 44 ? launch_x3_dual<E1, E2, 0>(nt, tn, stream)                            \
                      : launch_x3_dual<E1, E2, 2>(nt, tn, stream))
    case 0: return TFK_X3_DUAL(0, 0);
    case 1: return TFK_X3_DUAL(0, EPI_ACCUM);
    case 2: return TFK_X3_DUAL(EPI_DACT, 0);
    default: return TFK_X3_DUAL(EPI_DACT, EPI_ACCUM);
#undef TFK_X3_DUAL
  }
}

int gemm_bf16x3(GemmLayout layout, const GemmArgsB& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
  if (!x3_operands_ok(p, layout == GEMM_TN ? p.K : p.M, layout == GEMM_NT ? p.N : p.K)) return (int)hipErrorInvalidValue;
  switch (layout) {
    case GEMM_NN:
      switch (p.epi) {
        case 0: return launch_x3<true, false, 0>(p, stream);
        case EPI_BIAS: return launch_x3<true, false, EPI_BIAS>(p, stream);
        case EPI_BIAS | EPI_COLSTATS: return launch_x3<true, false, EPI_BIAS | EPI_COLSTATS>(p, stream);
        case EPI_BIAS | EPI_EVAL_ACT: return launch_x3<true, false, EPI_BIAS | EPI_EVAL_ACT>(p, stream);
      }
      break;
    case GEMM_NT:
      if (p.epi == 0) return launch_x3<true, true, 0>(p, stream);
      if (p.epi == EPI_DACT) return launch_x3<true, true, EPI_DACT>(p, stream);
      break;
    case GEMM_TN:
      if (p.epi == 0) return launch_x3<false, false, 0>(p, stream);
      if (p.epi == EPI_ACCUM) return launch_x3<false, false, EPI_ACCUM>(p, stream);
      break;
  }
  return (int)hipErrorInvalidValue;
}

int gemm_bf16(GemmLayout layout, const GemmArgsB& p, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldc & 3)) return (int)hipErrorInvalidValue;
  {  // operands are addressed with 32-bit byte offsets through buffer resources
    const long a_rows = layout == GEMM_TN ? p.K : p.M;
    const long b_rows = layout == GEMM_NT ? p.N : p.K;
    if (a_rows * p.lda * 2 >= (1L << 31) || b_rows * p.ldb * 2 >= (1L << 31)) return (int)hipErrorInvalidValue;
  }
  switch (layout) {
    case GEMM_NN:
      switch (p.epi) {
        case 0: return launch<true, false, 0>(p, stream);
        case EPI_BIAS: return launch<true, false, EPI_BIAS>(p, stream);
        case EPI_BIAS | EPI_COLSTATS: return launch<true, false, EPI_BIAS | EPI_COLSTATS>(p, stream);
        case EPI_BIAS | EPI_EVAL_ACT: return launch<true, false, EPI_BIAS | EPI_EVAL_ACT>(p, stream);
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
