// fp32 MFMA GEMM for gfx950 (CDNA4): LDS-tiled, register-staged, double-buffered.
//
// Replaces the TensorFlow matmul kernels behind neuralNetworks/classifiers/layer.py:52 and the
// tf.gradients of it (neuralNetworks/trainer.py:155) -- see gemm_f32.h for the three layouts.
//
// Hardware mapping (MI355X_MICROARCH.md / cdna_hip_programming.md section 3):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles/SIMD, 157 TF chip peak.
//     Operand lane map: A lane l holds A[i = l&31][kslot = l>>5]; B lane l holds B[kslot = l>>5][j = l&31];
//     D reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//   * The contraction index may be permuted freely as long as A and B agree. Within each group of
//     8 consecutive k, MFMA step t (0..3) feeds lane-half h with k = 8g + 4h + t. An operand that is
//     contiguous along k in memory ("KC") can then fetch its four steps with ONE ds_read_b128 per lane;
//     an operand contiguous along m/n ("MC") uses one conflict-free ds_read_b32 per step.
//   * LDS rows of KC tiles are padded by 4 floats (row stride 36 floats = 9 x 16 B, odd in 16-B slots)
//     so the 16-lane service groups of ds_read_b128 hit 16 distinct slots.
//   * One barrier per K-tile (two LDS buffers); the next tile's global loads are issued before the
//     MFMA block of the current tile so L2/HBM latency hides under ~1-4k cycles of matrix work.
//   * Tiles are walked in a grouped order (8 tile-rows, column-major inside) and the sequence is cut
//     into 8 contiguous chunks, one per XCD (block b runs on XCD b % 8), so each XCD's L2 sees a
//     compact ~8x8 patch of tiles that share A row-panels and B column-panels.
#include "gemm_f32.h"

#include <stdio.h>
#include <stdlib.h>

namespace tfk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int KC_PAD = 4;
constexpr int GROUP_ROWS = 8;
constexpr int NUM_XCD = 8;

template <int BM_, int BN_, int WM_, int WN_, bool A_KC_, bool B_KC_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr bool A_KC = A_KC_, B_KC = B_KC_;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static constexpr int NWAVES = WAVES_M * WAVES_N;
  static constexpr int NT = NWAVES * 64;
  static constexpr int FM = WM / 32, FN = WN / 32;
  // LDS images
  static constexpr int A_LD = A_KC ? (BK + KC_PAD) : BM;
  static constexpr int A_SZ = (A_KC ? BM : BK) * A_LD;
  static constexpr int B_LD = B_KC ? (BK + KC_PAD) : BN;
  static constexpr int B_SZ = (B_KC ? BN : BK) * B_LD;
  static constexpr int STAGE = A_SZ + B_SZ;
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  // global staging: float4 per thread per tile
  static constexpr int A_F4 = BM * BK / 4 / NT;
  static constexpr int B_F4 = BN * BK / 4 / NT;
  static_assert(BM % WM == 0 && BN % WN == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
  static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging divisibility");
};

// Operand tile loader: ROWS_KC = tile extent along m (or n); `kc` selects the memory order.
template <int EXT, int NT, int NF4, bool KC>
__device__ __forceinline__ void load_tile(float4 (&r)[NF4], const float* __restrict__ base, int ld,
                                          int ext0, int ext_lim, int k0, int k_lim, int tid) {
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int idx = tid + i * NT;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      // memory [ext][k]: 8 float4 per 32-k row
      const int e = ext0 + (idx >> 3);
      const int k = k0 + ((idx & 7) << 2);
      if (e < ext_lim && k < k_lim) v = *reinterpret_cast<const float4*>(base + (size_t)e * ld + k);
    } else {
      // memory [k][ext]: EXT/4 float4 per k row
      constexpr int C4 = EXT / 4;
      const int k = k0 + idx / C4;
      const int e = ext0 + ((idx % C4) << 2);
      if (k < k_lim && e < ext_lim) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + e);
    }
    r[i] = v;
  }
}

template <int EXT, int NT, int NF4, bool KC, int LD>
__device__ __forceinline__ void store_tile(const float4 (&r)[NF4], float* __restrict__ s, int tid) {
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int idx = tid + i * NT;
    int off;
    if (KC) {
      off = (idx >> 3) * LD + ((idx & 7) << 2);
    } else {
      constexpr int C4 = EXT / 4;
      off = (idx / C4) * LD + ((idx % C4) << 2);
    }
    *reinterpret_cast<float4*>(s + off) = r[i];
  }
}

// Fragment fetch for one 8-k group g: f[frag][t] is the operand of MFMA step t.
template <int NF, bool KC, int LD>
__device__ __forceinline__ void read_frags(float (&f)[NF][4], const float* __restrict__ s, int ext_base,
                                           int g, int i, int h) {
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    if (KC) {
      const float4 v = *reinterpret_cast<const float4*>(s + (ext_base + q * 32 + i) * LD + g * 8 + h * 4);
      f[q][0] = v.x; f[q][1] = v.y; f[q][2] = v.z; f[q][3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[q][t] = s[(g * 8 + h * 4 + t) * LD + ext_base + q * 32 + i];
    }
  }
}

template <class T>
__global__ void __launch_bounds__(T::NT)
gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int i = lane & 31;
  const int h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WAVES_N;
  const int wn = wave % T::WAVES_N;

  // ---- XCD-aware grouped tile order (bijective for any grid size) ----
  int tm, tn;
  {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid % NUM_XCD, loc = bid / NUM_XCD;
    const int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
    const int seq = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per_group = GROUP_ROWS * tiles_n;
    const int grp = seq / per_group;
    const int first_m = grp * GROUP_ROWS;
    const int gsize = min(GROUP_ROWS, tiles_m - first_m);
    const int within = seq - grp * per_group;
    tm = first_m + within % gsize;
    tn = within / gsize;
  }
  const int m0 = tm * T::BM, n0 = tn * T::BN;

  // limits for the zero-padded float4 accesses
  const int Mp = (p.M + 3) & ~3, Np = (p.N + 3) & ~3, Kp = (p.K + 3) & ~3;
  const int a_ext_lim = T::A_KC ? p.M : Mp;
  const int a_k_lim = T::A_KC ? Kp : p.K;
  const int b_ext_lim = T::B_KC ? p.N : Np;
  const int b_k_lim = T::B_KC ? Kp : p.K;

  f32x16 acc[T::FM][T::FN];
#pragma unroll
  for (int a = 0; a < T::FM; ++a)
#pragma unroll
    for (int b = 0; b < T::FN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[T::A_F4], rb[T::B_F4];
  const int nk = (p.K + BK - 1) / BK;

  load_tile<T::BM, T::NT, T::A_F4, T::A_KC>(ra, p.A, p.lda, m0, a_ext_lim, 0, a_k_lim, tid);
  load_tile<T::BN, T::NT, T::B_F4, T::B_KC>(rb, p.B, p.ldb, n0, b_ext_lim, 0, b_k_lim, tid);
  store_tile<T::BM, T::NT, T::A_F4, T::A_KC, T::A_LD>(ra, smem, tid);
  store_tile<T::BN, T::NT, T::B_F4, T::B_KC, T::B_LD>(rb, smem + T::A_SZ, tid);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {
      const int k0 = (kt + 1) * BK;
      load_tile<T::BM, T::NT, T::A_F4, T::A_KC>(ra, p.A, p.lda, m0, a_ext_lim, k0, a_k_lim, tid);
      load_tile<T::BN, T::NT, T::B_F4, T::B_KC>(rb, p.B, p.ldb, n0, b_ext_lim, k0, b_k_lim, tid);
    }
    const float* As = smem + cur * T::STAGE;
    const float* Bs = As + T::A_SZ;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float fa[T::FM][4], fb[T::FN][4];
      read_frags<T::FM, T::A_KC, T::A_LD>(fa, As, wm * T::WM, g, i, h);
      read_frags<T::FN, T::B_KC, T::B_LD>(fb, Bs, wn * T::WN, g, i, h);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < T::FM; ++a)
#pragma unroll
          for (int b = 0; b < T::FN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][t], fb[b][t], acc[a][b], 0, 0, 0);
    }
    if (more) {
      float* Ad = smem + (cur ^ 1) * T::STAGE;
      store_tile<T::BM, T::NT, T::A_F4, T::A_KC, T::A_LD>(ra, Ad, tid);
      store_tile<T::BN, T::NT, T::B_F4, T::B_KC, T::B_LD>(rb, Ad + T::A_SZ, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: D reg r of lane (i,h) is row (r&3) + 8*(r>>2) + 4*h, col i of its 32x32 fragment ----
  const bool do_bias = (p.epi & EPI_BIAS) != 0;
  const bool do_acc = (p.epi & EPI_ACCUM) != 0;
  const bool do_relu = (p.epi & EPI_RELU) != 0;
#pragma unroll
  for (int b = 0; b < T::FN; ++b) {
    const int col = n0 + wn * T::WN + b * 32 + i;
    if (col >= p.N) continue;
    const float bv = do_bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < T::FM; ++a) {
      const int rbase = m0 + wm * T::WM + a * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          float* dst = p.C + (size_t)row * p.ldc + col;
          float v = acc[a][b][r] + bv;
          if (do_acc) v += *dst;
          if (do_relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
  }
}

template <class T>
int launch(const GemmArgs& p, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    if (T::LDS_BYTES > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_done = true;
  }
  const int tiles_m = (p.M + T::BM - 1) / T::BM;
  const int tiles_n = (p.N + T::BN - 1) / T::BN;
  if (tiles_m <= 0 || tiles_n <= 0) return 0;
  hipLaunchKernelGGL(gemm_f32_kernel<T>, dim3(tiles_m * tiles_n), dim3(T::NT), T::LDS_BYTES, stream, p,
                     tiles_m, tiles_n);
  return (int)hipGetLastError();
}

template <bool A_KC, bool B_KC>
int dispatch_cfg(const GemmArgs& p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 0: return launch<Tile<128, 128, 64, 64, A_KC, B_KC>>(p, s);
    case 1: return launch<Tile<128, 64, 64, 32, A_KC, B_KC>>(p, s);
    case 2: return launch<Tile<64, 128, 32, 64, A_KC, B_KC>>(p, s);
    case 3: return launch<Tile<64, 64, 32, 32, A_KC, B_KC>>(p, s);
    case 4: return launch<Tile<128, 128, 64, 32, A_KC, B_KC>>(p, s);
    case 5: return launch<Tile<256, 128, 64, 64, A_KC, B_KC>>(p, s);
    default: return (int)hipErrorInvalidValue;
  }
}

int g_forced_cfg = -2;  // -2: env not read yet; -1: heuristic

}  // namespace

const char* gemm_f32_config_name(int cfg) {
  static const char* names[kNumGemmConfigs] = {"128x128/4w64x64", "128x64/4w64x32", "64x128/4w32x64",
                                               "64x64/4w32x32",   "128x128/8w64x32", "256x128/8w64x64"};
  return (cfg >= 0 && cfg < kNumGemmConfigs) ? names[cfg] : "?";
}

void gemm_f32_force_config(int cfg) { g_forced_cfg = cfg; }

int gemm_f32_pick_config(GemmLayout layout, int M, int N, int K) {
  (void)layout;
  (void)K;
  // Fill the 256 CUs first, then prefer the larger tile (fewer L2 reads per flop).
  static const int bm[kNumGemmConfigs] = {128, 128, 64, 64, 128, 256};
  static const int bn[kNumGemmConfigs] = {128, 64, 128, 64, 128, 128};
  static const int order[] = {0, 2, 1, 3};
  for (int c : order) {
    const long tiles = (long)((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]);
    if (tiles >= 256) return c;
  }
  return 3;
}

int gemm_f32(GemmLayout layout, const GemmArgs& args, int cfg, hipStream_t stream) {
  if (g_forced_cfg == -2) {
    const char* e = getenv("TFK_GEMM_CFG");
    g_forced_cfg = e ? atoi(e) : -1;
  }
  if (cfg < 0) cfg = (g_forced_cfg >= 0) ? g_forced_cfg : gemm_f32_pick_config(layout, args.M, args.N, args.K);
  if ((args.lda & 3) || (args.ldb & 3) || (args.ldc & 0)) return (int)hipErrorInvalidValue;
  switch (layout) {
    case GEMM_NN: return dispatch_cfg<true, false>(args, cfg, stream);
    case GEMM_NT: return dispatch_cfg<true, true>(args, cfg, stream);
    case GEMM_TN: return dispatch_cfg<false, false>(args, cfg, stream);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace tfk
