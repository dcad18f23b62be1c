// fp32 MFMA GEMM for gfx950 (CDNA4): LDS-tiled and software-pipelined; tiles reach LDS by LDS-DMA (the 64x64
// configuration of the hot path) or through registers (the larger tiles).
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
//   * LDS-DMA pipeline (Tile<..., DMA>): see DmaLoader and the K loop under `if constexpr (T::DMA)`.
//   * Register pipeline (NSTAGE = 3): iteration t writes the register-staged tile t+2 into ring slot (t+2)%3,
//     issues the global loads of tile t+3, and computes tile t.  Tile t+1 has been visible in LDS since
//     the previous barrier, so the fragments of its first k-group are fetched BEFORE this iteration's
//     barrier, under the last MFMAs of tile t: the matrix pipe never waits for an LDS round trip after a
//     barrier.  Fragment registers are double-buffered inside a tile as well.  (NSTAGE = 2 is the plain
//     double buffer for the tiles whose 3-slot ring would not fit two blocks per CU.)
//   * Tiles are walked in a grouped order (8 tile-rows, column-major inside) and the sequence is cut
//     into 8 contiguous chunks, one per XCD (block b runs on XCD b % 8), so each XCD's L2 sees a
//     compact ~8x8 patch of tiles that share A row-panels and B column-panels.
//   * Epilogues are compile-time: the accumulate form loads its 16 C values per fragment as one batch.
#include "gemm_f32.h"

#include <stdio.h>
#include <stdlib.h>

#ifndef TFK_ABL
#define TFK_ABL 0  // tools/gemm_ablate.hip only: timing variants with pieces of the K loop removed
#endif

namespace tfk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int NG = BK / 8;  // k-groups per tile (even: fragment double-buffer parity is tile-invariant)
constexpr int KC_PAD = 4;
constexpr int GROUP_ROWS = 8;
constexpr int NUM_XCD = 8;

// NSTAGE_: LDS ring slots (2 or 3); PF_: global-load prefetch distance in tiles beyond the ring (1 or 2
// register sets in flight; 2 only with the 3-slot ring).
// DMA_: the tiles go from memory to the LDS ring by LDS-DMA (buffer_load ... lds) instead of through registers;
// NSTAGE_ is then the ring depth (4 or 5 slots) and PF_ is unused.
template <int BM_, int BN_, int WM_, int WN_, int NSTAGE_, int PF_, bool A_KC_, bool B_KC_, bool DMA_ = false>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_, PF = PF_;
  static constexpr bool DMA = DMA_;
  static_assert(DMA_ || PF_ == 1 || (PF_ >= 2 && PF_ <= 8 && NSTAGE_ == 3), "prefetch distance");
  static constexpr bool A_KC = A_KC_, B_KC = B_KC_;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static constexpr int NWAVES = WAVES_M * WAVES_N;
  static constexpr int NT = NWAVES * 64;
  static constexpr int FM = WM / 32, FN = WN / 32;
  // Independent accumulator chains per wave.  Back-to-back MFMAs on the SAME accumulator keep the matrix pipe
  // full only if NOTHING is issued between them (an interposed ds_write / load / read costs ~+43 cycles on
  // the dependent MFMA: MI355X_MICROARCH.md, per-instruction constants) -- measured here as 66 % instead of
  // 88 % pipe use on the 32x32 wave tile.  So the four k-steps of a group are spread over KS accumulators
  // (summed in the epilogue) such that every wave owns >= 4 chains and consecutive MFMAs never depend.
  static constexpr int KS = (FM * FN >= 4) ? 1 : (FM * FN == 2 ? 2 : 4);
  // LDS images
  // (a DMA image is lane-linear, so its k-contiguous rows are unpadded and XOR-swizzled instead)
  static constexpr int A_LD = A_KC ? (BK + (DMA ? 0 : KC_PAD)) : BM;
  static constexpr int A_SZ = (A_KC ? BM : BK) * A_LD;
  static constexpr int B_LD = B_KC ? (BK + (DMA ? 0 : KC_PAD)) : BN;
  static constexpr int B_SZ = (B_KC ? BN : BK) * B_LD;
  static constexpr int STAGE = A_SZ + B_SZ;
  static constexpr int LDS_BYTES = NSTAGE * STAGE * 4;
  // global staging: float4 per thread per tile
  static constexpr int A_F4 = BM * BK / 4 / NT;
  static constexpr int B_F4 = BN * BK / 4 / NT;
  static_assert(BM % WM == 0 && BN % WN == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
  static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging divisibility");
  static_assert(DMA ? (NSTAGE >= 3 && NSTAGE <= 5) : (NSTAGE == 2 || NSTAGE == 3), "ring depth");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// Operand tile loader: EXT = tile extent along m (or n); KC selects the memory order.
// Loads go through a buffer resource (hardware bounds check): an out-of-range element is requested at an
// offset beyond num_records and comes back as 0 WITHOUT a branch, so the K-loop stays one basic block and
// the loads / LDS writes can be scheduled into the gaps between MFMAs.
constexpr int kOOB = (int)0x80000000;

template <int EXT, int NT, int NF4, bool KC>
struct TileLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[NF4];   // byte offset of this thread's float4 inside the matrix, k-tile term excluded;
                   // kOOB when its m/n coordinate is outside the matrix
  int kidx[NF4];   // its k coordinate inside a tile
  int kstride;     // bytes per unit of k (4 for KC, 4*ld for MC)
  int k_lim;

  __device__ __forceinline__ void init(const float* base, int ld, int rows, int ext0, int ext_lim, int k_lim_,
                                       int tid) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * ld * 4, 0x00020000);
    k_lim = k_lim_;
    kstride = KC ? 4 : ld * 4;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int idx = tid + i * NT;
      int e, k;
      if (KC) {  // memory [ext][k]: 8 float4 per 32-k row
        e = ext0 + (idx >> 3);
        k = (idx & 7) << 2;
        voff[i] = e < ext_lim ? (e * ld + k) * 4 : kOOB;
      } else {   // memory [k][ext]: EXT/4 float4 per k row
        constexpr int C4 = EXT / 4;
        k = idx / C4;
        e = ext0 + ((idx % C4) << 2);
        voff[i] = e < ext_lim ? (k * ld + e) * 4 : kOOB;
      }
      kidx[i] = k;
    }
  }
  __device__ __forceinline__ void load(float4 (&r)[NF4], int k0) const {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int off = (k0 + kidx[i] < k_lim) ? voff[i] : kOOB;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, k0 * kstride, 0);
      r[i] = *reinterpret_cast<const float4*>(&v);
    }
  }
};

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

// LDS-DMA staging.  One `buffer_load_dwordx4 ... lds` moves 64 x 16 B per wave from per-lane global addresses to
// ONE contiguous KiB of LDS (M0 + lane * 16): no staging registers, no ds_write pass.  The LDS image is therefore
// lane-linear: 16-B chunk q of an operand tile sits at byte q * 16, and any permutation has to be applied on the
// SOURCE side.  k-contiguous tiles ([ext][32 k], 8 chunks per row) store k-chunk c of row r at chunk position
// c ^ ((r >> 1) & 7): the four 16-lane service groups of the fragments' ds_read_b128 ({0-3,12-15,20-27}, ...) then
// touch 16 distinct 16-B slots of the 256-B bank row.  m/n-contiguous tiles ([32 k][ext]) stay linear.
// hipcc does not count these loads (inline asm): completion is the loop's own s_waitcnt vmcnt(N) + barrier.
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int EXT, int NT, int NF4, bool KC>
struct DmaLoader {
  i32x4 rsrc;
  int voff[NF4];
  int kidx[NF4];
  int kstride;
  int k_lim;

  __device__ __forceinline__ void init(const float* base, int ld, int rows, int ext0, int ext_lim, int k_lim_,
                                       int tid) {
    const unsigned long long a = (unsigned long long)base;
    rsrc[0] = (int)(unsigned)a;
    rsrc[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    rsrc[2] = rows * ld * 4;
    rsrc[3] = 0x00020000;
    k_lim = k_lim_;
    kstride = KC ? 4 : ld * 4;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int idx = tid + i * NT;  // chunk index inside the tile image
      int e, k;
      if (KC) {
        const int r = idx >> 3;
        k = ((idx & 7) ^ ((r >> 1) & 7)) << 2;
        e = ext0 + r;
        voff[i] = e < ext_lim ? (e * ld + k) * 4 : kOOB;
      } else {
        constexpr int C4 = EXT / 4;
        k = idx / C4;
        e = ext0 + ((idx % C4) << 2);
        voff[i] = e < ext_lim ? (k * ld + e) * 4 : kOOB;
      }
      kidx[i] = k;
    }
  }
  // piece i of the tile at k0 -> image at LDS byte address `image`
  __device__ __forceinline__ void issue(int i, unsigned image, int k0, int wave) const {
    const int off = (k0 + kidx[i] < k_lim) ? voff[i] : kOOB;
    const unsigned dst = image + (unsigned)(wave * 64 + i * NT) * 16u;
    const int soff = k0 * kstride;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(dst), "v"(off), "s"(rsrc), "s"(soff)
                 : "memory");
  }
};

// Fragment fetch for one 8-k group g: f[frag][t] is the operand of MFMA step t.
template <int NF, bool KC, int LD, bool SWZ = false>
__device__ __forceinline__ void read_frags(float (&f)[NF][4], const float* __restrict__ s, int ext_base,
                                           int g, int i, int h) {
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    if (KC && SWZ) {
      const int r = ext_base + q * 32 + i;
      const float4 v = *reinterpret_cast<const float4*>(s + r * LD + (((g * 2 + h) ^ ((r >> 1) & 7)) << 2));
      f[q][0] = v.x; f[q][1] = v.y; f[q][2] = v.z; f[q][3] = v.w;
    } else if (KC) {
      const float4 v = *reinterpret_cast<const float4*>(s + (ext_base + q * 32 + i) * LD + g * 8 + h * 4);
      f[q][0] = v.x; f[q][1] = v.y; f[q][2] = v.z; f[q][3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[q][t] = s[(g * 8 + h * 4 + t) * LD + ext_base + q * 32 + i];
    }
  }
}

// One output tile: `bid` is the block's index among the tiles_m * tiles_n tiles of THIS contraction.
template <class T, int EPI>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, int tiles_m, int tiles_n, int bid, float* smem) {
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

  f32x16 acc[T::KS][T::FM][T::FN];
#pragma unroll
  for (int q = 0; q < T::KS; ++q)
#pragma unroll
    for (int a = 0; a < T::FM; ++a)
#pragma unroll
      for (int b = 0; b < T::FN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][a][b][r] = 0.f;

  float4 ra[T::PF][T::A_F4], rb[T::PF][T::B_F4];
  float fa[2][T::FM][4], fb[2][T::FN][4];
  const int nk = (p.K + BK - 1) / BK;

  TileLoader<T::BM, T::NT, T::A_F4, T::A_KC> lda_;
  TileLoader<T::BN, T::NT, T::B_F4, T::B_KC> ldb_;
  lda_.init(p.A, p.lda, T::A_KC ? p.M : p.K, m0, a_ext_lim, a_k_lim, tid);
  ldb_.init(p.B, p.ldb, T::B_KC ? p.N : p.K, n0, b_ext_lim, b_k_lim, tid);
#define TFK_LOAD(set, kt)            \
  do {                               \
    lda_.load(ra[set], (kt) * BK);   \
    ldb_.load(rb[set], (kt) * BK);   \
  } while (0)
#define TFK_STORE(set, slot)                                                                    \
  do {                                                                                          \
    store_tile<T::BM, T::NT, T::A_F4, T::A_KC, T::A_LD>(ra[set], (slot), tid);                  \
    store_tile<T::BN, T::NT, T::B_F4, T::B_KC, T::B_LD>(rb[set], (slot) + T::A_SZ, tid);        \
  } while (0)
#define TFK_FRAGS(buf, slot, g)                                                            \
  do {                                                                                     \
    read_frags<T::FM, T::A_KC, T::A_LD, T::DMA>(fa[buf], (slot), wm * T::WM, (g), i, h);           \
    read_frags<T::FN, T::B_KC, T::B_LD, T::DMA>(fb[buf], (slot) + T::A_SZ, wn * T::WN, (g), i, h); \
  } while (0)
#define TFK_MFMA(buf)                                                                                       \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                             \
  _Pragma("unroll") for (int a = 0; a < T::FM; ++a)                                                         \
  _Pragma("unroll") for (int b = 0; b < T::FN; ++b)                                                         \
    acc[t % T::KS][a][b] =                                                                                  \
        __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][a][t], fb[buf][b][t], acc[t % T::KS][a][b], 0, 0, 0)

  if constexpr (T::DMA) {
    // LDS-DMA ring of NS slots.  Iteration kt computes tile kt from slot kt % NS; tile kt+1 has been visible since
    // the previous barrier (its first fragments are fetched before this iteration's barrier, as in the register
    // ring below); tiles kt+2 .. kt+NS-2 are in flight; the pieces of tile kt+NS-1 are issued, one per MFMA gap of
    // the first k-group, into the slot tile kt-1 left at that barrier.  The iteration ends by waiting for this
    // wave's pieces of tile kt+2 (vmcnt: the NS-3 younger tiles may stay in flight) and the barrier that makes
    // every wave's pieces of it visible.
    constexpr int NS = T::NSTAGE;
    constexpr int NPA = T::A_F4, NPT = T::A_F4 + T::B_F4;  // pieces per thread per tile
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
    DmaLoader<T::BM, T::NT, T::A_F4, T::A_KC> da;
    DmaLoader<T::BN, T::NT, T::B_F4, T::B_KC> db;
    da.init(p.A, p.lda, T::A_KC ? p.M : p.K, m0, a_ext_lim, a_k_lim, tid);
    db.init(p.B, p.ldb, T::B_KC ? p.N : p.K, n0, b_ext_lim, b_k_lim, tid);
#define TFK_PIECE(j, slotidx, kt)                                                                  \
  do {                                                                                             \
    if ((j) < NPA) da.issue((j), lds0 + (unsigned)((slotidx) * T::STAGE * 4), (kt) * BK, wave);    \
    else db.issue((j) - NPA, lds0 + (unsigned)(((slotidx) * T::STAGE + T::A_SZ) * 4), (kt) * BK, wave); \
  } while (0)
#define TFK_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#define TFK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory")
#define TFK_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef TFK_DMA_PLACE
#define TFK_DMA_PLACE 2  // 0: every piece in k-group 0; 1: spread over the 4 groups; 2: over groups 0 and 1
#endif
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
#pragma unroll
      for (int j = 0; j < NPT; ++j) TFK_PIECE(j, t, t);
    TFK_VMCNT((NS - 3) * NPT);
    TFK_LDS_BARRIER();
    TFK_FRAGS(0, smem, 0);
#define TFK_DITER(R, KT)                                                                        \
  do {                                                                                          \
    const float* cur = smem + (R) * T::STAGE;                                                   \
    const float* nxt = smem + (((R) + 1) % NS) * T::STAGE;                                      \
    constexpr int WR = ((R) + NS - 1) % NS;                                                     \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                            \
      if (g + 1 < NG) {                                                                         \
        TFK_FRAGS((g + 1) & 1, cur, g + 1);                                                     \
      } else {                                                                                  \
        TFK_FRAGS(0, nxt, 0);                                                                   \
      }                                                                                         \
      TFK_SB();                                                                                 \
      /* pieces [p0, p1) of tile KT+NS-1 ride behind the first MFMAs of this k-group */         \
      const int p0 = TFK_DMA_PLACE == 0 ? (g == 0 ? 0 : NPT) : g * NPT / (NG / TFK_DMA_PLACE);  \
      const int p1 = TFK_DMA_PLACE == 0 ? NPT : min(NPT, (g + 1) * NPT / (NG / TFK_DMA_PLACE)); \
      int pc = p0;                                                                              \
      _Pragma("unroll") for (int t = 0; t < 4; ++t)                                             \
      _Pragma("unroll") for (int a = 0; a < T::FM; ++a)                                         \
      _Pragma("unroll") for (int b = 0; b < T::FN; ++b) {                                       \
        acc[t % T::KS][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(                            \
            fa[g & 1][a][t], fb[g & 1][b][t], acc[t % T::KS][a][b], 0, 0, 0);                   \
        if (pc < p1 && !(TFK_ABL & 1)) {                                                        \
          TFK_SB();                                                                             \
          TFK_PIECE(pc, WR, (KT) + NS - 1);                                                     \
          TFK_SB();                                                                             \
        }                                                                                       \
        ++pc;                                                                                   \
      }                                                                                         \
      _Pragma("unroll") for (; pc < p1; ++pc) TFK_PIECE(pc, WR, (KT) + NS - 1);                 \
      TFK_SB();                                                                                 \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (!(TFK_ABL & 16)) TFK_VMCNT((NS - 3) * NPT);                                             \
    if (!(TFK_ABL & 8)) TFK_LDS_BARRIER();                                                      \
  } while (0)
    for (int kt = 0; kt < nk; kt += NS) {
      TFK_DITER(0, kt);
      TFK_DITER(1, kt + 1);
      TFK_DITER(2, kt + 2);
      if (NS >= 4) TFK_DITER(3 % NS, kt + 3);
      if (NS >= 5) TFK_DITER(4 % NS, kt + 4);
    }
    // the epilogues reuse the ring as scratch: nothing may still be landing in it
    TFK_VMCNT(0);
    TFK_LDS_BARRIER();
#undef TFK_DITER
#undef TFK_PIECE
  } else if (T::NSTAGE == 3) {
    // ring slots as rotating pointers: s0 = tile t, s1 = tile t+1, s2 = tile t+2
    float* s0 = smem;
    float* s1 = smem + T::STAGE;
    float* s2 = smem + 2 * T::STAGE;
    // Prologue (unconditional: tiles past the end of K load as zeros): tiles 0 and 1 go to the ring, tiles
    // 2 .. 1+PF stay in the register sets until their iteration stores them.
    if (T::PF >= 2) {  // both ring tiles in flight at once: one global round trip instead of two
      TFK_LOAD(0, 0);
      TFK_LOAD(1 % T::PF, 1);
      TFK_STORE(0, s0);
      TFK_STORE(1 % T::PF, s1);
    } else {
      TFK_LOAD(0, 0);
      TFK_STORE(0, s0);
      TFK_LOAD(0, 1);
      TFK_STORE(0, s1);
    }
    TFK_LOAD(0, 2);
    if (T::PF >= 2) TFK_LOAD(1 % T::PF, 3);
    if (T::PF >= 3) TFK_LOAD(2 % T::PF, 4);
    if (T::PF >= 4) TFK_LOAD(3 % T::PF, 5);
    if (T::PF >= 5) TFK_LOAD(4 % T::PF, 6);
    if (T::PF >= 6) TFK_LOAD(5 % T::PF, 7);
    if (T::PF >= 7) TFK_LOAD(6 % T::PF, 8);
    if (T::PF >= 8) TFK_LOAD(7 % T::PF, 9);
    __syncthreads();
    TFK_FRAGS(0, s0, 0);
    // One iteration = ONE basic block.  Schedule pins (sched_barrier): every k-group first issues the NEXT
    // group's fragment reads, then its four MFMAs; the ring-slot write of tile kt+2 rides in the gaps of
    // group 0 and the global loads of tile kt+2+PF follow at once into the register set just freed, so their
    // data has PF full MFMA blocks (PF x ~1000 cycles for a 32x32 wave tile) to arrive.  Past the end of K
    // the loads return zeros (kOOB) and the MFMAs of a rounded-up trip count add zeros.
// TFK_ABL (tools/gemm_ablate.sh only): timing-only variants with pieces of the loop removed --
// 1 global loads, 2 ring-slot writes, 4 fragment reads, 8 barrier.  Results are then meaningless.
// Instruction placement inside one iteration (sched_group_barrier = "emit N instructions of this class
// here"): an MFMA occupies the SIMD's matrix pipe for 64 cycles, and whatever the wave issues before its
// next MFMA must fit in that gap or the pipe idles.  Measured (tools/gemm_ablate.sh): four back-to-back
// ds_write_b128 (13+ issue cycles each plus LDS data-FIFO back-pressure) in one gap cost 20 % of the kernel;
// so group 0 interleaves ONE ring-slot write per MFMA, group 1 ONE global load per MFMA, and every group
// starts with the fragment reads of the group after it.
#define TFK_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define TFK_ITER(SET, KT)                                                   \
  do {                                                                      \
    constexpr int MFG = 4 * T::FM * T::FN;          /* MFMAs per k-group */  \
    constexpr int NST = T::A_F4 + T::B_F4;          /* staging float4 per thread */ \
    constexpr int NIL = NST < MFG ? NST : MFG;                              \
    if (!(TFK_ABL & 4)) TFK_FRAGS(1, s0, 1);                                \
    TFK_MFMA(0);                                                            \
    if (!(TFK_ABL & 2)) TFK_STORE(SET, s2);                                 \
    TFK_SGB(0x100, 16);                             /* DS_READ */            \
    _Pragma("unroll") for (int j = 0; j < NIL; ++j) {                       \
      TFK_SGB(0x008, 1);                            /* MFMA */               \
      TFK_SGB(0x200, 1);                            /* DS_WRITE */           \
    }                                                                       \
    TFK_SGB(0x200, 8);                                                      \
    TFK_SGB(0x008, MFG);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                      \
    if (!(TFK_ABL & 4)) TFK_FRAGS(0, s0, 2);                                \
    if (!(TFK_ABL & 1)) TFK_LOAD(SET, (KT) + 2 + T::PF);                    \
    TFK_MFMA(1);                                                            \
    TFK_SGB(0x100, 16);                                                     \
    _Pragma("unroll") for (int j = 0; j < NIL; ++j) {                       \
      TFK_SGB(0x008, 1);                                                    \
      TFK_SGB(0x020, 1);                            /* VMEM_READ */          \
    }                                                                       \
    TFK_SGB(0x020, 8);                                                      \
    TFK_SGB(0x008, MFG);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                      \
    _Pragma("unroll") for (int g = 2; g < NG; ++g) {                        \
      if (!(TFK_ABL & 4)) {                                                 \
        if (g + 1 < NG) {                                                   \
          TFK_FRAGS((g + 1) & 1, s0, g + 1);                                \
        } else {                                                            \
          TFK_FRAGS(0, s1, 0); /* next tile: visible since the last barrier */ \
        }                                                                   \
      }                                                                     \
      TFK_MFMA(g & 1);                                                      \
      TFK_SGB(0x100, 16);                                                   \
      TFK_SGB(0x008, MFG);                                                  \
      __builtin_amdgcn_sched_barrier(0);                                    \
    }                                                                       \
    if (!(TFK_ABL & 8)) __syncthreads();                                    \
    float* tmp = s0; s0 = s1; s1 = s2; s2 = tmp;                            \
  } while (0)
    // the register set index must be a literal: unroll the tile loop by PF (a rounded-up trip count only
    // adds MFMAs on all-zero tiles)
    for (int kt = 0; kt < nk; kt += T::PF) {
      TFK_ITER(0, kt);
      if (T::PF >= 2) TFK_ITER(1 % T::PF, kt + 1);
      if (T::PF >= 3) TFK_ITER(2 % T::PF, kt + 2);
      if (T::PF >= 4) TFK_ITER(3 % T::PF, kt + 3);
      if (T::PF >= 5) TFK_ITER(4 % T::PF, kt + 4);
      if (T::PF >= 6) TFK_ITER(5 % T::PF, kt + 5);
      if (T::PF >= 7) TFK_ITER(6 % T::PF, kt + 6);
      if (T::PF >= 8) TFK_ITER(7 % T::PF, kt + 7);
    }
#undef TFK_ITER
  } else {
    float* s0 = smem;
    float* s1 = smem + T::STAGE;
    TFK_LOAD(0, 0);
    TFK_STORE(0, s0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      TFK_LOAD(0, kt + 1);  // pinned first: its latency hides under the whole MFMA block
      __builtin_amdgcn_sched_barrier(0);
      TFK_FRAGS(0, s0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) TFK_FRAGS((g + 1) & 1, s0, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        TFK_MFMA(g & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      TFK_STORE(0, s1);
      __syncthreads();
      float* tmp = s0; s0 = s1; s1 = tmp;
    }
  }
#undef TFK_LOAD
#undef TFK_STORE
#undef TFK_FRAGS
#undef TFK_MFMA

  // ---- epilogue: D reg r of lane (i,h) is row (r&3) + 8*(r>>2) + 4*h, col i of its 32x32 fragment ----
  const bool full_tile = (m0 + T::BM <= p.M) && (n0 + T::BN <= p.N);
  if constexpr ((EPI & EPI_COLSTATS) != 0) {
    // Batch-norm statistics of this tile's rows, two-pass (mean, then squared deviations) so that the later
    // Chan merge over tile rows never subtracts large numbers.  A lane holds 16 rows of one column per fragment,
    // its partner lane ^ 32 the interleaved other 16; the WAVES_M waves stacked along m meet in LDS (the K loop
    // has ended behind a barrier, so the ring is free).
    float* red = smem;  // [WAVES_M][BN]
    const int vlim = p.row_vend ? min(p.M, p.row_vend[m0 >> 6]) : p.M;  // (stacked pass: rows of this tile's segment)
    const int n_tile = max(1, min(T::BM, vlim - m0));
    float cmean[T::FN];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int b = 0; b < T::FN; ++b) {
        const int colc = min(n0 + wn * T::WN + b * 32 + i, p.N - 1);
        const float bv = (EPI & EPI_BIAS) ? p.bias[colc] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < T::FM; ++a) {
          const int rbase = m0 + wm * T::WM + a * 32 + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[0][a][b][r];
#pragma unroll
            for (int q = 1; q < T::KS; ++q) v += acc[q][a][b][r];
            v += bv;
            if (rbase + (r & 3) + 8 * (r >> 2) < vlim) s += pass == 0 ? v : (v - cmean[b]) * (v - cmean[b]);
          }
        }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wm * T::BN + wn * T::WN + b * 32 + i] = s;
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < T::FN; ++b) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < T::WAVES_M; ++w) t += red[w * T::BN + wn * T::WN + b * 32 + i];
        if (pass == 0) {
          cmean[b] = t / (float)n_tile;
        } else if (wm == 0 && h == 0) {
          const int col = n0 + wn * T::WN + b * 32 + i;
          if (col < p.N) {
            p.stats[((size_t)0 * tiles_m + tm) * p.ldc + col] = cmean[b];
            p.stats[((size_t)1 * tiles_m + tm) * p.ldc + col] = t;
          }
        }
      }
      __syncthreads();
    }
  }
  if constexpr ((EPI & EPI_DACT) != 0) {
    // The result is da of a hidden layer: turn it into du = da * f'(a) in the accumulators and reduce, over this
    // tile's rows, the two column sums of batch-norm's backward (same lane / wave reduction as above).
    float* red = smem;  // [2][WAVES_M][BN]
#pragma unroll
    for (int b = 0; b < T::FN; ++b) {
      const int col = n0 + wn * T::WN + b * 32 + i;
      const int colc = min(col, p.N - 1);
      const float mu = p.act_mean[colc], rsd = p.act_rstd[colc];
      // ReLU chains: du is non-zero only where a > 0, and there the normalised pre-activation is a * keep - beta
      // (a = relu(xhat + beta) * mask / keep): the second statistic needs no read of z at all -- half of this
      // epilogue's traffic.  Other nonlinearities read z and use (z - mean) * rstd.
      const bool from_a = p.act_nonlin == 0 && p.act_beta != nullptr;
      const float be = from_a ? p.act_beta[colc] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < T::FM; ++a) {
        const int rbase = m0 + wm * T::WM + a * 32 + 4 * h;
        float av[16], zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
          av[r] = p.act_a[(size_t)row * p.ldc + colc];
        }
        if (!from_a) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
            zv[r] = p.act_z[(size_t)row * p.ldc + colc];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[0][a][b][r];
#pragma unroll
          for (int q = 1; q < T::KS; ++q) {
            v += acc[q][a][b][r];
            acc[q][a][b][r] = 0.f;
          }
          float d1;  // f' through the output, as kernels.hip nonlin_bwd
          switch (p.act_nonlin) {
            case 0: d1 = av[r] > 0.f ? p.act_scale : 0.f; break;  // (a > 0) / keep behind dropout, else 1
            case 1: d1 = av[r] * (1.f - av[r]); break;
            case 2: d1 = 1.f - av[r] * av[r]; break;
            default: d1 = 1.f;
          }
          const float du = v * d1;
          acc[0][a][b][r] = du;
          if (rbase + (r & 3) + 8 * (r >> 2) < p.M) {
            s1 += du;
            s2 += du * (from_a ? av[r] * p.act_keep - be : (zv[r] - mu) * rsd);
          }
        }
      }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0) {
        red[(0 * T::WAVES_M + wm) * T::BN + wn * T::WN + b * 32 + i] = s1;
        red[(1 * T::WAVES_M + wm) * T::BN + wn * T::WN + b * 32 + i] = s2;
      }
    }
    __syncthreads();
    if (wm == 0 && h == 0) {
#pragma unroll
      for (int b = 0; b < T::FN; ++b) {
        const int col = n0 + wn * T::WN + b * 32 + i;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < T::WAVES_M; ++w) {
          t1 += red[(0 * T::WAVES_M + w) * T::BN + wn * T::WN + b * 32 + i];
          t2 += red[(1 * T::WAVES_M + w) * T::BN + wn * T::WN + b * 32 + i];
        }
        if (col < p.N) {
          p.stats[((size_t)0 * p.stats_stride + tm) * p.ldc + col] = t1;
          p.stats[((size_t)1 * p.stats_stride + tm) * p.ldc + col] = t2;
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < T::FN; ++b) {
    const int col = n0 + wn * T::WN + b * 32 + i;
    const bool col_ok = col < p.N;
    const int colc = col_ok ? col : p.N - 1;
    float bv = 0.f;
    if (EPI & EPI_BIAS) bv = p.bias[colc];
    // EPI_EVAL_ACT: evaluation-mode batch norm + nonlinearity of the layer, the operations of bn_stats_eval +
    // act_forward (kernels.hip) in their order, so the result is what the three-launch path stores
    bool ev_bn = false;
    float ev_mu = 0.f, ev_rs = 1.f, ev_be = 0.f;
    if constexpr ((EPI & EPI_EVAL_ACT) != 0) {
      ev_bn = p.act_mean != nullptr;
      if (ev_bn) {
        ev_mu = p.act_mean[colc];
        ev_rs = rsqrtf(p.act_rstd[colc] + p.bn_eps);
        ev_be = p.act_beta[colc];
      }
    }
#pragma unroll
    for (int a = 0; a < T::FM; ++a) {
      const int rbase = m0 + wm * T::WM + a * 32 + 4 * h;
      float old[16];
      if (EPI & EPI_ACCUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = rbase + (r & 3) + 8 * (r >> 2);
          row = row < p.M ? row : p.M - 1;  // clamped address: one batch of 16 loads, no branches
          old[r] = p.C[(size_t)row * p.ldc + colc];
        }
      }
      float outv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[0][a][b][r];
#pragma unroll
        for (int q = 1; q < T::KS; ++q) v += acc[q][a][b][r];
        v += bv;
        if (EPI & EPI_ACCUM) v += old[r];
        if (EPI & EPI_RELU) v = fmaxf(v, 0.f);
        if constexpr ((EPI & EPI_EVAL_ACT) != 0) {
          if (ev_bn) v = (v - ev_mu) * ev_rs + ev_be;
          switch (p.act_nonlin) {
            case 0: v = fmaxf(v, 0.f); break;
            case 1: v = 1.f / (1.f + expf(-v)); break;
            case 2: v = tanhf(v); break;
            default: break;
          }
        }
        outv[r] = v;
      }
      if (full_tile) {  // block-uniform: straight-line stores, no per-element exec masking
#pragma unroll
        for (int r = 0; r < 16; ++r)
          p.C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * p.ldc + col] = outv[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (col_ok && row < p.M) p.C[(size_t)row * p.ldc + col] = outv[r];
        }
      }
    }
  }
}

template <class T, int EPI>
__global__ void __launch_bounds__(T::NT)
gemm_f32_kernel(GemmArgs p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (p.nsplit > 1) {  // split-K: this block contracts k in [k0, k0 + ksplit) into its own partial result
    const int k0 = blockIdx.y * p.ksplit;
    p.A += T::A_KC ? (size_t)k0 : (size_t)k0 * p.lda;
    p.B += T::B_KC ? (size_t)k0 : (size_t)k0 * p.ldb;
    p.K = min(p.ksplit, p.K - k0);
    p.C += (size_t)blockIdx.y * p.split_stride;
  }
  gemm_tile<T, EPI>(p, tiles_m, tiles_n, blockIdx.x, smem);
}

// Two INDEPENDENT contractions in one launch (backward: dA of a layer and the dW that consumes the same dZ): the
// blocks of the second start on a CU the moment a block of the first retires, so one kernel's drain (epilogue
// stores, straggling CUs) overlaps the other's ramp-up instead of idling the matrix pipes.
template <class T1, int EPI1, class T2, int EPI2>
__global__ void __launch_bounds__(T1::NT)
gemm_f32_dual_kernel(GemmArgs p1, GemmArgs p2, int tiles_m1, int tiles_n1, int tiles_m2, int tiles_n2) {
  static_assert(T1::NT == T2::NT, "both halves use the same block size");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n1 = tiles_m1 * tiles_n1;
  if ((int)blockIdx.x < n1) gemm_tile<T1, EPI1>(p1, tiles_m1, tiles_n1, blockIdx.x, smem);
  else gemm_tile<T2, EPI2>(p2, tiles_m2, tiles_n2, blockIdx.x - n1, smem);
}

int g_min_lds = 0;        // env TFK_GEMM_MIN_LDS (experiments): lower bound of the LDS request
int g_even_spread = 1;    // env TFK_GEMM_EVEN_SPREAD=0 disables the residency cap below

template <class T, int EPI>
int launch(const GemmArgs& p, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<T, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int tiles_m = (p.M + T::BM - 1) / T::BM;
  const int tiles_n = (p.N + T::BN - 1) / T::BN;
  if (tiles_m <= 0 || tiles_n <= 0) return 0;
  // The LDS request doubles as the residency limit (blocks per CU = floor(160 KiB / request)): when the grid
  // is a small multiple of the 256 CUs the blocks must spread evenly, or the CU that received an extra
  // block sets the kernel time.
  int lds = T::LDS_BYTES;
  const int tiles = tiles_m * tiles_n;
  const int per_cu = (tiles * p.nsplit + 255) / 256;  // blocks per CU of an even spread
  if (g_even_spread && per_cu <= 4) {
    const int cap = (160 * 1024 / per_cu) & ~1023;
    const int floor_next = (160 * 1024 / (per_cu + 1) / 1024 + 1) * 1024;  // just too big for per_cu + 1 blocks
    if (lds < floor_next && floor_next <= cap) lds = floor_next;
  }
  if (g_min_lds > lds) lds = g_min_lds;
  if (lds > 160 * 1024) lds = 160 * 1024;
  hipLaunchKernelGGL((gemm_f32_kernel<T, EPI>), dim3(tiles, p.nsplit), dim3(T::NT), lds, stream, p, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

struct CfgDesc {
  int bm, bn;
  const char* name;
};
const CfgDesc kCfg[kNumGemmConfigs] = {
    {128, 128, "128x128/4w64x64/s2"},   {128, 64, "128x64/4w64x32/s3p2"},   {64, 128, "64x128/4w32x64/s3p2"},
    {64, 64, "64x64/4w32x32/s3p6|4|3"},     {128, 128, "128x128/8w64x32/s3p1"}, {256, 128, "256x128/8w64x64/s2"},
    {64, 128, "64x128/8w32x32/s3p2"},   {128, 64, "128x64/8w32x32/s3p2"},   {64, 64, "64x64/4w32x32/s3p1"},
    {64, 64, "64x64/4w32x32/dma4"},     {128, 64, "128x64/4w64x32/dma4"},   {64, 128, "64x128/4w32x64/dma4"},
    {128, 128, "128x128/4w64x64/dma4"},
};

// Prefetch distance of the 64x64 configuration per layout (stand-alone 1024x2048x2048 / 2048x2048x1024, same box):
// NN 4 -> 6: 125.1 -> 128.2 TF;  NT 4 (6: 131.9 -> 131.1);  TN 2 -> 3: 121.9 -> 126.4 TF
template <bool A_KC, bool B_KC>
constexpr int kPf3 = A_KC ? (B_KC ? 4 : 6) : 3;

template <bool A_KC, bool B_KC, int EPI>
int dispatch_cfg(const GemmArgs& p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 0: return launch<Tile<128, 128, 64, 64, 2, 1, A_KC, B_KC>, EPI>(p, s);
    case 1: return launch<Tile<128, 64, 64, 32, 3, 2, A_KC, B_KC>, EPI>(p, s);
    case 2: return launch<Tile<64, 128, 32, 64, 3, 2, A_KC, B_KC>, EPI>(p, s);
    // k-contiguous operands (NN / NT) arrive later than m/n-contiguous ones: 4 tiles of prefetch vs 2
    // (measured +3..5 %, profiles/r01_gemm_ablation.txt)
    case 3: return launch<Tile<64, 64, 32, 32, 3, kPf3<A_KC, B_KC>, A_KC, B_KC>, EPI>(p, s);
    case 4: return launch<Tile<128, 128, 64, 32, 3, 1, A_KC, B_KC>, EPI>(p, s);
    case 5: return launch<Tile<256, 128, 64, 64, 2, 1, A_KC, B_KC>, EPI>(p, s);
    case 6: return launch<Tile<64, 128, 32, 32, 3, 2, A_KC, B_KC>, EPI>(p, s);
    case 7: return launch<Tile<128, 64, 32, 32, 3, 2, A_KC, B_KC>, EPI>(p, s);
    case 8: return launch<Tile<64, 64, 32, 32, 3, 1, A_KC, B_KC>, EPI>(p, s);
    case 9: return launch<Tile<64, 64, 32, 32, 4, 1, A_KC, B_KC, true>, EPI>(p, s);
    case 10: return launch<Tile<128, 64, 64, 32, 4, 1, A_KC, B_KC, true>, EPI>(p, s);
    case 11: return launch<Tile<64, 128, 32, 64, 4, 1, A_KC, B_KC, true>, EPI>(p, s);
    case 12: return launch<Tile<128, 128, 64, 64, 4, 1, A_KC, B_KC, true>, EPI>(p, s);
    default: return (int)hipErrorInvalidValue;
  }
}

// The epilogues each layout is used with (anything else is rejected): keeps the kernel count at 6 x configs.
int dispatch_epi(GemmLayout layout, const GemmArgs& p, int cfg, hipStream_t s) {
  switch (layout) {
    case GEMM_NN:
      switch (p.epi) {
        case 0: return dispatch_cfg<true, false, 0>(p, cfg, s);
        case EPI_BIAS: return dispatch_cfg<true, false, EPI_BIAS>(p, cfg, s);
        case EPI_BIAS | EPI_RELU: return dispatch_cfg<true, false, EPI_BIAS | EPI_RELU>(p, cfg, s);
        case EPI_BIAS | EPI_COLSTATS: return dispatch_cfg<true, false, EPI_BIAS | EPI_COLSTATS>(p, cfg, s);
        case EPI_BIAS | EPI_EVAL_ACT: return dispatch_cfg<true, false, EPI_BIAS | EPI_EVAL_ACT>(p, cfg, s);
      }
      break;
    case GEMM_NT:
      if (p.epi == 0) return dispatch_cfg<true, true, 0>(p, cfg, s);
      if (p.epi == EPI_DACT) return dispatch_cfg<true, true, EPI_DACT>(p, cfg, s);
      break;
    case GEMM_TN:
      if (p.epi == 0) return dispatch_cfg<false, false, 0>(p, cfg, s);
      if (p.epi == EPI_ACCUM) return dispatch_cfg<false, false, EPI_ACCUM>(p, cfg, s);
      break;
  }
  return (int)hipErrorInvalidValue;
}

template <int EPI_NT, int EPI_TN, bool DMA>
int launch_dual(const GemmArgs& a, const GemmArgs& w, hipStream_t stream) {
  // config 3 (register-staged ring) or config 9 (LDS-DMA ring) for both halves
  typedef Tile<64, 64, 32, 32, DMA ? 4 : 3, DMA ? 1 : kPf3<true, true>, true, true, DMA> TA;      // NT
  typedef Tile<64, 64, 32, 32, DMA ? 4 : 3, DMA ? 1 : kPf3<false, false>, false, false, DMA> TW;  // TN
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_dual_kernel<TA, EPI_NT, TW, EPI_TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int tma = (a.M + 63) / 64, tna = (a.N + 63) / 64, tmw = (w.M + 63) / 64, tnw = (w.N + 63) / 64;
  int lds = TA::LDS_BYTES > TW::LDS_BYTES ? TA::LDS_BYTES : TW::LDS_BYTES;
  const int floor3 = (160 * 1024 / 3 / 1024 + 1) * 1024;  // two blocks per CU, as the single launches run
  if (g_even_spread && lds < floor3) lds = floor3;
  if (g_min_lds > lds) lds = g_min_lds;
  hipLaunchKernelGGL((gemm_f32_dual_kernel<TA, EPI_NT, TW, EPI_TN>), dim3(tma * tna + tmw * tnw), dim3(TA::NT), lds,
                     stream, a, w, tma, tna, tmw, tnw);
  return (int)hipGetLastError();
}

// C (+)= sum over the split-K partials, in chunk order
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int nsplit, size_t stride, float* __restrict__ C, size_t n4,
                     int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
  for (int k = 1; k < nsplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + 4 * i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (accumulate) {
    const float4 c = *reinterpret_cast<const float4*>(C + 4 * i);
    s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
  }
  *reinterpret_cast<float4*>(C + 4 * i) = s;
}

int g_forced_cfg = -2;  // -2: env not read yet; -1: heuristic
int g_dma = 1;          // env TFK_GEMM_DMA=0: the 64x64 tile stages through registers (config 3), not LDS-DMA (9)

}  // namespace

const char* gemm_f32_config_name(int cfg) {
  return (cfg >= 0 && cfg < kNumGemmConfigs) ? kCfg[cfg].name : "?";
}

int gemm_f32_config_bm(int cfg) { return (cfg >= 0 && cfg < kNumGemmConfigs) ? kCfg[cfg].bm : 0; }

void gemm_f32_force_config(int cfg) { g_forced_cfg = cfg; }

int gemm_f32_pick_config(GemmLayout layout, int M, int N, int K) {
  if (g_forced_cfg == -2) {
    const char* e = getenv("TFK_GEMM_CFG");
    g_forced_cfg = e ? atoi(e) : -1;
    if ((e = getenv("TFK_GEMM_MIN_LDS"))) g_min_lds = atoi(e);
    if ((e = getenv("TFK_GEMM_EVEN_SPREAD"))) g_even_spread = atoi(e);
    if ((e = getenv("TFK_GEMM_DMA"))) g_dma = atoi(e);
  }
  if (g_forced_cfg >= 0) return g_forced_cfg;
  {  // (debugging) TFK_GEMM_CFG_NN / _NT / _TN: force the configuration of one layout only
    static int per_layout[3] = {-2, -2, -2};
    if (per_layout[0] == -2) {
      const char* names[3] = {"TFK_GEMM_CFG_NN", "TFK_GEMM_CFG_NT", "TFK_GEMM_CFG_TN"};
      for (int l = 0; l < 3; ++l) {
        const char* e = getenv(names[l]);
        per_layout[l] = e ? atoi(e) : -1;
      }
    }
    if (per_layout[(int)layout] >= 0) return per_layout[(int)layout];
  }
  (void)K;
  // Measured on MI355X (profiles/r01_gemm_sweep_v3.txt): the 128x128 tile (4 waves of 64x64: half the LDS
  // staging per MFMA of a 64x64 tile) wins once it yields two blocks per CU; below that the 64x64 tile with
  // its 3-slot ring and 2-tile prefetch (two or more independent blocks per CU) is fastest for all layouts.
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  // The 64x64 tile stages through LDS-DMA (config 9) unless TFK_GEMM_DMA=0 asks for the register ring (config 3):
  // stand-alone +0.5..1.5 %, BASELINE cfg2 step +1.4 % (profiles/r01_gemm_dma.txt).
  // Round 2 (profiles/r02_gemm_f32_sweep.txt): the 128x128 tile on the LDS-DMA ring (config 12) matches or beats its
  // register-staged twin (config 0) on every cfg4-size contraction (NT 140 vs 136 TF); 128x64 / 64x128 DMA tiles
  // (10, 11) do not beat 64x64 on the 1024-frame shapes.
  return tiles128 >= 512 ? (g_dma ? 12 : 0) : (g_dma ? 9 : 3);
}

// shortest contraction that is cut (env TFK_SPLITK_MIN_K; two launches -- the partial GEMM and the reduction -- must beat one)
int gemm_f32_splitk_min_k() {
  static const int v = [] { const char* q = getenv("TFK_SPLITK_MIN_K"); const int x = q ? atoi(q) : 2048; return x >= 1024 ? x : 1024; }();
  return v;
}

int gemm_f32(GemmLayout layout, const GemmArgs& args, int cfg, hipStream_t stream) {
  if (cfg < 0) cfg = gemm_f32_pick_config(layout, args.M, args.N, args.K);
  if ((args.lda & 3) || (args.ldb & 3)) return (int)hipErrorInvalidValue;
  if (args.M <= 0 || args.N <= 0 || args.K <= 0) return (int)hipErrorInvalidValue;
  {  // operands are addressed with 32-bit byte offsets through buffer resources
    const long a_rows = layout == GEMM_TN ? args.K : args.M;
    const long b_rows = layout == GEMM_NT ? args.N : args.K;
    if (a_rows * args.lda * 4 >= (1L << 31) || b_rows * args.ldb * 4 >= (1L << 31)) return (int)hipErrorInvalidValue;
  }
  // split-K for chip-starved, long contractions (only the plain / accumulating epilogues)
  const int bm = kCfg[cfg].bm, bn = kCfg[cfg].bn;
  const int tiles = ((args.M + bm - 1) / bm) * ((args.N + bn - 1) / bn);
  if (args.splitk_ws && (args.epi == 0 || args.epi == EPI_ACCUM) && tiles < 256 && args.K >= gemm_f32_splitk_min_k()) {
    int nsplit = (512 + tiles - 1) / tiles;
    if (nsplit > args.K / 512) nsplit = args.K / 512;
    if (nsplit > 32) nsplit = 32;
    const size_t stride = (size_t)args.M * args.ldc;
    if (nsplit > 1 && stride * nsplit <= args.splitk_ws_floats) {
      GemmArgs q = args;
      q.ksplit = ((args.K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
      q.nsplit = (args.K + q.ksplit - 1) / q.ksplit;
      q.split_stride = stride;
      q.C = args.splitk_ws;
      q.epi = 0;
      const int rc = dispatch_epi(layout, q, cfg, stream);
      if (rc != 0) return rc;
      const size_t n4 = stride / 4;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, args.splitk_ws,
                         q.nsplit, stride, args.C, n4, (args.epi & EPI_ACCUM) ? 1 : 0);
      return (int)hipGetLastError();
    }
  }
  return dispatch_epi(layout, args, cfg, stream);
}

int gemm_f32_dual(const GemmArgs& nt, const GemmArgs& tn, hipStream_t stream) {
  // only the shapes the heuristic gives the 64x64 configuration to, and no split-K
  const int small = gemm_f32_pick_config(GEMM_NT, nt.M, nt.N, nt.K);
  if ((small != 3 && small != 9) || gemm_f32_pick_config(GEMM_TN, tn.M, tn.N, tn.K) != small) return -1;
  if (((tn.M + 63) / 64) * ((tn.N + 63) / 64) < 256 && tn.K >= 2048) return -1;  // that one wants split-K
  if ((nt.lda & 3) || (nt.ldb & 3) || (tn.lda & 3) || (tn.ldb & 3)) return (int)hipErrorInvalidValue;
  const int key = (nt.epi == EPI_DACT ? 2 : nt.epi == 0 ? 0 : -8) + (tn.epi == EPI_ACCUM ? 1 : tn.epi == 0 ? 0 : -8);
  switch (key + (small == 9 ? 4 : 0)) {
    case 0: return launch_dual<0, 0, false>(nt, tn, stream);
    case 1: return launch_dual<0, EPI_ACCUM, false>(nt, tn, stream);
    case 2: return launch_dual<EPI_DACT, 0, false>(nt, tn, stream);
    case 3: return launch_dual<EPI_DACT, EPI_ACCUM, false>(nt, tn, stream);
    case 4: return launch_dual<0, 0, true>(nt, tn, stream);
    case 5: return launch_dual<0, EPI_ACCUM, true>(nt, tn, stream);
    case 6: return launch_dual<EPI_DACT, 0, true>(nt, tn, stream);
    case 7: return launch_dual<EPI_DACT, EPI_ACCUM, true>(nt, tn, stream);
  }
  return -1;
}

}  // namespace tfk
