// Memory and LDS layouts of the fp32-emulating contraction ("bf16x3", gemm_bf16.h) -- index arithmetic only, shared by the
// kernels (gemm_bf16.hip, kernels.hip) and by a host-side checker (tools/x3_layout_check.cpp, runs without a GPU).
//
// MEMORY.  An fp32 array [rows, ld] (ld a multiple of 8) is held as three bf16 planes p0 + p1 + p2 = x exactly, INTERLEAVED per
// 32 consecutive elements of the flat index i = row * ld + col:
//       element i, plane q  at  (i >> 5) * 96 + q * 32 + (i & 31)                       (bf16 elements from the array's base)
// so the 32-k row segment a ring slot stages is 192 contiguous bytes (ld a multiple of 32) instead of three 64-byte pieces
// 2 * rows * ld bytes apart, and an element-wise producer writes its three planes into one 192-byte neighbourhood.  With
// separate planes (round 4) a k-contiguous operand filled LDS at ~21 B/clk/CU -- every 64-byte piece pulls a 128-byte line
// through the L1-miss path -- against ~36 for 256-byte rows (profiles/r04_gemm_f32x3_ablation.txt, r03_lds_fill_paths.txt).
// The interleave is a function of the FLAT index: the optimiser writes the shadow of the weight matrices straight from the
// arena offset (no row / column arithmetic), and a matrix whose leading dimension is not a multiple of 32 (2000 pdfs) still
// works -- its odd rows just start in the middle of a block.
//
// LDS (one ring slot = 32 k of all three planes of both operands; every image is written lane-linearly by
// `buffer_load_dwordx4 ... lds`, 16-byte chunk n of an image at byte 16 n, so the permutations below are applied on the SOURCE
// side of the DMA and undone by the fragment reads):
//   k-contiguous operand, EXT rows:   row r at r * 192: plane q at + q * 64, its k-chunk c (8 k) at slot c ^ ((r >> 2) & 3).
//       A lane's MFMA operand (8 consecutive k of one row and plane) is one ds_read_b128; the 16 lanes of a service group
//       (rows with i & 3 = 0..3 four times, (i >> 2) & 3 distinct among equals) touch 16 distinct 16-byte bank slots.
//   k-strided operand, EXT columns:   k-row r at r * EXT * 6: 64-byte QUADRANT Q = 3 * (ext / 32) + q (32 ext of plane q) at
//       quadrant Q ^ (r & 3) (EXT = 128: 768-byte rows, all starting on bank 0) or Q ^ ((r >> 1) & 1) (EXT = 64: 384-byte
//       rows alternating between bank 0 and bank 32).  ds_read_b64_tr_b16 reads the four k-rows of a 16-lane group from four
//       different quadrants of the 256-byte bank row.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define X3_HD __host__ __device__ __forceinline__
#else
#define X3_HD inline
#endif

namespace tfk {
namespace x3 {

constexpr int kBlock = 32;     // elements per interleave block
constexpr int kSlotK = 32;     // k per ring slot

// element offset of plane 0 of flat element i (planes 1, 2: + 32, + 64)
X3_HD size_t il(size_t i) { return (i >> 5) * 96 + (i & 31); }

// ---- k-contiguous image: 12 chunks of 16 bytes per row ----
X3_HD int kc_swz(int r) { return (r >> 2) & 3; }
// chunk n of the image holds k-chunk c (k = 8 c .. 8 c + 7) of plane q of row r
X3_HD void kc_decode(int n, int& r, int& q, int& c) {
  r = n / 12;
  const int rem = n - r * 12;
  q = rem >> 2;
  c = (rem & 3) ^ kc_swz(r);
}
X3_HD int kc_addr(int r, int q, int c) { return r * 192 + q * 64 + ((c ^ kc_swz(r)) << 4); }

// ---- k-strided image: EXT * 6 bytes per k-row = EXT / 32 * 3 quadrants of 4 chunks ----
template <int EXT>
X3_HD int ks_swz(int r) {
  return EXT == 64 ? ((r >> 1) & 1) : (r & 3);
}
// chunk n of the image holds ext-chunk e8 (8 ext) of block b (32 ext), plane q, of k-row r
template <int EXT>
X3_HD void ks_decode(int n, int& r, int& b, int& q, int& e8) {
  constexpr int CPR = EXT * 6 / 16;
  r = n / CPR;
  const int pos = n - r * CPR;
  const int Q = (pos >> 2) ^ ks_swz<EXT>(r);
  b = Q / 3;
  q = Q - 3 * b;
  e8 = pos & 3;
}
template <int EXT>
X3_HD int ks_addr(int r, int b, int q, int e8) {
  return r * (EXT * 6) + ((((3 * b + q) ^ ks_swz<EXT>(r)) << 2) + e8) * 16;
}

}  // namespace x3
}  // namespace tfk
