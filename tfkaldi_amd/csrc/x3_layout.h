// Memory and LDS layouts of the fp32-emulating contraction ("bf16x3", gemm_bf16.h) -- index arithmetic only, shared by the
// kernels (gemm_bf16.hip, kernels.hip) and by a host-side checker (tools/x3_layout_check.cpp, runs without a GPU).
//
// MEMORY.  An fp32 matrix [rows, ld] (ld a multiple of 32, rows rounded up to even) is held as three bf16 planes
// p0 + p1 + p2 = x exactly, TILED: the unit is the 2-row x 32-column block (R, B) = (row / 2, col / 32), 384 bytes =
// three 128-byte lines, one per plane, each line holding the block's two rows of that plane:
//       element (row, col), plane q  at  ((row / 2) * (ld / 32) + col / 32) * 192 + q * 64 + (row & 1) * 32 + col % 32
// (bf16 elements from the matrix's base).  Why: the L1-miss path from L2 into a CU moves whole 128-byte lines at ~36 B/clk/CU,
// and a ring slot of the contraction holds 32 k of both operands.  With one plane per array (round 4) a k-contiguous operand
// fetched a 64-byte piece per row, plane and slot -- half of every line it pulled, 21 B/clk/CU of useful bytes
// (profiles/r04_gemm_f32x3_ablation.txt); planes interleaved per row gave 192-byte runs, 1.5 lines each of which the second was
// shared with the NEXT slot: 27 B/clk/CU.  Here whatever a block stages in one slot consists of COMPLETE lines in both uses of a
// matrix: as the k-contiguous operand (a tile of 128 rows x 32 k = 64 units, each wholly inside it) and as the k-strided operand
// (32 k-rows x 128 columns = 16 row pairs x 4 units).
//
// LDS (one ring slot = 32 k of all three planes of both operands; every image is written lane-linearly by
// `buffer_load_dwordx4 ... lds`, 16-byte chunk n of an image at byte 16 n, so the permutations below are applied on the SOURCE
// side of the DMA and undone by the fragment reads; consecutive chunks of an image come from consecutive memory):
//   k-contiguous operand, EXT rows:   row pair P at P * 384: [plane q][row parity h][4 chunks of 8 k], the k-chunk c of row r at
//       slot c ^ ((r >> 2) & 3).  A lane's MFMA operand (8 consecutive k of one row and plane) is one ds_read_b128; the 16 lanes
//       of a service group touch 16 distinct 16-byte bank slots.
//   k-strided operand, EXT columns:   k-row pair P at P * (EXT / 32) * 384: per 32-column block b the unit's six 64-byte QUADRANTS
//       n = 6 b + 2 q + h (plane q, row parity h), quadrant n stored at n ^ (2 * (P & 1)): the four k-rows a 16-lane group of
//       ds_read_b64_tr_b16 reads (two pairs) then sit in four different quadrants of the 256-byte bank row.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define X3_HD __host__ __device__ __forceinline__
#else
#define X3_HD inline
#endif

namespace tfk {
namespace x3 {

constexpr int kSlotK = 32;  // k per ring slot = columns per unit

// TFK_X3_M16 = 1 (compile-time, off): the contraction issues v_mfma_f32_16x16x32_bf16 (one 32-k step per ring slot) instead of
// 32x32x16 (two 16-k steps).  Built because a loop of nothing but MFMAs runs 12 % faster in that shape under the socket's power
// limit (profiles/r05_mfma_bf16_energy.txt); in the kernel it buys nothing -- the pair of a layer 83.1 against 83.7 us sustained,
// a stacked pass 307.4 against 311.2, the training step 1.113 against 1.113 ms (three runs each on one box) -- so the default
// stays the shape the epilogues are written for.  The variant passes tests/test_gpu_f32x3.py and tools/x3_layout_check.cpp.
// The images keep their structure; what changes is which lane reads what: lane l of a fragment of 16 rows (columns) holds the
// 8 k of k-chunk l >> 4 of row (column) l & 15 -- and therefore the two swizzles that keep those reads free of bank conflicts.
#ifndef TFK_X3_M16
#define TFK_X3_M16 0
#endif

// elements a matrix of `rows` rows occupies (all three planes)
X3_HD size_t elems(size_t rows, int ld) { return ((rows + 1) >> 1) * (size_t)(ld >> 5) * 192; }
// element offset of plane 0 of element (row, col); planes 1, 2: + 64, + 128
X3_HD size_t at(size_t row, int col, int ld) {
  return ((row >> 1) * (size_t)(ld >> 5) + (size_t)(col >> 5)) * 192 + (row & 1) * 32 + (col & 31);
}

// ---- k-contiguous image: 24 chunks of 16 bytes per row pair ----
#if TFK_X3_M16
// rows r, r + 4, r + 8, r + 12 of a 16-row fragment share the upper bank bits (from r & 3); a ds_read_b128 service group holds
// them with k-chunks (c, c ^ 1, c ^ 1, c): {0, 0, 3, 3} separates the four in every group
X3_HD int kc_swz(int r) { return ((r >> 3) & 1) * 3; }
#else
X3_HD int kc_swz(int r) { return (r >> 2) & 3; }
#endif
// chunk n of the image holds k-chunk c (k = 8 c .. 8 c + 7) of plane q of row r
X3_HD void kc_decode(int n, int& r, int& q, int& c) {
  const int P = n / 24, rem = n - P * 24;
  q = rem >> 3;
  r = 2 * P + ((rem >> 2) & 1);
  c = (rem & 3) ^ kc_swz(r);
}
X3_HD int kc_addr(int r, int q, int c) { return (r >> 1) * 384 + q * 128 + (r & 1) * 64 + ((c ^ kc_swz(r)) << 4); }

// ---- k-strided image: EXT / 32 units of 6 quadrants (64 bytes) per k-row pair ----
// chunk n of the image holds ext-chunk e8 (8 ext) of block b (32 ext), plane q, of k-row r
template <int EXT>
X3_HD void ks_decode(int n, int& r, int& b, int& q, int& e8) {
  constexpr int CPP = EXT / 32 * 24;  // chunks per row pair
  const int P = n / CPP, pos = n - P * CPP;
  const int nq = (pos >> 2) ^ ((P & 1) << 1);
  b = nq / 6;
  const int rem = nq - 6 * b;
  q = rem >> 1;
  r = 2 * P + (rem & 1);
#if TFK_X3_M16
  e8 = (pos & 3) ^ (((P >> 2) & 1) << 1);  // the two 16-column halves of a quadrant swap in every other group of four row pairs
#else
  e8 = pos & 3;
#endif
}
template <int EXT>
X3_HD int ks_addr(int r, int b, int q, int e8) {
  const int P = r >> 1;
#if TFK_X3_M16
  e8 ^= ((P >> 2) & 1) << 1;
#endif
  return P * (EXT / 32 * 384) + ((((6 * b + 2 * q + (r & 1)) ^ ((P & 1) << 1)) << 2) + e8) * 16;
}

// ---- fragment reads: two per-lane byte offsets per operand, everything else an immediate ----
// k-contiguous (ds_read_b128): lane (i = lane & 31, kb = lane >> 5) reads row 32 * frag + i, k-chunk 2 ks + kb of plane q at
//   kc_lane_off(lane, frag0, ks) + kc_imm(f, q),  frag = frag0 + f
X3_HD int kc_lane_off(int lane, int frag0, int ks) { return kc_addr(frag0 * 32 + (lane & 31), 0, 2 * ks + (lane >> 5)); }
X3_HD int kc_imm(int f, int q) { return f * (16 * 384) + q * 128; }
// k-strided (two ds_read_b64_tr_b16, the second `hi` four k-rows further): lane (kb, half, j, qq) = (lane >> 5, (lane >> 4) & 1,
//   (lane >> 2) & 3, lane & 3) reads 8 bytes of k-row 16 ks + 8 kb + j (+ 4), columns 32 * frag + 16 half + 4 qq .. + 3, at
//   ks_lane_off(lane, frag0, X & 1) + ks_imm(X, ks, hi),  X = 3 f + q.  The 128-byte line of (frag, q) inside a row pair is
//   Xt = 3 frag + q, stored at Xt ^ t (t = parity of the lane's row pair): Xt + t for even Xt, Xt - t for odd.
template <int EXT>
X3_HD int ks_lane_off(int lane, int frag0, int x_odd) {
  const int kb = lane >> 5, half = (lane >> 4) & 1, j = (lane >> 2) & 3, qq = lane & 3;
  const int t = (j >> 1) & 1;
  const int base = (4 * kb + (j >> 1)) * (EXT / 32 * 384) + (j & 1) * 64 + (2 * half + (qq >> 1)) * 16 + ((qq & 1) << 3) +
                   128 * 3 * frag0;
  const int xt_odd = (frag0 + x_odd) & 1;  // parity of Xt = 3 frag0 + X
  // the immediate is 128 * X for even X and 128 * (X - 1) for odd X: an odd X carries its own + 128 here
  return base + (xt_odd ? -128 * t : 128 * t) + (x_odd ? 128 : 0);
}
template <int EXT>
X3_HD int ks_imm(int X, int ks, int hi) {
  return 128 * (X & ~1) + (ks * 8 + hi * 2) * (EXT / 32 * 384);
}

// ---- fragment reads for the 16x16x32 shape (TFK_X3_M16): fragments of 16 rows (columns), one 32-k step per slot ----
// k-contiguous (ds_read_b128): lane (i = lane & 15, g = lane >> 4) reads row 16 * frag + i, k-chunk g of plane q at
//   kc16_lane_off(lane, frag0) + kc16_imm(f, q),  frag = frag0 + f  (frag0 even: a wave's tile starts on a 32-row boundary)
X3_HD int kc16_lane_off(int lane, int frag0) { return kc_addr(frag0 * 16 + (lane & 15), 0, lane >> 4); }
X3_HD int kc16_imm(int f, int q) { return f * (8 * 384) + q * 128; }
// k-strided (two ds_read_b64_tr_b16, `hi` four k-rows further): lane (g, j, qq) = (lane >> 4, (lane >> 2) & 3, lane & 3) reads 8 bytes
//   of k-row 8 g + j (+ 4), columns 16 * frag + 4 qq .. + 3, at  ks16_lane_off(lane, frag0, X & 1, f & 1) + ks16_imm(X, hi),
//   X = 3 (f >> 1) + q (the 128-byte line of the fragment's 32-column unit and plane, as in ks_lane_off).  The fragment's half of
//   the quadrant is (f & 1) ^ (g & 1): the halves swap in every other group of four row pairs (ks_decode), so that the two lane
//   groups a 32-lane service group is made of -- same columns, k-rows 8 apart -- sit in different banks.
template <int EXT>
X3_HD int ks16_lane_off(int lane, int frag0, int x_odd, int f_odd) {
  const int g = lane >> 4, j = (lane >> 2) & 3, qq = lane & 3;
  const int t = (j >> 1) & 1;
  const int unit0 = frag0 >> 1;  // (frag0 even)
  const int base = (4 * g + (j >> 1)) * (EXT / 32 * 384) + (j & 1) * 64 + ((f_odd ^ (g & 1)) << 5) + (qq >> 1) * 16 + ((qq & 1) << 3) +
                   128 * 3 * unit0;
  const int xt_odd = (unit0 + x_odd) & 1;
  return base + (xt_odd ? -128 * t : 128 * t) + (x_odd ? 128 : 0);
}
template <int EXT>
X3_HD int ks16_imm(int X, int hi) {
  return 128 * (X & ~1) + hi * 2 * (EXT / 32 * 384);
}

#if defined(__HIPCC__)
// x = p1 + p2 + p3 exactly, each a bf16, by ROUND TO NEAREST (even): p1 = RN(x), p2 = RN(x - p1), p3 = x - p1 - p2.
// Both subtractions are exact in fp32 (x - p1 keeps the low 16 bits of x's significand, sign included; the remainder after
// two planes has at most 8 significant bits and IS a bf16).  |p2| <= 2^-8 |x| and |p3| <= 2^-16 |x| -- half of what truncation
// leaves (rounds 4 / 5) -- so the three plane products the contraction drops (a2 b3, a3 b2, a3 b3) are bounded by 2^-23 |a b|
// instead of 2^-21.  The planes of one value may differ in sign.  Edges: a finite |x| above bf16's largest value (2^127 * 1.9922)
// would round to Inf: that value is split by truncation instead (the split stays exact, the bound for such a value is truncation's);
// Inf splits into (Inf, NaN, NaN) and NaN into NaNs (Inf - Inf): a non-finite operand surfaces as NaN in every result it touches.
__device__ __forceinline__ void split3(float x, uint16_t& p1, uint16_t& p2, uint16_t& p3) {
  const uint32_t ux = __builtin_bit_cast(uint32_t, x);
  uint32_t b1 = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)x) << 16;
  const bool edge = (b1 & 0x7fffffffu) == 0x7f800000u && (ux & 0x7fffffffu) < 0x7f800000u;
  if (edge) b1 = ux & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, b1);
  uint32_t b2 = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)r1) << 16;
  if (edge) b2 = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;  // (planes of one sign: no partial sum of them exceeds |x|)
  const float r2 = r1 - __builtin_bit_cast(float, b2);
  p1 = (uint16_t)(b1 >> 16);
  p2 = (uint16_t)(b2 >> 16);
  p3 = (uint16_t)(__builtin_bit_cast(uint32_t, r2) >> 16);
}
#endif

}  // namespace x3
}  // namespace tfk
