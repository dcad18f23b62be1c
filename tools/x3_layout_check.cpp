// CPU tool / test helper: the index arithmetic of the fp32-emulating contraction's operand layouts (tfkaldi_amd/csrc/x3_layout.h)
// checked without a GPU.  For every operand kind the kernel instantiates it emulates (1) the LDS-DMA fill of one ring slot from a
// tiled three-plane array -- thread t, piece j writes the 16-byte chunk t + j * NTH of the image from the source address
// DmaOperand3::init computes -- and (2) the fragment reads of Frag3, and asserts that every lane receives exactly the elements the
// MFMA operand needs; then it simulates the LDS bank schedule of those reads (ds_read_b128: four 16-lane service groups;
// ds_read_b64_tr_b16: two 32-lane groups) and asserts that no group touches a bank twice.
//   g++ -O1 -std=c++17 [-DTFK_X3_M16=1] -I tfkaldi_amd/csrc tools/x3_layout_check.cpp -o /tmp/x3_layout_check && /tmp/x3_layout_check
// (-DTFK_X3_M16=1: the 16x16x32 variant of x3_layout.h -- its swizzles and its fragment reads)
#include <stdio.h>
#include <stdlib.h>

#include <set>
#include <vector>

#include "x3_layout.h"

using namespace tfk;

static int g_fail = 0;
#define EXPECT(c, ...)                \
  do {                                \
    if (!(c)) {                       \
      if (g_fail < 20) { printf("FAIL: " __VA_ARGS__); printf("\n"); } \
      ++g_fail;                       \
    }                                 \
  } while (0)

// value stored for (flat index, plane): unique per element
static uint32_t tag(size_t flat, int q) { return (uint32_t)(flat * 3 + q + 1); }

// the tiled three-plane array of an [rows, ld] matrix
static std::vector<uint32_t> make_array(int rows, int ld) {
  std::vector<uint32_t> a(x3::elems(rows, ld), 0);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < ld; ++c)
      for (int q = 0; q < 3; ++q) a[x3::at(r, c, ld) + q * 64] = tag((size_t)r * ld + c, q);
  return a;
}

template <bool KC, int EXT, int NTH>
static void check(int rows, int ld, int ext0, int k0, const char* name) {
  constexpr int NP = EXT * 12 / NTH;
  const std::vector<uint32_t> arr = make_array(rows, ld);
  std::vector<uint32_t> img((size_t)EXT * 96, 0xffffffffu);  // elements of the slot image (EXT * 192 bytes)
  // ---- DMA (DmaOperand3::init + issue) ----
  for (int tid = 0; tid < NTH; ++tid)
    for (int j = 0; j < NP; ++j) {
      const int n = tid + j * NTH;
      size_t src;  // element offset of the chunk's source
      if (KC) {
        int r, q, c;
        x3::kc_decode(n, r, q, c);
        src = x3::at(ext0 + r, c * 8, ld) + q * 64 + (size_t)k0 * 6;  // soff = k0 * 12 bytes
      } else {
        int r, b, q, e8;
        x3::ks_decode<EXT>(n, r, b, q, e8);
        src = x3::at(r, ext0 + b * 32 + e8 * 8, ld) + q * 64 + (size_t)k0 * ld * 3;  // soff = k0 * ld * 6 bytes
      }
      EXPECT(src + 8 <= arr.size(), "%s: source chunk out of range", name);
      for (int e = 0; e < 8; ++e) img[(size_t)n * 8 + e] = src + e < arr.size() ? arr[src + e] : 0;
    }
#if TFK_X3_M16
  // ---- fragment reads of the 16x16x32 shape (Frag3M): fragments of 16 rows (columns), all 32 k of the slot ----
  for (int frag0 = 0; frag0 < EXT / 16; frag0 += 2)
    for (int pl = 0; pl < 3; ++pl)
      for (int f = 0; frag0 + f < EXT / 16; ++f) {
        std::vector<int> addr(64), addr_hi(64);
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4;
          if (KC) {
            const int a = x3::kc16_lane_off(lane, frag0) + x3::kc16_imm(f, pl);
            addr[lane] = a;
            const int row = ext0 + (frag0 + f) * 16 + (lane & 15);
            for (int e = 0; e < 8; ++e)
              EXPECT(img[a / 2 + e] == tag((size_t)row * ld + k0 + 8 * g + e, pl), "%s: KC16 fragment f%d+%d pl%d lane %d", name, frag0, f, pl,
                     lane);
          } else {
            const int j = (lane >> 2) & 3, q = lane & 3;
            const int X = 3 * (f >> 1) + pl;
            for (int hi = 0; hi < 2; ++hi) {
              const int off = x3::ks16_lane_off<EXT>(lane, frag0, X & 1, f & 1);
              EXPECT(off >= 0, "%s: negative lane offset", name);
              const int a = off + x3::ks16_imm<EXT>(X, hi);
              (hi ? addr_hi : addr)[lane] = a;
              EXPECT(x3::ks16_imm<EXT>(X, hi) < 65536, "%s: immediate out of range", name);
              const int krow = k0 + 8 * g + j + 4 * hi;
              const int ext = ext0 + (frag0 + f) * 16 + 4 * q;
              for (int e = 0; e < 4; ++e)
                EXPECT(img[a / 2 + e] == tag((size_t)krow * ld + ext + e, pl), "%s: KS16 fragment f%d+%d pl%d lane %d hi%d", name, frag0, f, pl,
                       lane, hi);
            }
          }
        }
        if (KC) {
          static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                            {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                            {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                            {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
          for (auto& grp : groups) {
            std::set<int> slots;
            for (int l : grp) slots.insert((addr[l] / 16) % 16);
            EXPECT(slots.size() == 16, "%s: ds_read_b128 bank conflict, 16x16x32 (f%d pl%d): %zu slots", name, f, pl, slots.size());
          }
        } else {
          for (int hi = 0; hi < 2; ++hi)
            for (int grp = 0; grp < 2; ++grp) {
              std::set<int> banks;
              for (int l = 32 * grp; l < 32 * grp + 32; ++l) {
                const int a = hi ? addr_hi[l] : addr[l];
                banks.insert((a / 4) % 64);
                banks.insert((a / 4 + 1) % 64);
              }
              EXPECT(banks.size() == 64, "%s: ds_read_b64_tr_b16 bank conflict, 16x16x32 (f%d pl%d): %zu banks", name, f, pl, banks.size());
            }
        }
      }
#endif
  // ---- fragment reads (Frag3::init + read), for a wave whose first fragment is frag0 ----
  const int NF = EXT / 32;
  for (int frag0 = 0; frag0 < (TFK_X3_M16 ? 0 : NF); ++frag0)
  for (int pl = 0; pl < 3; ++pl)
    for (int f = 0; frag0 + f < NF; ++f)
      for (int ks = 0; ks < 2; ++ks) {
        std::vector<int> addr(64), addr_hi(64);
        for (int lane = 0; lane < 64; ++lane) {
          if (KC) {
            const int i = lane & 31, kb = lane >> 5;
            const int a = x3::kc_lane_off(lane, frag0, ks) + x3::kc_imm(f, pl);
            addr[lane] = a;
            const int row = ext0 + (frag0 + f) * 32 + i;
            for (int e = 0; e < 8; ++e)
              EXPECT(img[a / 2 + e] == tag((size_t)row * ld + k0 + 16 * ks + 8 * kb + e, pl), "%s: KC fragment f%d+%d ks%d pl%d lane %d",
                     name, frag0, f, ks, pl, lane);
          } else {
            const int kb = lane >> 5, half = (lane >> 4) & 1, j = (lane >> 2) & 3, q = lane & 3;
            const int X = 3 * f + pl;
            EXPECT(x3::ks_lane_off<EXT>(lane, frag0, X & 1) >= 0, "%s: negative lane offset", name);
            for (int hi = 0; hi < 2; ++hi) {
              const int a = x3::ks_lane_off<EXT>(lane, frag0, X & 1) + x3::ks_imm<EXT>(X, ks, hi);
              (hi ? addr_hi : addr)[lane] = a;
              EXPECT(x3::ks_imm<EXT>(X, ks, hi) < 65536, "%s: immediate out of range", name);
              const int krow = k0 + 16 * ks + 8 * kb + j + 4 * hi;
              const int ext = ext0 + (frag0 + f) * 32 + 16 * half + 4 * q;
              for (int e = 0; e < 4; ++e)
                EXPECT(img[a / 2 + e] == tag((size_t)krow * ld + ext + e, pl), "%s: KS fragment f%d+%d ks%d pl%d lane %d hi%d",
                       name, frag0, f, ks, pl, lane, hi);
            }
          }
        }
        // ---- bank schedule ----
        if (KC) {
          static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                            {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                            {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                            {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
          for (auto& g : groups) {
            std::set<int> slots;
            for (int l : g) slots.insert((addr[l] / 16) % 16);
            EXPECT(slots.size() == 16, "%s: ds_read_b128 bank conflict (f%d ks%d pl%d): %zu slots", name, f, ks, pl, slots.size());
          }
        } else {
          for (int hi = 0; hi < 2; ++hi)
            for (int g = 0; g < 2; ++g) {
              std::set<int> banks;
              for (int l = 32 * g; l < 32 * g + 32; ++l) {
                const int a = hi ? addr_hi[l] : addr[l];
                banks.insert((a / 4) % 64);
                banks.insert((a / 4 + 1) % 64);
              }
              EXPECT(banks.size() == 64, "%s: ds_read_b64_tr_b16 bank conflict (f%d ks%d pl%d): %zu banks", name, f, ks, pl, banks.size());
            }
        }
      }
  // ---- source contiguity: how many 128-byte lines one wave instruction touches ----
  double lines = 0;
  int instr = 0;
  for (int w = 0; w < NTH / 64; ++w)
    for (int j = 0; j < NP; ++j) {
      std::set<size_t> ln;
      for (int lane = 0; lane < 64; ++lane) {
        const int n = w * 64 + lane + j * NTH;
        size_t src;
        if (KC) {
          int r, q, c;
          x3::kc_decode(n, r, q, c);
          src = x3::at(ext0 + r, c * 8, ld) + q * 64 + (size_t)k0 * 6;
        } else {
          int r, b, q, e8;
          x3::ks_decode<EXT>(n, r, b, q, e8);
          src = x3::at(r, ext0 + b * 32 + e8 * 8, ld) + q * 64 + (size_t)k0 * ld * 3;
        }
        ln.insert(src * 2 / 128);
      }
      lines += ln.size();
      ++instr;
    }
  printf("%-34s ld %5d k0 %4d: ok so far, %.1f 128-byte lines per 1-KiB wave instruction (8 = every byte of a line used)\n", name, ld, k0,
         lines / instr);
}

int main() {
  for (int ld : {2048, 448, 2016}) {
    for (int k0 : {0, 32, 96}) {
      check<true, 128, 256>(256, ld, 128, k0, "k-contiguous, 128 rows, 4 waves");
      check<true, 128, 512>(256, ld, 128, k0, "k-contiguous, 128 rows, 8 waves");
      check<true, 64, 256>(256, ld, 64, k0, "k-contiguous,  64 rows, 4 waves");
    }
  }
  for (int ld : {2048, 2016, 448}) {
    for (int k0 : {0, 32, 64}) {
      check<false, 128, 256>(128, ld, 128, k0, "k-strided, 128 columns, 4 waves");
      check<false, 128, 512>(128, ld, 128, k0, "k-strided, 128 columns, 8 waves");
      check<false, 64, 256>(128, ld, 192, k0, "k-strided,  64 columns, 4 waves");
    }
  }
  if (g_fail) {
    printf("%d FAILURES\n", g_fail);
    return 1;
  }
  printf("x3 layout: all checks passed\n");
  return 0;
}
