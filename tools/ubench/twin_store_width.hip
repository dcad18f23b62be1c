// GPU micro-benchmark (round 6): does the WIDTH of a lane's plane store matter for the BN / activation passes?
// hb_apply / bn_act_forward write the three-plane twin as three 8-byte stores per float4 of columns (kernels.hip: st4_twin).
// Variant B gives a thread 8 columns x 2 rows instead of 4 columns x 4 rows: the same loads in flight per thread, the same
// 512 resident blocks, but 16-byte plane stores (a wave's store instruction then covers 8 whole 128-byte lines instead of 4).
// Both variants: read two fp32 [T, H] matrices (streaming stores like the product, TFK_BN_NT=3), write 6 B per element in the
// tiled x3 layout (x3_layout.h), no statistics -- the memory skeleton of hb_apply at cfg2 (16 MB in, 12 MB out).
// Buffers rotate over NBUF copies so that nothing is served from a cache that the step would not have.
// build: hipcc --offload-arch=gfx950 -O3 -o build/twin_store_width tools/ubench/twin_store_width.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../tfkaldi_amd/csrc/x3_layout.h"
using namespace tfk;
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split(float v, uint16_t& a, uint16_t& b, uint16_t& c) { x3::split3(v, a, b, c); }

// A: block (32, 8): thread = 4 columns, rows y, y + 8, y + 16, y + 24 of a 32-row slab (the product's geometry)
__global__ void __launch_bounds__(256) var_a(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  float4 gv[4], zv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gv[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    zv[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v[4] = {gv[j].x * zv[j].x, gv[j].y * zv[j].y, gv[j].z * zv[j].z, gv[j].w * zv[j].w};
    u16x4 q0, q1, q2;
    uint16_t a, b, c;
    split(v[0], a, b, c); q0.x = a; q1.x = b; q2.x = c;
    split(v[1], a, b, c); q0.y = a; q1.y = b; q2.y = c;
    split(v[2], a, b, c); q0.z = a; q1.z = b; q2.z = c;
    split(v[3], a, b, c); q0.w = a; q1.w = b; q2.w = c;
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d));
    __builtin_nontemporal_store(q1, reinterpret_cast<u16x4*>(d + 64));
    __builtin_nontemporal_store(q2, reinterpret_cast<u16x4*>(d + 128));
  }
}
// B: block (32, 8): thread = 8 columns, rows y, y + 8 of a 16-row slab
__global__ void __launch_bounds__(256) var_b(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 8;
  const int r0 = blockIdx.y * 16 + threadIdx.y;
  float4 gv[2][2], zv[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      gv[j][k] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col + 4 * k);
      zv[j][k] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col + 4 * k);
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    u16x8 q0, q1, q2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float v[4] = {gv[j][k].x * zv[j][k].x, gv[j][k].y * zv[j][k].y, gv[j][k].z * zv[j][k].z, gv[j][k].w * zv[j][k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t a, b, c;
        split(v[i], a, b, c);
        q0[4 * k + i] = a; q1[4 * k + i] = b; q2[4 * k + i] = c;
      }
    }
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);  // (8 columns share a 32-column unit)
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x8*>(d));
    __builtin_nontemporal_store(q1, reinterpret_cast<u16x8*>(d + 64));
    __builtin_nontemporal_store(q2, reinterpret_cast<u16x8*>(d + 128));
  }
}
// C: as A, plain (cached) stores -- the pre-round-6 form, for scale
__global__ void __launch_bounds__(256) var_c(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  float4 gv[4], zv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gv[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    zv[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v[4] = {gv[j].x * zv[j].x, gv[j].y * zv[j].y, gv[j].z * zv[j].z, gv[j].w * zv[j].w};
    u16x4 q0, q1, q2;
    uint16_t a, b, c;
    split(v[0], a, b, c); q0.x = a; q1.x = b; q2.x = c;
    split(v[1], a, b, c); q0.y = a; q1.y = b; q2.y = c;
    split(v[2], a, b, c); q0.z = a; q1.z = b; q2.z = c;
    split(v[3], a, b, c); q0.w = a; q1.w = b; q2.w = c;
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);
    *reinterpret_cast<u16x4*>(d) = q0;
    *reinterpret_cast<u16x4*>(d + 64) = q1;
    *reinterpret_cast<u16x4*>(d + 128) = q2;
  }
}

// R: as A with NR rows per thread (slab of 8 * NR rows per block): 1 -> 2048 blocks ... 8 -> 256 blocks
template <int NR>
__global__ void __launch_bounds__(256) var_r(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 8 * NR + threadIdx.y;
  float4 gv[NR], zv[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    gv[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    zv[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const float v[4] = {gv[j].x * zv[j].x, gv[j].y * zv[j].y, gv[j].z * zv[j].z, gv[j].w * zv[j].w};
    u16x4 q0, q1, q2;
    uint16_t a, b, c;
    split(v[0], a, b, c); q0.x = a; q1.x = b; q2.x = c;
    split(v[1], a, b, c); q0.y = a; q1.y = b; q2.y = c;
    split(v[2], a, b, c); q0.z = a; q1.z = b; q2.z = c;
    split(v[3], a, b, c); q0.w = a; q1.w = b; q2.w = c;
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d));
    __builtin_nontemporal_store(q1, reinterpret_cast<u16x4*>(d + 64));
    __builtin_nontemporal_store(q2, reinterpret_cast<u16x4*>(d + 128));
  }
}
// P: pipelined -- 512 blocks as A, but each thread's four rows go as two batches of two: the second batch's loads are in flight
// while the first batch is stored (half the bytes in flight per thread, loads and stores overlapping inside a block)
__global__ void __launch_bounds__(256) var_p(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  float4 gv[2], zv[2], gn[2], zn[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gv[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    zv[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gn[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 16 + 8 * j) * ld + col);
    zn[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 16 + 8 * j) * ld + col);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 G = h ? gn[j] : gv[j], Z = h ? zn[j] : zv[j];
      const float v[4] = {G.x * Z.x, G.y * Z.y, G.z * Z.z, G.w * Z.w};
      u16x4 q0, q1, q2;
      uint16_t a, b, c;
      split(v[0], a, b, c); q0.x = a; q1.x = b; q2.x = c;
      split(v[1], a, b, c); q0.y = a; q1.y = b; q2.y = c;
      split(v[2], a, b, c); q0.z = a; q1.z = b; q2.z = c;
      split(v[3], a, b, c); q0.w = a; q1.w = b; q2.w = c;
      uint16_t* d = tw + x3::at((size_t)(r0 + 16 * h + 8 * j), col, ld);
      __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d));
      __builtin_nontemporal_store(q1, reinterpret_cast<u16x4*>(d + 64));
      __builtin_nontemporal_store(q2, reinterpret_cast<u16x4*>(d + 128));
    }
  }
}
// L: loads only (the same 16 MB, one store per block so that nothing is optimised away);  S: stores only (12 MB of planes)
__global__ void __launch_bounds__(256) var_l(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    const float4 b = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  if (s == 123.456f) tw[0] = 1;
}
__global__ void __launch_bounds__(256) var_s(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld) {
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u16x4 q0 = {(uint16_t)col, (uint16_t)r0, 3, 4};
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d));
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d + 64));
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d + 128));
  }
}
__global__ void var_empty() {}

// H<0>: A + hb_apply's statistics prologue as the product has it (the dA epilogue's 8 partial sums per column for two quantities,
// one per row lane, merged by two reduce_rows of two barriers each + a broadcast through LDS: five barriers before the first
// store can go out) + its closing column sum (one more reduce_rows);  H<1>: the same sums, in the same order, behind ONE write +
// barrier + row lane 0 summing both quantities + one broadcast barrier (two barriers)
template <int MODE>
__global__ void __launch_bounds__(256) var_h(const float* __restrict__ g, const float* __restrict__ z, uint16_t* __restrict__ tw, int ld,
                                             const float* __restrict__ ws, float* __restrict__ ws_out) {
  __shared__ float4 sm[8][32];
  __shared__ float4 sm2[8][32];
  __shared__ float4 smm[2][32];
  const int col = (blockIdx.x * 32 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * 32 + threadIdx.y;
  float4 gv[4], zv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gv[j] = *reinterpret_cast<const float4*>(g + (size_t)(r0 + 8 * j) * ld + col);
    zv[j] = *reinterpret_cast<const float4*>(z + (size_t)(r0 + 8 * j) * ld + col);
  }
  const float4 p1 = *reinterpret_cast<const float4*>(ws + ((size_t)0 * 64 + threadIdx.y) * ld + col);
  const float4 p2 = *reinterpret_cast<const float4*>(ws + ((size_t)1 * 64 + threadIdx.y) * ld + col);
  float4 m1, m2;
  auto rr = [&](float4 v) {
    sm[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.y == 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float4 q = sm[k][threadIdx.x]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
    __syncthreads();
    return t;
  };
  if (MODE == 0) {
    const float4 a = rr(p1), b = rr(p2);
    if (threadIdx.y == 0) { smm[0][threadIdx.x] = a; smm[1][threadIdx.x] = b; }
    __syncthreads();
  } else {
    sm[threadIdx.y][threadIdx.x] = p1;
    sm2[threadIdx.y][threadIdx.x] = p2;
    __syncthreads();
    if (threadIdx.y < 2) {  // row lane 0 sums the first quantity, row lane 1 the second (two waves... one wave: y = 0, 1 share a wave)
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float4 q = threadIdx.y ? sm2[k][threadIdx.x] : sm[k][threadIdx.x]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
      smm[threadIdx.y][threadIdx.x] = t;
    }
    __syncthreads();
  }
  m1 = smm[0][threadIdx.x];
  m2 = smm[1][threadIdx.x];
  float4 sz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v[4] = {gv[j].x - m1.x - zv[j].x * m2.x, gv[j].y - m1.y - zv[j].y * m2.y, gv[j].z - m1.z - zv[j].z * m2.z,
                        gv[j].w - m1.w - zv[j].w * m2.w};
    sz.x += v[0]; sz.y += v[1]; sz.z += v[2]; sz.w += v[3];
    u16x4 q0, q1, q2;
    uint16_t a, b, c;
    split(v[0], a, b, c); q0.x = a; q1.x = b; q2.x = c;
    split(v[1], a, b, c); q0.y = a; q1.y = b; q2.y = c;
    split(v[2], a, b, c); q0.z = a; q1.z = b; q2.z = c;
    split(v[3], a, b, c); q0.w = a; q1.w = b; q2.w = c;
    uint16_t* d = tw + x3::at((size_t)(r0 + 8 * j), col, ld);
    __builtin_nontemporal_store(q0, reinterpret_cast<u16x4*>(d));
    __builtin_nontemporal_store(q1, reinterpret_cast<u16x4*>(d + 64));
    __builtin_nontemporal_store(q2, reinterpret_cast<u16x4*>(d + 128));
  }
  sz = rr(sz);
  if (threadIdx.y == 0) *reinterpret_cast<float4*>(ws_out + (size_t)blockIdx.y * ld + col) = sz;
}


int main() {
  const int T = 1024, H = 2048, NBUF = 24, ITERS = 240;
  const size_t n = (size_t)T * H;
  float *g, *z;
  uint16_t* tw;
  CK(hipMalloc(&g, NBUF * n * 4)); CK(hipMalloc(&z, NBUF * n * 4)); CK(hipMalloc(&tw, NBUF * n * 6));
  CK(hipMemset(g, 0x3c, NBUF * n * 4)); CK(hipMemset(z, 0x3d, NBUF * n * 4));
  float *wsp, *wso;
  CK(hipMalloc(&wsp, 2 * 64 * H * 4)); CK(hipMalloc(&wso, 64 * H * 4)); CK(hipMemset(wsp, 0, 2 * 64 * H * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"A: 4 columns x 4 rows, 8-byte streamed plane stores (the product's geometry)",
                         "B: 8 columns x 2 rows, 16-byte streamed plane stores",
                         "C: as A, cached stores",
                         "R1: 1 row per thread, 2048 blocks", "R2: 2 rows per thread, 1024 blocks", "R8: 8 rows per thread, 256 blocks",
                         "P: as A, two batches of two rows (second batch's loads over the first batch's stores)",
                         "L: the loads of A alone (16 MB)", "S: the stores of A alone (12 MB)", "E: an empty kernel",
                         "H0: A + hb_apply's statistics prologue (five barriers) and closing column sum", "H1: the same sums behind two barriers"};
  for (int rep = 0; rep < 3; ++rep)
    for (int var = 0; var < 12; ++var) {
      for (int it = -10; it < ITERS; ++it) {
        if (it == 0) CK(hipEventRecord(e0, 0));
        const int b = (it + 10) % NBUF;
        const float* gp = g + b * n;
        const float* zp = z + b * n;
        uint16_t* tp = tw + b * n * 3;
        switch (var) {
          case 0: hipLaunchKernelGGL(var_a, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 1: hipLaunchKernelGGL(var_b, dim3(H / 256, T / 16), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 2: hipLaunchKernelGGL(var_c, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 3: hipLaunchKernelGGL(var_r<1>, dim3(H / 128, T / 8), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 4: hipLaunchKernelGGL(var_r<2>, dim3(H / 128, T / 16), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 5: hipLaunchKernelGGL(var_r<8>, dim3(H / 128, T / 64), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 6: hipLaunchKernelGGL(var_p, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 7: hipLaunchKernelGGL(var_l, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 8: hipLaunchKernelGGL(var_s, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H); break;
          case 9: hipLaunchKernelGGL(var_empty, dim3(1), dim3(64), 0, 0); break;
          case 10: hipLaunchKernelGGL(var_h<0>, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H, wsp, wso); break;
          default: hipLaunchKernelGGL(var_h<1>, dim3(H / 128, T / 32), dim3(32, 8), 0, 0, gp, zp, tp, H, wsp, wso); break;
        }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rep %d  %-95s %6.2f us per launch back to back\n", rep, names[var], 1e3 * ms / ITERS);
    }
  return 0;
}
