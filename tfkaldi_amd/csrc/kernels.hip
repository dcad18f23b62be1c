// Non-GEMM kernels of the DNN hot path for gfx950 (wave64).  All are L2/HBM-streaming kernels:
// 16-byte accesses, 512-byte contiguous segments per half-wave, column-tiled two-level reductions
// (deterministic: no floating-point atomics anywhere).
//
// Reference semantics restated here (TensorFlow ops the reference calls):
//   batch norm   neuralNetworks/classifiers/activation.py:159-161  tf.contrib.layers.batch_norm
//                (center=True, scale=False, biased batch variance, EMA of mean/variance)
//   nonlin       activation.py:84 (tf.nn.relu / sigmoid / tanh / identity; nnet.py:48-62)
//   L2Norm       activation.py:101-111  s = mean(x^2, axis=1); x/s where s > 1
//   Dropout      activation.py:140-141  tf.nn.dropout(x, keep_prob) in training mode only
//   softmax-CE   neuralNetworks/trainer.py:526-531  reduce_sum(softmax_cross_entropy_with_logits)
//   softmax      neuralNetworks/decoder.py:44
//   mean/clip/Adam  trainer.py:174-184
#include "kernels.h"

#include <math.h>
#include <stdlib.h>

namespace tfk {
namespace {

constexpr int NONLIN_RELU = 0, NONLIN_SIGMOID = 1, NONLIN_TANH = 2, NONLIN_LINEAR = 3;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Block-wide reductions for 1-D blocks; `sm` holds >= blockDim.x/64 floats.
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < nw; ++k) t += sm[k];
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float t = sm[0];
  for (int k = 1; k < nw; ++k) t = fmaxf(t, sm[k]);
  __syncthreads();
  return t;
}

// Philox4x32-7 counter-based RNG: the keep mask of element (row, col) is a pure function of
// (seed, call, layer, row, col/4), so forward and backward regenerate it instead of storing it.
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
struct Keep4 { bool k[4]; };
__device__ __forceinline__ Keep4 keep_mask(const ActDesc& d, int row, int c4) {
  const uint4 r = philox4x32(make_uint4((uint32_t)c4, (uint32_t)row, d.layer, d.call),
                             make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32)));
  // keep with probability `keep`: drop iff u < (1 - keep) * 2^32
  const uint32_t thr = (uint32_t)fminf((1.0f - d.keep) * 4294967296.0f, 4294967295.0f);
  Keep4 m;
  m.k[0] = r.x >= thr; m.k[1] = r.y >= thr; m.k[2] = r.z >= thr; m.k[3] = r.w >= thr;
  return m;
}

__device__ __forceinline__ float nonlin_fwd(float u, int nonlin) {
  switch (nonlin) {
    case NONLIN_RELU: return fmaxf(u, 0.f);
    case NONLIN_SIGMOID: return 1.f / (1.f + expf(-u));
    case NONLIN_TANH: return tanhf(u);
    default: return u;
  }
}
// derivative expressed through the OUTPUT v = f(u)
__device__ __forceinline__ float nonlin_bwd(float v, int nonlin) {
  switch (nonlin) {
    case NONLIN_RELU: return v > 0.f ? 1.f : 0.f;
    case NONLIN_SIGMOID: return v * (1.f - v);
    case NONLIN_TANH: return 1.f - v * v;
    default: return 1.f;
  }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// bf16 twin of a GEMM operand (mixed-precision mode): the same 4 values, round-to-nearest-even, 8 bytes
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint16_t to_bf16(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ void st4_twin(const Twin& t, size_t row, int col, float4 v) {
  if (!t.p) return;
  if (t.x3) {  // three interleaved planes whose sum is the value exactly
    uint16_t a[4], b[4], c[4];
    twin_split3(v.x, a[0], b[0], c[0]); twin_split3(v.y, a[1], b[1], c[1]);
    twin_split3(v.z, a[2], b[2], c[2]); twin_split3(v.w, a[3], b[3], c[3]);
    u16x4 q1, q2, q3;
    q1.x = a[0]; q1.y = a[1]; q1.z = a[2]; q1.w = a[3];
    q2.x = b[0]; q2.y = b[1]; q2.z = b[2]; q2.w = b[3];
    q3.x = c[0]; q3.y = c[1]; q3.z = c[2]; q3.w = c[3];
    uint16_t* d = t.p + x3::at(row, col, t.ld);  // (col is a multiple of 4: the four values share a unit)
    *reinterpret_cast<u16x4*>(d) = q1;
    *reinterpret_cast<u16x4*>(d + 64) = q2;
    *reinterpret_cast<u16x4*>(d + 128) = q3;
    return;
  }
  u16x4 q;
  q.x = to_bf16(v.x); q.y = to_bf16(v.y); q.z = to_bf16(v.z); q.w = to_bf16(v.w);
  *reinterpret_cast<u16x4*>(t.p + row * t.ld + col) = q;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming variants for the column-tiled BN / activation passes (env TFK_BN_NT, bit 0: the fp32 output, bit 1: the operand twins):
// written once, read by a LATER kernel from another XCD's side of the fabric -- nothing this kernel's L2 needs to keep
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  v4f_nt q;
  q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
  __builtin_nontemporal_store(q, reinterpret_cast<v4f_nt*>(p));
}
__device__ __forceinline__ void st4_twin_nt(const Twin& t, size_t row, int col, float4 v) {
  if (!t.p) return;
  if (t.x3) {
    uint16_t a[4], b[4], c[4];
    twin_split3(v.x, a[0], b[0], c[0]); twin_split3(v.y, a[1], b[1], c[1]);
    twin_split3(v.z, a[2], b[2], c[2]); twin_split3(v.w, a[3], b[3], c[3]);
    u16x4 q1, q2, q3;
    q1.x = a[0]; q1.y = a[1]; q1.z = a[2]; q1.w = a[3];
    q2.x = b[0]; q2.y = b[1]; q2.z = b[2]; q2.w = b[3];
    q3.x = c[0]; q3.y = c[1]; q3.z = c[2]; q3.w = c[3];
    uint16_t* d = t.p + x3::at(row, col, t.ld);
    __builtin_nontemporal_store(q1, reinterpret_cast<u16x4*>(d));
    __builtin_nontemporal_store(q2, reinterpret_cast<u16x4*>(d + 64));
    __builtin_nontemporal_store(q3, reinterpret_cast<u16x4*>(d + 128));
    return;
  }
  u16x4 q;
  q.x = to_bf16(v.x); q.y = to_bf16(v.y); q.z = to_bf16(v.z); q.w = to_bf16(v.w);
  __builtin_nontemporal_store(q, reinterpret_cast<u16x4*>(t.p + row * t.ld + col));
}
__device__ __forceinline__ float4 ld4s(const float* p, int nt) {  // (bit 2 of TFK_BN_NT: streaming loads of what is read once)
  if (nt & 4) {
    const v4f_nt q = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
  }
  return *reinterpret_cast<const float4*>(p);
}
int bn_nt_mode(const Twin& tw, int rows, int ld) {
  // default (round 6, tools/bn_nt_ablate.sh -> profiles/r06_bn_nt.txt): the fp32 output always streams (nothing reads it before the
  // backward pass); the operand twin streams when it is too large for the next contraction to find it in the L2s anyway -- the
  // three planes of the emulated arithmetic at cfg2 (12 MB: bn_act_forward 9.1 -> 7.8 us, hb_apply 9.8 -> 8.9 us, the step -8 ..
  // -10 us), bf16 at cfg4 (16 MB) -- and stays cached when it fits (bf16 at cfg3, 4 MB: streamed, the pass gains 1 us and the
  // contraction that reads it next loses 1.2 + 0.7 us).  Streaming loads on top (bit 2) change nothing measurable.
  static const int forced = [] { const char* q = getenv("TFK_BN_NT"); return q ? atoi(q) : -1; }();
  if (forced >= 0) return forced;
  const size_t twin_bytes = tw.p ? (size_t)rows * ld * (tw.x3 ? 6 : 2) : 0;
  return 1 | (twin_bytes >= ((size_t)8 << 20) ? 2 : 0);
}
__device__ __forceinline__ float& el(float4& v, int i) { return reinterpret_cast<float*>(&v)[i]; }
__device__ __forceinline__ float el(const float4& v, int i) { return reinterpret_cast<const float*>(&v)[i]; }

// ------------------------------------------------------------------------------------------------
// Column-tiled geometry: block = 32 float4-columns (128 columns) x 8 row lanes; grid = (column
// blocks, row splits).  A half-wave reads 512 contiguous bytes of one row.
// ------------------------------------------------------------------------------------------------
constexpr int CT_X = 32, CT_Y = 8;
constexpr int RB = 4;  // rows a thread loads as one batch (its loads are issued back to back: these kernels
                       // are latency-bound, not bandwidth-bound, on L2-resident [T, H] activations)

struct ColTile {
  int c4;      // float4 column index
  int col;     // first column
  bool valid;  // col < ld
  int r0, r1;  // row range of this block
};
__device__ __forceinline__ ColTile col_tile(int T, int ld, int rows_per) {
  ColTile t;
  t.c4 = blockIdx.x * CT_X + threadIdx.x;
  t.col = t.c4 * 4;
  t.valid = t.col < ld;
  t.r0 = blockIdx.y * rows_per;
  t.r1 = min(T, t.r0 + rows_per);
  return t;
}
// Sum a float4 over the 8 row lanes; result valid in threadIdx.y == 0.
__device__ __forceinline__ float4 reduce_rows(float4 v, float4 (*sm)[CT_X]) {
  sm[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.y == 0) {
#pragma unroll
    for (int k = 0; k < CT_Y; ++k) {
      const float4 q = sm[k][threadIdx.x];
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
  }
  __syncthreads();
  return t;
}

// ---- batch-norm forward statistics ----
__global__ void __launch_bounds__(CT_X * CT_Y)
bn_stats_partial_kernel(const float* __restrict__ z, int T, int ld, int rows_per, int rs, float* __restrict__ ws) {
  __shared__ float4 sm[CT_Y][CT_X];
  __shared__ float4 sm_mean[CT_X];
  const ColTile t = col_tile(T, ld, rows_per);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s = zero4;
  float4 keep[RB];  // the first row batch stays in registers for the second pass (all of it when T <= 2048)
#pragma unroll
  for (int j = 0; j < RB; ++j) keep[j] = zero4;
  if (t.valid) {
    const int rfirst = t.r0 + threadIdx.y;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = rfirst + j * CT_Y;
      if (r < t.r1) keep[j] = ld4(z + (size_t)r * ld + t.col);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) { s.x += keep[j].x; s.y += keep[j].y; s.z += keep[j].z; s.w += keep[j].w; }
    for (int r = rfirst + RB * CT_Y; r < t.r1; r += CT_Y) {
      const float4 v = ld4(z + (size_t)r * ld + t.col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  s = reduce_rows(s, sm);
  const int n = max(t.r1 - t.r0, 0);
  if (threadIdx.y == 0) {
    const float inv = n > 0 ? 1.f / (float)n : 0.f;
    sm_mean[threadIdx.x] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
  __syncthreads();
  const float4 mu = sm_mean[threadIdx.x];
  float4 q = zero4;
  if (t.valid) {
    const int rfirst = t.r0 + threadIdx.y;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (rfirst + j * CT_Y < t.r1) {
        const float dx = keep[j].x - mu.x, dy = keep[j].y - mu.y, dz = keep[j].z - mu.z, dw = keep[j].w - mu.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    }
    for (int r = rfirst + RB * CT_Y; r < t.r1; r += CT_Y) {
      const float4 v = ld4(z + (size_t)r * ld + t.col);
      const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
  }
  q = reduce_rows(q, sm);
  if (threadIdx.y == 0 && t.valid) {
    st4(ws + ((size_t)0 * rs + blockIdx.y) * ld + t.col, mu);
    st4(ws + ((size_t)1 * rs + blockIdx.y) * ld + t.col, q);
  }
}

// geometry of the "final" kernels that combine the per-chunk partials of a column
constexpr int FIN_COLS = 64, FIN_KL = 16, FIN_PER = kMaxRowSplits / FIN_KL;

__global__ void __launch_bounds__(FIN_COLS * FIN_KL)
bn_stats_final_kernel(const float* __restrict__ ws, int T, int H, int ld, int rows_per, int rs,
                                      float eps, float decay, float* __restrict__ mean, float* __restrict__ rstd,
                                      float* __restrict__ e_mean, float* __restrict__ e_var) {
  // block = 64 columns x 16 chunk lanes: every thread issues its <= 16 chunk loads back to back (one
  // memory round trip), then the sixteen lanes of a column combine through LDS in a fixed order.
  __shared__ float sm[FIN_KL][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x;
  const int ky = threadIdx.y;
  const bool live = c < H;
  float mk[FIN_PER], qk[FIN_PER], nk[FIN_PER];
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) {
    const int k = ky + j * FIN_KL;
    const bool ok = live && k < rs;
    nk[j] = ok ? (float)max(min(T, (k + 1) * rows_per) - k * rows_per, 0) : 0.f;
    mk[j] = ok ? ws[((size_t)0 * rs + k) * ld + c] : 0.f;
    qk[j] = ok ? ws[((size_t)1 * rs + k) * ld + c] : 0.f;
  }
  // Chan et al. merge of the per-chunk (n, mean, M2)
  float tot = 0.f;
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) tot += nk[j] * mk[j];
  sm[ky][threadIdx.x] = tot;
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < FIN_KL; ++k) acc += sm[k][threadIdx.x];
  const float mu = acc / (float)T;
  __syncthreads();
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) {
    const float d = mk[j] - mu;
    m2 += qk[j] + nk[j] * d * d;
  }
  sm[ky][threadIdx.x] = m2;
  __syncthreads();
  if (ky != 0 || c >= ld) return;
  if (!live) {  // padding columns stay neutral
    mean[c] = 0.f; rstd[c] = 0.f;
    return;
  }
  acc = 0.f;
#pragma unroll
  for (int k = 0; k < FIN_KL; ++k) acc += sm[k][threadIdx.x];
  const float var = acc / (float)T;
  mean[c] = mu;              // var is biased, as tf.nn.moments
  rstd[c] = rsqrtf(var + eps);
  e_mean[c] = decay * e_mean[c] + (1.f - decay) * mu;
  e_var[c] = decay * e_var[c] + (1.f - decay) * var;
}

__global__ void bn_stats_eval_kernel(const float* __restrict__ mov_mean, const float* __restrict__ mov_var, int H,
                                     float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  mean[c] = mov_mean[c];
  rstd[c] = rsqrtf(mov_var[c] + eps);
}

// ---- activation chain forward (one block walks rows; threads walk float4 columns) ----
template <bool L2>
__global__ void __launch_bounds__(256)
act_forward_kernel(ActDesc d, const float* __restrict__ z, float* __restrict__ a, float* __restrict__ vbuf,
                   float* __restrict__ rowscale, const float* __restrict__ mean, const float* __restrict__ rstd,
                   const float* __restrict__ beta, int T, int H, int ld, Twin tw) {
  __shared__ float sm[4];
  const int nc4 = ld >> 2;
  const bool drop = d.train && d.keep < 1.f;
  const float inv_keep = drop ? 1.f / d.keep : 1.f;
  for (int row = blockIdx.x; row < T; row += gridDim.x) {
    const float* zr = z + (size_t)row * ld;
    float* ar = a + (size_t)row * ld;
    float ss = 0.f;
    for (int c4 = threadIdx.x; c4 < nc4; c4 += blockDim.x) {
      const int col = c4 << 2;
      float4 u = ld4(zr + col);
      if (d.bn) {
        const float4 mu = ld4(mean + col), rs = ld4(rstd + col), be = ld4(beta + col);
        u.x = (u.x - mu.x) * rs.x + be.x; u.y = (u.y - mu.y) * rs.y + be.y;
        u.z = (u.z - mu.z) * rs.z + be.z; u.w = (u.w - mu.w) * rs.w + be.w;
      }
      float4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) el(v, k) = (col + k < H) ? nonlin_fwd(el(u, k), d.nonlin) : 0.f;
      if (L2) {
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        st4(vbuf + (size_t)row * ld + col, v);
      } else {
        if (drop) {
          const Keep4 m = keep_mask(d, row, c4);
#pragma unroll
          for (int k = 0; k < 4; ++k) el(v, k) = m.k[k] ? el(v, k) * inv_keep : 0.f;
        }
        st4(ar + col, v);
        st4_twin(tw, row, col, v);
      }
    }
    if (L2) {
      const float s = block_sum(ss, sm) / (float)H;  // mean SQUARE (not RMS): activation.py:103
      if (threadIdx.x == 0) rowscale[row] = s;
      const float scale = s > 1.f ? 1.f / s : 1.f;
      for (int c4 = threadIdx.x; c4 < nc4; c4 += blockDim.x) {
        const int col = c4 << 2;
        float4 v = ld4(vbuf + (size_t)row * ld + col);  // own writes
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (drop) {
          const Keep4 m = keep_mask(d, row, c4);
#pragma unroll
          for (int k = 0; k < 4; ++k) el(v, k) = m.k[k] ? el(v, k) * inv_keep : 0.f;
        }
        st4(ar + col, v);
        st4_twin(tw, row, col, v);
      }
    }
  }
}

// ---- fused BN (training) + nonlinearity + dropout, column-tiled; statistics from the GEMM epilogue ----
__global__ void __launch_bounds__(CT_X * CT_Y)
bn_act_forward_kernel(ActDesc d, const float* __restrict__ z, float* __restrict__ a, const float* __restrict__ st,
                      int nchunk, int chunk_rows, int T, int H, int ld, int rows_per, float eps, float decay,
                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ e_mean,
                      float* __restrict__ e_var, const float* __restrict__ beta, Twin tw, int T_apply, int slab, int nt) {
  // T rows carry the statistics; rows [T, T_apply) are the padding behind a segment of a stacked pass: they get
  // (finite) outputs from the segment's statistics and count for nothing.  slab = chunks per statistics slab.
  __shared__ float4 sm[CT_Y][CT_X];
  __shared__ float4 smm[CT_X];
  const ColTile t = col_tile(T_apply, ld, rows_per);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Everything this block reads is requested up front -- the chunk statistics (up to KREG chunk pairs per thread
  // stay in registers) and the first batch of z rows -- so that the kernel pays ONE memory round trip before its
  // two LDS reductions instead of three dependent ones (these launches are latency-bound: ~7 us for 16 MB).
  constexpr int KREG = 2;
  const bool in_regs = nchunk <= KREG * CT_Y;  // block-uniform
  float4 mreg[KREG], qreg[KREG];
  float nreg[KREG];
#pragma unroll
  for (int j = 0; j < KREG; ++j) {
    const int k = threadIdx.y + j * CT_Y;
    const bool ok = t.valid && k < nchunk;
    nreg[j] = ok ? (float)max(min(T, (k + 1) * chunk_rows) - k * chunk_rows, 0) : 0.f;
    mreg[j] = ok ? ld4(st + ((size_t)0 * slab + k) * ld + t.col) : zero4;
    qreg[j] = ok ? ld4(st + ((size_t)1 * slab + k) * ld + t.col) : zero4;
  }
  const int rb0 = t.r0 + threadIdx.y;
  float4 zv[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int r = rb0 + j * CT_Y;
    zv[j] = (t.valid && rb0 < t.r1) ? ld4s(z + (size_t)(r < t.r1 ? r : rb0) * ld + t.col, nt) : zero4;
  }
  const float4 be = t.valid ? ld4(beta + t.col) : zero4;
  // Chan merge of the per-chunk (n, mean, M2): row lane y takes chunks y, y + 8, ...
  float4 tot = zero4;
#pragma unroll
  for (int j = 0; j < KREG; ++j) {
    tot.x += nreg[j] * mreg[j].x; tot.y += nreg[j] * mreg[j].y;
    tot.z += nreg[j] * mreg[j].z; tot.w += nreg[j] * mreg[j].w;
  }
  if (!in_regs && t.valid)
    for (int k = threadIdx.y + KREG * CT_Y; k < nchunk; k += CT_Y) {
      const float n = (float)max(min(T, (k + 1) * chunk_rows) - k * chunk_rows, 0);
      const float4 m = ld4(st + ((size_t)0 * slab + k) * ld + t.col);
      tot.x += n * m.x; tot.y += n * m.y; tot.z += n * m.z; tot.w += n * m.w;
    }
  tot = reduce_rows(tot, sm);
  if (threadIdx.y == 0) {
    const float inv = 1.f / (float)T;
    smm[threadIdx.x] = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
  }
  __syncthreads();
  const float4 mu = smm[threadIdx.x];
  float4 m2 = zero4;
#pragma unroll
  for (int j = 0; j < KREG; ++j) {
    const float4 m = mreg[j], q = qreg[j];
    const float n = nreg[j];
    m2.x += q.x + n * (m.x - mu.x) * (m.x - mu.x); m2.y += q.y + n * (m.y - mu.y) * (m.y - mu.y);
    m2.z += q.z + n * (m.z - mu.z) * (m.z - mu.z); m2.w += q.w + n * (m.w - mu.w) * (m.w - mu.w);
  }
  if (!in_regs && t.valid)
    for (int k = threadIdx.y + KREG * CT_Y; k < nchunk; k += CT_Y) {
      const float n = (float)max(min(T, (k + 1) * chunk_rows) - k * chunk_rows, 0);
      const float4 m = ld4(st + ((size_t)0 * slab + k) * ld + t.col);
      const float4 q = ld4(st + ((size_t)1 * slab + k) * ld + t.col);
      m2.x += q.x + n * (m.x - mu.x) * (m.x - mu.x); m2.y += q.y + n * (m.y - mu.y) * (m.y - mu.y);
      m2.z += q.z + n * (m.z - mu.z) * (m.z - mu.z); m2.w += q.w + n * (m.w - mu.w) * (m.w - mu.w);
    }
  m2 = reduce_rows(m2, sm);
  if (threadIdx.y == 0) {
    const float inv = 1.f / (float)T;  // biased variance, as tf.nn.moments
    float4 var = make_float4(m2.x * inv, m2.y * inv, m2.z * inv, m2.w * inv);
    float4 rs4;
#pragma unroll
    for (int k = 0; k < 4; ++k) el(rs4, k) = (t.col + k < H) ? rsqrtf(el(var, k) + eps) : 0.f;
    smm[threadIdx.x] = rs4;
    if (blockIdx.y == 0 && t.valid) {  // one block per column tile publishes the statistics
      float4 mu_w = mu;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.col + k >= H) el(mu_w, k) = 0.f;
      st4(mean + t.col, mu_w);
      st4(rstd + t.col, rs4);
      float4 em = ld4(e_mean + t.col), ev = ld4(e_var + t.col);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.col + k < H) {
          el(em, k) = decay * el(em, k) + (1.f - decay) * el(mu, k);
          el(ev, k) = decay * el(ev, k) + (1.f - decay) * el(var, k);
        }
      st4(e_mean + t.col, em);
      st4(e_var + t.col, ev);
    }
  }
  __syncthreads();
  if (!t.valid) return;
  const float4 rsd = smm[threadIdx.x];
  const bool drop = d.train && d.keep < 1.f;
  const float inv_keep = drop ? 1.f / d.keep : 1.f;
  for (int rb = rb0; rb < t.r1; rb += CT_Y * RB) {
    float4 zn[RB];  // next batch in flight while this one is finished
    const int rbn = rb + CT_Y * RB;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = rbn + j * CT_Y;
      zn[j] = rbn < t.r1 ? ld4s(z + (size_t)(r < t.r1 ? r : rbn) * ld + t.col, nt) : zero4;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int r = rb + j * CT_Y;
      if (r >= t.r1) continue;
      float4 v;
      v.x = (zv[j].x - mu.x) * rsd.x + be.x; v.y = (zv[j].y - mu.y) * rsd.y + be.y;
      v.z = (zv[j].z - mu.z) * rsd.z + be.z; v.w = (zv[j].w - mu.w) * rsd.w + be.w;
#pragma unroll
      for (int k = 0; k < 4; ++k) el(v, k) = (t.col + k < H) ? nonlin_fwd(el(v, k), d.nonlin) : 0.f;
      if (drop) {
        const Keep4 m = keep_mask(d, r, t.c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) el(v, k) = m.k[k] ? el(v, k) * inv_keep : 0.f;
      }
      if (nt & 1) st4_nt(a + (size_t)r * ld + t.col, v);
      else st4(a + (size_t)r * ld + t.col, v);
      if (nt & 2) st4_twin_nt(tw, r, t.col, v);
      else st4_twin(tw, r, t.col, v);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) zv[j] = zn[j];
  }
}

// ---- L2 chains: da -> du in place (row-wise) ----
__global__ void __launch_bounds__(256)
act_backward_rows_kernel(ActDesc d, float* __restrict__ da, const float* __restrict__ vbuf,
                         const float* __restrict__ rowscale, int T, int H, int ld) {
  __shared__ float sm[4];
  const int nc4 = ld >> 2;
  const bool drop = d.keep < 1.f;
  const float inv_keep = drop ? 1.f / d.keep : 1.f;
  for (int row = blockIdx.x; row < T; row += gridDim.x) {
    float* gr = da + (size_t)row * ld;
    const float* vr = vbuf + (size_t)row * ld;
    const float s = rowscale[row];
    float dot = 0.f;
    if (s > 1.f) {
      for (int c4 = threadIdx.x; c4 < nc4; c4 += blockDim.x) {
        const int col = c4 << 2;
        float4 g = ld4(gr + col);
        const float4 v = ld4(vr + col);
        if (drop) {
          const Keep4 m = keep_mask(d, row, c4);
#pragma unroll
          for (int k = 0; k < 4; ++k) el(g, k) = m.k[k] ? el(g, k) * inv_keep : 0.f;
        }
        dot += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
      }
      dot = block_sum(dot, sm);
    }
    // w = v / s  =>  dv_i = dw_i / s - v_i * (2 / (H s^2)) * sum_j dw_j v_j      (s = mean_j v_j^2)
    const float k1 = s > 1.f ? 1.f / s : 1.f;
    const float k2 = s > 1.f ? 2.f * dot / ((float)H * s * s) : 0.f;
    for (int c4 = threadIdx.x; c4 < nc4; c4 += blockDim.x) {
      const int col = c4 << 2;
      float4 g = ld4(gr + col);
      const float4 v = ld4(vr + col);
      if (drop) {
        const Keep4 m = keep_mask(d, row, c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) el(g, k) = m.k[k] ? el(g, k) * inv_keep : 0.f;
      }
      float4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dv = el(g, k) * k1 - el(v, k) * k2;
        el(o, k) = (col + k < H) ? dv * nonlin_bwd(el(v, k), d.nonlin) : 0.f;
      }
      st4(gr + col, o);
    }
  }
}

// du for the no-L2 chains, recomputed from (da, a) wherever it is needed
__device__ __forceinline__ float4 compute_du(const ActDesc& d, int pre_du, float4 g, float4 a, int row, int c4) {
  if (pre_du) return g;
  float4 o;
  if (d.keep < 1.f) {
    const Keep4 m = keep_mask(d, row, c4);
    const float inv_keep = 1.f / d.keep;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      el(o, k) = m.k[k] ? el(g, k) * inv_keep * nonlin_bwd(el(a, k) * d.keep, d.nonlin) : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) el(o, k) = el(g, k) * nonlin_bwd(el(a, k), d.nonlin);
  }
  return o;
}

// pass A (BN only): partial sums of du and du * xhat
__global__ void __launch_bounds__(CT_X * CT_Y)
hb_stats_kernel(ActDesc d, int pre_du, const float* __restrict__ da, const float* __restrict__ a,
                const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd, int T,
                int ld, int rows_per, int rs, float* __restrict__ ws) {
  __shared__ float4 sm[CT_Y][CT_X];
  const ColTile t = col_tile(T, ld, rows_per);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (t.valid) {
    const float4 mu = ld4(mean + t.col), rsd = ld4(rstd + t.col);
    for (int rb = t.r0 + threadIdx.y; rb < t.r1; rb += CT_Y * RB) {
      float4 g[RB], av[RB], zv[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = rb + j * CT_Y;
        const size_t off = (size_t)(r < t.r1 ? r : rb) * ld + t.col;
        g[j] = ld4(da + off);
        av[j] = pre_du ? g[j] : ld4(a + off);
        zv[j] = ld4(z + off);
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = rb + j * CT_Y;
        if (r >= t.r1) continue;
        const float4 du = compute_du(d, pre_du, g[j], av[j], r, t.c4);
        s1.x += du.x; s1.y += du.y; s1.z += du.z; s1.w += du.w;
        s2.x += du.x * (zv[j].x - mu.x) * rsd.x; s2.y += du.y * (zv[j].y - mu.y) * rsd.y;
        s2.z += du.z * (zv[j].z - mu.z) * rsd.z; s2.w += du.w * (zv[j].w - mu.w) * rsd.w;
      }
    }
  }
  s1 = reduce_rows(s1, sm);
  s2 = reduce_rows(s2, sm);
  if (threadIdx.y == 0 && t.valid) {
    st4(ws + ((size_t)0 * kMaxRowSplits + blockIdx.y) * ld + t.col, s1);
    st4(ws + ((size_t)1 * kMaxRowSplits + blockIdx.y) * ld + t.col, s2);
  }
}

// pass B: dz in place (+ partial column sums of dz).  rs_in = number of pass-A chunks in slabs 0 / 1 (written by
// hb_stats_kernel, or by the dA GEMM's EPI_DACT epilogue: then pre_du = 1 and `da` already holds du)
__global__ void __launch_bounds__(CT_X * CT_Y)
hb_apply_kernel(ActDesc d, int pre_du, float* __restrict__ da, const float* __restrict__ a,
                const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd, int T,
                int H, int ld, int rows_per, int rs_in, float* __restrict__ ws, Twin tw, int T_apply,
                float* __restrict__ ws_dz, int nt) {
  // rows [T, T_apply): padding behind a segment of a stacked pass -- their dz is written as ZERO (both contractions
  // that consume dz run over the padded rows) and they count for nothing.  ws_dz: where slab 2 of this launch goes.
  __shared__ float4 sm[CT_Y][CT_X];
  const ColTile t = col_tile(T_apply, ld, rows_per);
  float4 sz = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 zero4 = sz;
  float4 m1 = sz, m2 = sz;
  // first batch of rows requested before the statistics prologue: one memory round trip, not two
  const int rb0 = t.r0 + threadIdx.y;
  float4 g[RB], av[RB], zv[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int r = rb0 + j * CT_Y;
    const bool ok = t.valid && rb0 < t.r1;
    const size_t off = (size_t)(r < t.r1 ? r : rb0) * ld + t.col;
    g[j] = ok ? ld4s(da + off, nt) : zero4;
    av[j] = (ok && !pre_du) ? ld4s(a + off, nt) : g[j];
    zv[j] = (ok && d.bn) ? ld4s(z + off, nt) : g[j];
  }
  float4 mu = sz, rsd = sz;
  if (d.bn && t.valid) {
    mu = ld4(mean + t.col); rsd = ld4(rstd + t.col);
  }
  if (d.bn) {
    // column means of du and du*xhat from pass A's per-chunk partials: row lane y sums chunks y, y+8, ...
    // (<= 8 independent loads each), the block combines them through LDS
    float4 p1 = sz, p2 = sz;
    if (t.valid)
      for (int k = threadIdx.y; k < rs_in; k += CT_Y) {
        const float4 q1 = ld4(ws + ((size_t)0 * kMaxRowSplits + k) * ld + t.col);
        const float4 q2 = ld4(ws + ((size_t)1 * kMaxRowSplits + k) * ld + t.col);
        p1.x += q1.x; p1.y += q1.y; p1.z += q1.z; p1.w += q1.w;
        p2.x += q2.x; p2.y += q2.y; p2.z += q2.z; p2.w += q2.w;
      }
    __shared__ float4 smm[2][CT_X];
    p1 = reduce_rows(p1, sm);
    p2 = reduce_rows(p2, sm);
    if (threadIdx.y == 0) {
      const float invT = 1.f / (float)T;
      smm[0][threadIdx.x] = make_float4(p1.x * invT, p1.y * invT, p1.z * invT, p1.w * invT);
      smm[1][threadIdx.x] = make_float4(p2.x * invT, p2.y * invT, p2.z * invT, p2.w * invT);
    }
    __syncthreads();
    m1 = smm[0][threadIdx.x];
    m2 = smm[1][threadIdx.x];
  }
  if (t.valid) {
    for (int rb = rb0; rb < t.r1; rb += CT_Y * RB) {
      float4 gn[RB], an[RB], zn[RB];  // next batch in flight (other rows: da is rewritten in place below)
      const int rbn = rb + CT_Y * RB;
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = rbn + j * CT_Y;
        const bool ok = rbn < t.r1;
        const size_t off = (size_t)(r < t.r1 ? r : rbn) * ld + t.col;
        gn[j] = ok ? ld4s(da + off, nt) : zero4;
        an[j] = (ok && !pre_du) ? ld4s(a + off, nt) : gn[j];
        zn[j] = (ok && d.bn) ? ld4s(z + off, nt) : gn[j];
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = rb + j * CT_Y;
        if (r >= t.r1) continue;
        float4 dz = compute_du(d, pre_du, g[j], av[j], r, t.c4);
        if (d.bn) {
          // dz = rstd * (du - mean(du) - xhat * mean(du * xhat))
          dz.x = rsd.x * (dz.x - m1.x - (zv[j].x - mu.x) * rsd.x * m2.x);
          dz.y = rsd.y * (dz.y - m1.y - (zv[j].y - mu.y) * rsd.y * m2.y);
          dz.z = rsd.z * (dz.z - m1.z - (zv[j].z - mu.z) * rsd.z * m2.z);
          dz.w = rsd.w * (dz.w - m1.w - (zv[j].w - mu.w) * rsd.w * m2.w);
        }
        if (r >= T) dz = zero4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t.col + k >= H) el(dz, k) = 0.f;
        // mixed precision: both contractions that consume dz read its bf16 twin, nothing reads the fp32 copy again
        // (the bias gradient is the column sum taken right here): it is not stored -- a third of this kernel's traffic
        if (tw.p) {
          if (nt & 2) st4_twin_nt(tw, r, t.col, dz);
          else st4_twin(tw, r, t.col, dz);
        } else {
          st4(da + (size_t)r * ld + t.col, dz);
        }
        sz.x += dz.x; sz.y += dz.y; sz.z += dz.z; sz.w += dz.w;
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        g[j] = gn[j]; av[j] = an[j]; zv[j] = zn[j];
      }
    }
  }
  sz = reduce_rows(sz, sm);
  if (threadIdx.y == 0 && t.valid) st4(ws_dz + (size_t)blockIdx.y * ld + t.col, sz);
}

// g[c] (+)= sum over row splits of slab `which`, for a batch of (layer, vector) items: blockIdx.y = item
__global__ void __launch_bounds__(FIN_COLS * FIN_KL) grad_final_kernel(FinalBatch b) {
  __shared__ float sm[FIN_KL][FIN_COLS];
  const FinalItem it = b.it[blockIdx.y];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x;
  const int ky = threadIdx.y;
  float v[FIN_PER];
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) {
    const int k = ky + j * FIN_KL;
    v[j] = (c < it.N && k < it.rs) ? it.ws[((size_t)it.which * kMaxRowSplits + k) * it.ld + c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) s += v[j];
  sm[ky][threadIdx.x] = s;
  __syncthreads();
  if (ky != 0 || c >= it.N) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < FIN_KL; ++k) s += sm[k][threadIdx.x];
  it.g[c] = b.accumulate ? it.g[c] + s : s;
}

// Tall micro-batches: the two per-chunk partial sums of batch-norm's backward (slabs 0 and 1, `chunks` entries each)
// are reduced ONCE into entry 0 of their slab, so that the column-tiled apply kernel's blocks need not each re-read
// every chunk.  blockIdx.y = slab.
__global__ void __launch_bounds__(FIN_COLS * FIN_KL) chunk_totals_kernel(float* __restrict__ ws, int chunks, int ld) {
  __shared__ float sm[FIN_KL][FIN_COLS];
  const int c = blockIdx.x * FIN_COLS + threadIdx.x;
  const int ky = threadIdx.y;
  float* slab = ws + (size_t)blockIdx.y * kMaxRowSplits * ld;
  float v[FIN_PER];
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) {
    const int k = ky + j * FIN_KL;
    v[j] = (c < ld && k < chunks) ? slab[(size_t)k * ld + c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < FIN_PER; ++j) s += v[j];
  sm[ky][threadIdx.x] = s;
  __syncthreads();
  if (ky != 0 || c >= ld) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < FIN_KL; ++k) s += sm[k][threadIdx.x];
  slab[c] = s;
}

__global__ void __launch_bounds__(CT_X * CT_Y)
colsum_partial_kernel(const float* __restrict__ x, int T, int ld, int rows_per, int rs, float* __restrict__ ws) {
  __shared__ float4 sm[CT_Y][CT_X];
  const ColTile t = col_tile(T, ld, rows_per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t.valid)
    for (int r = t.r0 + threadIdx.y; r < t.r1; r += CT_Y) {
      const float4 v = ld4(x + (size_t)r * ld + t.col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  s = reduce_rows(s, sm);
  if (threadIdx.y == 0 && t.valid) st4(ws + (size_t)blockIdx.y * ld + t.col, s);
}

// the frames' losses summed by the 256 threads of ONE block (tid = flat thread index): strided partial sums, wave shuffles, the
// four wave sums added in wave order -- the same sequence of additions in loss_reduce_kernel and in colsum_loss_kernel, so
// training and evaluation report bit-identical sums for the same rows
__device__ __forceinline__ float loss_sum_256(const float* __restrict__ row_loss, int T, int tid, float* sm4) {
  float s = 0.f;
  for (int i = tid; i < T; i += 256) s += row_loss[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) sm4[tid >> 6] = s;
  __syncthreads();
  return ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
}
__device__ __forceinline__ void loss_commit(float s, float* __restrict__ scalars, int overwrite, float frames, float microbatches) {
  // overwrite: first micro-batch since the accumulators were (logically) re-initialised -- saves the memset
  scalars[0] = overwrite ? s : scalars[0] + s;
  scalars[1] = overwrite ? frames : scalars[1] + frames;
  scalars[2] = overwrite ? microbatches : scalars[2] + microbatches;
}
// Training: the column sums of dLogits (the output layer's bias gradient) and the sum of the frames' losses are both due right
// behind softmax_xent and independent of each other: ONE launch -- the column-sum grid plus a row of blocks (blockIdx.y == rs)
// whose first block sums the losses.  (Two launches of ~4.5 us each before.)
__global__ void __launch_bounds__(CT_X * CT_Y)
colsum_loss_kernel(const float* __restrict__ x, int T, int ld, int rows_per, int rs, float* __restrict__ ws,
                   const float* __restrict__ row_loss, int T_loss, float* __restrict__ scalars, int overwrite, float frames,
                   float microbatches) {
  __shared__ float4 sm[CT_Y][CT_X];
  if ((int)blockIdx.y == rs) {
    if (blockIdx.x != 0) return;
    const int tid = threadIdx.y * CT_X + threadIdx.x;
    const float s = loss_sum_256(row_loss, T_loss, tid, reinterpret_cast<float*>(sm));
    if (tid == 0) loss_commit(s, scalars, overwrite, frames, microbatches);
    return;
  }
  const ColTile t = col_tile(T, ld, rows_per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t.valid)
    for (int r = t.r0 + threadIdx.y; r < t.r1; r += CT_Y) {
      const float4 v = ld4(x + (size_t)r * ld + t.col);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  s = reduce_rows(s, sm);
  if (threadIdx.y == 0 && t.valid) st4(ws + (size_t)blockIdx.y * ld + t.col, s);
}

// ---- softmax cross-entropy: one 256-thread block per frame, the row held in registers ----
template <int NV>  // float4 per thread; NV == 0: generic (re-reads global)
__global__ void __launch_bounds__(256)
softmax_xent_kernel(float* __restrict__ logits, const int32_t* __restrict__ y, int O, int ld,
                    float* __restrict__ row_loss, int with_grad, Twin tw) {
  __shared__ float sm[4];
  const int row = blockIdx.x;
  float* zr = logits + (size_t)row * ld;
  const int label = y[row];
  const int nc4 = ld >> 2;
  if (label < 0) {
    // a row that holds no frame (padding between the segments of a stacked pass): no loss, and a ZERO gradient row --
    // everything backward computes from it is then zero as well
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (with_grad) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c4 = threadIdx.x; c4 < nc4; c4 += 256) {
        st4(zr + (c4 << 2), z4);
        st4_twin(tw, row, c4 << 2, z4);
      }
    }
    return;
  }
  const float zy = label < O ? zr[label] : 0.f;
  __syncthreads();  // zr[label] is read before anyone overwrites it
  constexpr int NVR = NV > 0 ? NV : 1;
  float4 v[NVR];
  float mx = -INFINITY;
  if (NV > 0) {
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int c4 = threadIdx.x + j * 256;
      v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (c4 < nc4) {
        const float4 t = ld4(zr + (c4 << 2));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((c4 << 2) + k < O) el(v[j], k) = el(t, k);
      }
      mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    }
  } else {
    for (int c = threadIdx.x; c < O; c += 256) mx = fmaxf(mx, zr[c]);
  }
  mx = block_max(mx, sm);
  float se = 0.f;
  if (NV > 0) {
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = expf(el(v[j], k) - mx);  // exp(-inf) = 0 for the masked tail
        el(v[j], k) = e;
        se += e;
      }
    }
  } else {
    for (int c = threadIdx.x; c < O; c += 256) se += expf(zr[c] - mx);
  }
  se = block_sum(se, sm);
  if (threadIdx.x == 0) row_loss[row] = (mx + logf(se)) - zy;  // summed by colsum_loss_kernel (training) / loss_reduce_kernel (evaluation)
  if (with_grad) {
    const float inv = 1.f / se;
    if (NV > 0) {
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        const int c4 = threadIdx.x + j * 256;
        if (c4 < nc4) {
          float4 g;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = (c4 << 2) + k;
            el(g, k) = c < O ? el(v[j], k) * inv - (c == label ? 1.f : 0.f) : 0.f;
          }
          st4(zr + (c4 << 2), g);
          st4_twin(tw, row, c4 << 2, g);
        }
      }
    } else {
      for (int c = threadIdx.x; c < O; c += 256) {
        const float g = expf(zr[c] - mx) * inv - (c == label ? 1.f : 0.f);
        zr[c] = g;
        if (tw.p) twin_put(tw, (size_t)row, c, g);
      }
    }
  }
}

__global__ void __launch_bounds__(256) loss_reduce_kernel(const float* __restrict__ row_loss, int T,
                                                          float* __restrict__ scalars, int overwrite, float frames,
                                                          float microbatches) {
  __shared__ float sm[4];
  const float s = loss_sum_256(row_loss, T, threadIdx.x, sm);
  if (threadIdx.x == 0) loss_commit(s, scalars, overwrite, frames, microbatches);
}

// decoder.py:44 softmax (prior == nullptr) or nnet.py:280-286 log(softmax / prior), one 256-thread block per frame;
// like softmax_xent_kernel the row is read ONCE (16-byte loads) and stays in registers for the three passes.
template <int NV>  // float4 per thread; NV == 0: generic (re-reads global)
__global__ void __launch_bounds__(256)
softmax_rows_kernel(const float* __restrict__ logits, int O, int ld, float* __restrict__ out, int64_t ldo,
                    const float* __restrict__ prior, int vec_out) {
  __shared__ float sm[4];
  const int row = blockIdx.x;
  const float* zr = logits + (size_t)row * ld;
  float* orow = out + (size_t)row * ldo;
  const int nc4 = ld >> 2;
  constexpr int NVR = NV > 0 ? NV : 1;
  float4 v[NVR];
  float mx = -INFINITY;
  if (NV > 0) {
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int c4 = threadIdx.x + j * 256;
      v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (c4 < nc4) {
        const float4 t = ld4(zr + (c4 << 2));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((c4 << 2) + k < O) el(v[j], k) = el(t, k);
      }
      mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    }
  } else {
    for (int c = threadIdx.x; c < O; c += 256) mx = fmaxf(mx, zr[c]);
  }
  mx = block_max(mx, sm);
  float se = 0.f;
  float4 ex[NVR];
  if (NV > 0) {
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = expf(el(v[j], k) - mx);  // exp(-inf) = 0 for the masked tail
        el(ex[j], k) = e;
        se += e;
      }
    }
  } else {
    for (int c = threadIdx.x; c < O; c += 256) se += expf(zr[c] - mx);
  }
  se = block_sum(se, sm);
  const float lse = mx + logf(se), inv = 1.f / se;
  if (NV > 0) {
#pragma unroll
    for (int j = 0; j < NVR; ++j) {
      const int c4 = threadIdx.x + j * 256, c = c4 << 2;
      if (c >= O) continue;
      float4 r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // z - lse - log prior instead of log(softmax / prior): finite where the posterior underflows
        if (prior) el(r, k) = c + k < O ? (el(v[j], k) - lse) - logf(prior[c + k]) : 0.f;
        else el(r, k) = el(ex[j], k) * inv;
      }
      if (vec_out && c + 3 < O) {
        st4(orow + c, r);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < O) orow[c + k] = el(r, k);
      }
    }
  } else {
    if (prior) {
      for (int c = threadIdx.x; c < O; c += 256) orow[c] = (zr[c] - lse) - logf(prior[c]);
    } else {
      for (int c = threadIdx.x; c < O; c += 256) orow[c] = expf(zr[c] - mx) * inv;
    }
  }
}

// ---- mean -> clip -> Adam (TF formulation), zeroing the gradient sum on the way out ----
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4s(const float* p) {
  if constexpr (NT) {
    const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
  } else {
    return ld4(p);
  }
}
template <bool NT>
__device__ __forceinline__ void st4s(float* p, float4 v) {
  if constexpr (NT) {
    v4f q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(p));
  } else {
    st4(p, v);
  }
}
// NT: 1 = streaming (non-temporal) accesses for g / m / v, which nothing re-reads before the next optimiser step (default);
//     2 = also for the parameters (read back by the next optimiser step only, when the GEMMs read operand twins); 3 = also for the
//     twins this kernel writes -- measured 131.5 -> 164.4 us at cfg2 (profiles/r06_bn_nt.txt): a thread's four weights are HALF of
//     a plane's 128-byte line (the other row of the unit belongs to a thread 2048 weights away), and a streamed half line is
//     not merged in the L2 the way a cached one is
// UN: float4 groups per thread per trip (their loads are all issued before the first is consumed)
template <int NT, int UN>
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4,
            const float* __restrict__ scalars, float lr_t, float b1, float b2, float eps, uint16_t* __restrict__ wb,
            size_t n4_wb, ShadowMap map, size_t first) {
  // a step without frames (G / 0): the reference would write NaN into every parameter; here the parameters are
  // left alone and tfk_apply_end reports the error
  if (!(scalars[1] > 0.f)) return;
  const float inv_n = 1.f / scalars[1];  // G / float(num_frames): trainer.py:174-175
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UN) {
    float4 gv[UN], mv[UN], vv[UN], wv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        gv[u] = ld4s<(NT > 0)>(g + 4 * i); mv[u] = ld4s<(NT > 0)>(m + 4 * i); vv[u] = ld4s<(NT > 0)>(v + 4 * i);
        wv[u] = ld4s<(NT > 1)>(w + 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t i = i0 + u * stride;
      if (i >= n4) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // Every operation is spelled out with its rounding: the result of an element must not depend on which unrolled
        // body or which launch geometry processed it (the compiler contracted the two bodies differently -- 1-ulp
        // differences between a span-wise and a whole-arena update, i.e. between the sharded and the serial step).
        const float gk = fminf(fmaxf(__fmul_rn(el(gv[u], k), inv_n), -1.f), 1.f);  // clip_by_value: trainer.py:178-179
        const float mk = __fmaf_rn(b1, el(mv[u], k), __fmul_rn(1.f - b1, gk));
        const float vk = __fmaf_rn(b2, el(vv[u], k), __fmul_rn(1.f - b2, __fmul_rn(gk, gk)));
        el(mv[u], k) = mk;
        el(vv[u], k) = vk;
        el(wv[u], k) = __fsub_rn(el(wv[u], k), __fdiv_rn(__fmul_rn(lr_t, mk), __fadd_rn(__fsqrt_rn(vk), eps)));
      }
      st4s<(NT > 0)>(m + 4 * i, mv[u]);
      st4s<(NT > 0)>(v + 4 * i, vv[u]);
      st4s<(NT > 1)>(w + 4 * i, wv[u]);
      if (wb && i < n4_wb) {  // bf16 shadow of the weight matrices
        Twin sh;
        if (map.n == 0) {  // same element offsets as the fp32 arena
          sh.p = wb; sh.ld = 1;
          if constexpr (NT > 2) st4_twin_nt(sh, 4 * i, 0, wv[u]);
          else st4_twin(sh, 4 * i, 0, wv[u]);  // (leading dimension 1: the "row" is the flat index)
        } else {  // x3: the tiled twin of the matrix this group of four belongs to (padding between matrices: none)
          const uint32_t e = (uint32_t)(first + 4 * i);
          int l = 0;
          while (l + 1 < map.n && e >= map.begin[l + 1]) ++l;
          const uint32_t rel = e - map.begin[l], r = rel / map.ld[l];
          if (r < map.rows[l]) {
            sh.p = wb + map.twin[l]; sh.ld = (int)map.ld_twin[l]; sh.x3 = 1;
            if constexpr (NT > 2) st4_twin_nt(sh, r, (int)(rel - r * map.ld[l]), wv[u]);
            else st4_twin(sh, r, (int)(rel - r * map.ld[l]), wv[u]);
          }
        }
      }
      // init_grads (trainer.py:350) costs no traffic: the next step's first micro-batch overwrites G
    }
  }
}

// End of an optimiser step, one launch: BN moving averages mov = decay^k * mov + increments (k = micro-batches of
// the step), re-initialisation of the increments, and the step's (loss, frames, k) handed to the host through
// mapped pinned memory (no copy kernel, no memset).
// `seq` (never 0) goes into word kStepSeqWord of the mapped memory BEHIND the four scalars (one thread writes all five, a
// system-scope fence in between): a host that polls that word has the step's loss without an event on the stream -- an event
// record between this launch and the optimiser's cost the stream 5.9 us of idle time per step (profiles/r06_loss_seq.txt).
__global__ void step_finish_kernel(float* __restrict__ mov, float* __restrict__ e, size_t n,
                                   const float* __restrict__ scalars, float decay, float* __restrict__ host,
                                   float* __restrict__ snap, unsigned seq) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    mov[i] = powf(decay, scalars[2]) * mov[i] + e[i];
    e[i] = 0.f;
  }
  if (i < 4 && snap) snap[i] = scalars[i];  // the step's (loss, frames, k) for an optimiser that runs past the next loss_reduce
  if (i == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) host[k] = scalars[k];
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned*>(host) + kStepSeqWord, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// fp32 [rows, lds] -> bf16 [rows, ldd] (ldd multiple of 8): one thread per 8-column chunk, padding columns zero
__global__ void __launch_bounds__(256)
to_bf16_rows_kernel(const float* __restrict__ src, int lds, uint16_t* __restrict__ dst, int ldd, int rows, int cols,
                    int x3) {
  const int nc8 = ldd >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)rows * nc8) return;
  const int r = (int)(idx / nc8), c = (int)(idx % nc8) << 3;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto pack = [](const uint16_t (&o)[8]) {
    u32x4 q;
    q.x = o[0] | ((uint32_t)o[1] << 16); q.y = o[2] | ((uint32_t)o[3] << 16);
    q.z = o[4] | ((uint32_t)o[5] << 16); q.w = o[6] | ((uint32_t)o[7] << 16);
    return q;
  };
  if (x3) {
    uint16_t o1[8], o2[8], o3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) twin_split3((c + k < cols) ? src[(size_t)r * lds + c + k] : 0.f, o1[k], o2[k], o3[k]);
    uint16_t* d = dst + x3::at((size_t)r, c, ldd);
    *reinterpret_cast<u32x4*>(d) = pack(o1);
    *reinterpret_cast<u32x4*>(d + 64) = pack(o2);
    *reinterpret_cast<u32x4*>(d + 128) = pack(o3);
    return;
  }
  uint16_t o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (c + k < cols) ? to_bf16(src[(size_t)r * lds + c + k]) : (uint16_t)0;
  *reinterpret_cast<u32x4*>(dst + (size_t)r * ldd + c) = pack(o);
}
__global__ void scale_kernel(float* __restrict__ x, size_t n, float f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= f;
}
__global__ void fill_kernel(float* __restrict__ x, size_t n, float f) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = f;
}
__global__ void dropout_mask_kernel(ActDesc d, float* __restrict__ out, int T, int H, int ld) {
  const int nc4 = ld >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * nc4) return;
  const int row = (int)(idx / nc4), c4 = (int)(idx % nc4);
  const Keep4 m = keep_mask(d, row, c4);
  float4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) el(o, k) = ((c4 << 2) + k < H && m.k[k]) ? 1.f : 0.f;
  st4(out + (size_t)row * ld + (c4 << 2), o);
}

// one thread per output element group of 4 (scalar inside: D need not be a multiple of 4)
__global__ void __launch_bounds__(256)
splice_kernel(const float* __restrict__ raw, int ldr, const int32_t* __restrict__ seg, int U, int T, int D, int context,
              const float* __restrict__ cmvn, float* __restrict__ out, int ldo, const int32_t* __restrict__ out_seg) {
  const int nc4 = ldo >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * nc4) return;
  const int t = (int)(idx / nc4), c4 = (int)(idx % nc4);
  // utterance of frame t: largest u with seg[u] <= t (binary search over U + 1 offsets)
  int lo = 0, hi = U;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] <= t) lo = mid; else hi = mid;
  }
  const int first = seg[lo], last = seg[lo + 1];
  const int F = D * (2 * context + 1);
  float4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = (c4 << 2) + k;
    float v = 0.f;
    if (col < F) {
      const int j = col / D, d = col - j * D;
      const int src = t + j - context;
      if (src >= first && src < last) {
        v = raw[(size_t)src * ldr + d];
        if (cmvn) v = __fdiv_rn(__fsub_rn(v, cmvn[(size_t)(2 * lo) * D + d]), cmvn[(size_t)(2 * lo + 1) * D + d]);
      }
    }
    el(o, k) = v;
  }
  // out_seg (stacked pass): utterance u's rows start at out_seg[u] of the output instead of seg[u]
  const int trow = out_seg ? out_seg[lo] + (t - first) : t;
  st4(out + (size_t)trow * ldo + (c4 << 2), o);
}

// position-weighted integer checksum of a span of 32-bit words (replica-consistency check of the data-parallel exchange:
// integer arithmetic, so the value does not depend on the order the blocks finish in)
__global__ void __launch_bounds__(256)
checksum_kernel(const uint32_t* __restrict__ p, size_t n, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    acc += (unsigned long long)p[i] * (unsigned long long)((i & 0xffffu) + 1u);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

inline dim3 ct_grid(int ld, int rs) { return dim3((ld / 4 + CT_X - 1) / CT_X, rs); }
inline dim3 ct_block() { return dim3(CT_X, CT_Y); }

}  // namespace

int row_splits(int T) {
  // rows per block of the column-tiled kernels (env TFK_CT_ROWS, experiments; 32 = four rows per thread, one batch of loads)
  static const int rows = [] { const char* q = getenv("TFK_CT_ROWS"); const int v = q ? atoi(q) : 32; return v >= 8 ? v : 32; }();
  int rs = (T + rows - 1) / rows;
  if (rs < 1) rs = 1;
  if (rs > kMaxRowSplits) rs = kMaxRowSplits;
  return rs;
}

// bn_act_forward alone (env TFK_CT_ROWS_FWD): its row splits feed no partial-sum slab, so its blocks may be cut finer than the
// backward pass's -- with streaming stores 16 rows per block beat 32 there (7.0 against 7.7 us at cfg2) while hb_apply loses
// 2.8 us with them (profiles/r06_bn_nt.txt)
int row_splits_fwd(int T) {
  static const int rows = [] { const char* q = getenv("TFK_CT_ROWS_FWD"); const int v = q ? atoi(q) : 16; return v >= 8 ? v : 16; }();
  if (getenv("TFK_CT_ROWS") && !getenv("TFK_CT_ROWS_FWD")) return row_splits(T);  // (the older switch alone: every column-tiled kernel)
  int rs = (T + rows - 1) / rows;
  if (rs < 1) rs = 1;
  if (rs > kMaxRowSplits) rs = kMaxRowSplits;
  return rs;
}

void bn_stats_train(hipStream_t s, const float* z, int T, int H, int ld, float eps, float decay, float* mean,
                    float* rstd, float* e_mean, float* e_var, float* ws) {
  const int rs = row_splits(T), rows_per = (T + rs - 1) / rs;
  hipLaunchKernelGGL(bn_stats_partial_kernel, ct_grid(ld, rs), ct_block(), 0, s, z, T, ld, rows_per, rs, ws);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((ld + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS, FIN_KL), 0, s, ws, T, H, ld, rows_per, rs, eps,
                     decay, mean, rstd, e_mean, e_var);
}

void bn_stats_eval(hipStream_t s, const float* mov_mean, const float* mov_var, int H, float eps, float* mean,
                   float* rstd) {
  hipLaunchKernelGGL(bn_stats_eval_kernel, dim3((H + 255) / 256), dim3(256), 0, s, mov_mean, mov_var, H, eps, mean,
                     rstd);
}

void bn_act_forward(hipStream_t s, const ActDesc& d, const float* z, float* a, const float* stats, int chunk_rows,
                    int T, int H, int ld, float eps, float decay, float* mean, float* rstd, float* e_mean,
                    float* e_var, const float* beta, Twin tw, int T_apply, int slab_chunks) {
  if (T_apply < T) T_apply = T;
  const int rs = row_splits_fwd(T_apply), rows_per = (T_apply + rs - 1) / rs;
  const int nchunk = (T + chunk_rows - 1) / chunk_rows;
  hipLaunchKernelGGL(bn_act_forward_kernel, ct_grid(ld, rs), ct_block(), 0, s, d, z, a, stats, nchunk, chunk_rows, T, H,
                     ld, rows_per, eps, decay, mean, rstd, e_mean, e_var, beta, tw, T_apply,
                     slab_chunks > 0 ? slab_chunks : nchunk, bn_nt_mode(tw, T_apply, ld));
}

void act_forward(hipStream_t s, const ActDesc& d, const float* z, float* a, float* v, float* rowscale,
                 const float* mean, const float* rstd, const float* beta, int T, int H, int ld, Twin tw) {
  const int grid = T < 4096 ? T : 4096;
  if (d.l2)
    hipLaunchKernelGGL(act_forward_kernel<true>, dim3(grid), dim3(256), 0, s, d, z, a, v, rowscale, mean, rstd, beta,
                       T, H, ld, tw);
  else
    hipLaunchKernelGGL(act_forward_kernel<false>, dim3(grid), dim3(256), 0, s, d, z, a, v, rowscale, mean, rstd, beta,
                       T, H, ld, tw);
}

void act_backward_rows(hipStream_t s, const ActDesc& d, float* da, const float* v, const float* rowscale, int T,
                       int H, int ld) {
  const int grid = T < 4096 ? T : 4096;
  hipLaunchKernelGGL(act_backward_rows_kernel, dim3(grid), dim3(256), 0, s, d, da, v, rowscale, T, H, ld);
}

void hidden_backward(hipStream_t s, const ActDesc& d, int pre_du, float* da, const float* a, const float* z,
                     const float* mean, const float* rstd, int T, int H, int ld, float* ws, int stats_chunks,
                     Twin tw, int T_apply, float* ws_dz) {
  if (T_apply < T) T_apply = T;
  const int rs = row_splits(T_apply), rows_per = (T_apply + rs - 1) / rs;
  if (d.bn && stats_chunks <= 0)  // (never with T_apply > T: a stacked pass always brings the EPI_DACT partial sums)
    hipLaunchKernelGGL(hb_stats_kernel, ct_grid(ld, rs), ct_block(), 0, s, d, pre_du, da, a, z, mean, rstd, T, ld,
                       rows_per, rs, ws);
  hipLaunchKernelGGL(hb_apply_kernel, ct_grid(ld, rs), ct_block(), 0, s, d, pre_du, da, a, z, mean, rstd, T, H, ld,
                     rows_per, stats_chunks > 0 ? stats_chunks : rs, ws, tw, T_apply,
                     ws_dz ? ws_dz : ws + (size_t)2 * kMaxRowSplits * ld, bn_nt_mode(tw, T_apply, ld));
}

void bn_stats_from_chunks(hipStream_t s, const float* stats, int chunk_rows, int T, int H, int ld, float eps, float decay,
                          float* mean, float* rstd, float* e_mean, float* e_var) {
  const int nchunk = (T + chunk_rows - 1) / chunk_rows;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((ld + FIN_COLS - 1) / FIN_COLS), dim3(FIN_COLS, FIN_KL), 0, s, stats, T, H,
                     ld, chunk_rows, nchunk, eps, decay, mean, rstd, e_mean, e_var);
}

void chunk_totals(hipStream_t s, float* ws, int chunks, int ld) {
  hipLaunchKernelGGL(chunk_totals_kernel, dim3((ld + FIN_COLS - 1) / FIN_COLS, 2), dim3(FIN_COLS, FIN_KL), 0, s, ws, chunks,
                     ld);
}

void colsum_partial(hipStream_t s, const float* x, int T, int ld, float* ws) {
  const int rs = row_splits(T), rows_per = (T + rs - 1) / rs;
  hipLaunchKernelGGL(colsum_partial_kernel, ct_grid(ld, rs), ct_block(), 0, s, x, T, ld, rows_per, rs, ws);
}

void grad_final(hipStream_t s, const FinalBatch& b) {
  if (b.n <= 0) return;
  int maxn = 0;
  for (int i = 0; i < b.n; ++i) maxn = b.it[i].N > maxn ? b.it[i].N : maxn;
  hipLaunchKernelGGL(grad_final_kernel, dim3((maxn + FIN_COLS - 1) / FIN_COLS, b.n), dim3(FIN_COLS, FIN_KL), 0, s, b);
}

void softmax_xent(hipStream_t s, float* logits, const int32_t* y, int T, int O, int ld, float* row_loss,
                  int with_grad, Twin tw) {
  const int nc4 = ld / 4;
  const dim3 g(T), b(256);
#define TFK_SMX(NV) hipLaunchKernelGGL(softmax_xent_kernel<NV>, g, b, 0, s, logits, y, O, ld, row_loss, with_grad, tw)
  if (nc4 <= 256) TFK_SMX(1);
  else if (nc4 <= 512) TFK_SMX(2);
  else if (nc4 <= 1024) TFK_SMX(4);
  else if (nc4 <= 2048) TFK_SMX(8);
  else TFK_SMX(0);
#undef TFK_SMX
}

void loss_reduce(hipStream_t s, const float* row_loss, int T, float* scalars, bool overwrite, int frames, int microbatches) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, T, scalars, overwrite ? 1 : 0,
                     (float)(frames >= 0 ? frames : T), (float)microbatches);
}

void colsum_loss(hipStream_t s, const float* x, int T, int ld, float* ws, const float* row_loss, int T_loss, float* scalars,
                 bool overwrite, int frames, int microbatches) {
  const int rs = row_splits(T), rows_per = (T + rs - 1) / rs;
  const dim3 g = ct_grid(ld, rs);
  hipLaunchKernelGGL(colsum_loss_kernel, dim3(g.x, rs + 1), ct_block(), 0, s, x, T, ld, rows_per, rs, ws, row_loss, T_loss,
                     scalars, overwrite ? 1 : 0, (float)(frames >= 0 ? frames : T_loss), (float)microbatches);
}

void softmax_rows(hipStream_t s, const float* logits, int T, int O, int ld, float* out, int64_t ldo,
                  const float* prior) {
  const int nc4 = ld / 4;
  const int vec = ((ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0) ? 1 : 0;  // 16-byte stores when the rows allow them
  const dim3 g(T), b(256);
  if (nc4 <= 256) hipLaunchKernelGGL(softmax_rows_kernel<1>, g, b, 0, s, logits, O, ld, out, ldo, prior, vec);
  else if (nc4 <= 512) hipLaunchKernelGGL(softmax_rows_kernel<2>, g, b, 0, s, logits, O, ld, out, ldo, prior, vec);
  else if (nc4 <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<4>, g, b, 0, s, logits, O, ld, out, ldo, prior, vec);
  else if (nc4 <= 2048) hipLaunchKernelGGL(softmax_rows_kernel<8>, g, b, 0, s, logits, O, ld, out, ldo, prior, vec);
  else hipLaunchKernelGGL(softmax_rows_kernel<0>, g, b, 0, s, logits, O, ld, out, ldo, prior, vec);
}

void adam_apply(hipStream_t s, float* w, float* g, float* m, float* v, size_t n, const float* scalars, float lr_t,
                float beta1, float beta2, float eps, int grid_cap, uint16_t* wb, size_t n_wb, const ShadowMap* map,
                size_t first) {
  ShadowMap sm;
  if (map) sm = *map;
  else sm.n = 0;
  const size_t n4 = n / 4;
  static const int un_div = [] { const char* q = getenv("TFK_ADAM_UNROLL"); const int u = q ? atoi(q) : 2; return u == 4 ? 4 : u == 2 ? 2 : 1; }();
  size_t blocks = ((n4 + un_div - 1) / un_div + 255) / 256;
  // One trip per thread (grid = every group of 4 parameters / UN), not a grid-stride loop over a capped grid: round 2 measured,
  // at BASELINE cfg4's 152 M parameters in mixed precision, 887 us with round 1's cap of 8192 blocks, 851 / 843 / 809 at 16 k / 32 k /
  // 64 k and 742 us uncapped (5.7 TB/s); cfg2 (26 M parameters) is indifferent (105-109 us).  profiles/r02_adam_experiments.txt
  static const size_t max_blocks = [] { const char* q = getenv("TFK_ADAM_GRID"); return (size_t)(q ? atoi(q) : (1 << 22)); }();
  if (blocks > max_blocks) blocks = max_blocks;
  if (grid_cap > 0 && blocks > (size_t)grid_cap) blocks = grid_cap;
  if (blocks == 0) return;
  // streaming accesses measured 116.7 -> 107.8 us on cfg2 (6.75 TB/s); TFK_ADAM_NT=0 restores cached ones
  // (TFK_ADAM_NT=2: also the parameters; 3: also the twins the kernel writes -- tools/adam_nt_ablate.sh, profiles/r06_bn_nt.txt.
  //  Unset: 2 in mixed precision -- the fp32 masters are read and written by this kernel alone, and streamed they leave the L2s and
  //  the Infinity Cache to the bf16 shadow and the activations: the optimiser itself takes 5 us longer at cfg3, the contractions
  //  of the next step 8-15 us less, the step -1.3 % (cfg4 -1.0 %) -- and 1 elsewhere (emulated fp32 at cfg2: no difference))
  static const int forced = [] { const char* q = getenv("TFK_ADAM_NT"); return q ? atoi(q) : -1; }();
  const int nt = forced >= 0 ? forced : ((wb && sm.n == 0) ? 2 : 1);
  static const int un = [] { const char* q = getenv("TFK_ADAM_UNROLL"); return q ? atoi(q) : 2; }();
#define TFK_ADAM_LAUNCH(NTV, UNV)                                                                                  \
  hipLaunchKernelGGL((adam_kernel<NTV, UNV>), dim3((unsigned)blocks), dim3(256), 0, s, w, g, m, v, n4, scalars, lr_t, \
                     beta1, beta2, eps, wb, n_wb / 4, sm, first)
  if (nt >= 3) {
    if (un == 4) TFK_ADAM_LAUNCH(3, 4); else if (un == 2) TFK_ADAM_LAUNCH(3, 2); else TFK_ADAM_LAUNCH(3, 1);
  } else if (nt == 2) {
    if (un == 4) TFK_ADAM_LAUNCH(2, 4); else if (un == 2) TFK_ADAM_LAUNCH(2, 2); else TFK_ADAM_LAUNCH(2, 1);
  } else if (nt) {
    if (un == 4) TFK_ADAM_LAUNCH(1, 4); else if (un == 2) TFK_ADAM_LAUNCH(1, 2); else TFK_ADAM_LAUNCH(1, 1);
  } else {
    TFK_ADAM_LAUNCH(0, 1);
  }
#undef TFK_ADAM_LAUNCH
}

void step_finish(hipStream_t s, float* moving, float* e, size_t n, const float* scalars, float decay, float* host,
                 float* snap, unsigned seq) {
  const size_t blocks = n ? (n + 255) / 256 : 1;
  hipLaunchKernelGGL(step_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, moving, e, n, scalars, decay, host, snap, seq);
}
void to_bf16_rows(hipStream_t s, const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols, int x3) {
  const size_t n = (size_t)rows * (ldd / 8);
  if (n == 0) return;
  hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, lds, dst, ldd, rows,
                     cols, x3);
}
void scale_inplace(hipStream_t s, float* x, size_t n, float factor) {
  if (n == 0) return;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, factor);
}
void fill(hipStream_t s, float* x, size_t n, float value) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, value);
}
void splice_frames(hipStream_t s, const float* raw, int ldr, const int32_t* seg, int U, int T, int D, int context,
                   const float* cmvn, float* out, int ldo, const int32_t* out_seg) {
  const size_t n = (size_t)T * (ldo / 4);
  if (n == 0) return;
  hipLaunchKernelGGL(splice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, ldr, seg, U, T, D, context,
                     cmvn, out, ldo, out_seg);
}
void checksum_words(hipStream_t s, const uint32_t* p, size_t n, unsigned long long* out) {
  if (n == 0) return;
  size_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, out);
}
void dropout_mask(hipStream_t s, const ActDesc& d, float* out, int T, int H, int ld) {
  const size_t n = (size_t)T * (ld / 4);
  if (n == 0) return;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, out, T, H, ld);
}

}  // namespace tfk
