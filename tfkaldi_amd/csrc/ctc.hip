// CTC loss and its gradient w.r.t. the logits, gfx950 (wave64).  See ctc.h for what it restates.
//
//   ctc_softmax      one block per frame: softmax row (kept for the gradient) and its log-sum-exp
//   ctc_gather       log p_t(state) for the 2S+1 states of the frame's utterance, contiguous per frame
//   ctc_alpha_beta   ONE WAVE PER UTTERANCE: every lane owns R consecutive states in registers, the s-1 / s-2
//                    neighbours of a lane's first states come from the previous lane by DPP wave shifts -- the time
//                    recursion runs without LDS and without barriers.  The forward sweep (alpha, loss) and the
//                    backward sweep (beta) of an utterance are two independent waves of one launch
//   ctc_grad         one wave per frame: state posteriors from alpha, beta and log Z, folded onto the classes in
//                    LABEL ORDER (repeated labels
//                    and the S+1 blanks are summed in a fixed order, so the result is bitwise reproducible)
// Log space, fp32, with a large finite "minus infinity" so that no inf - inf can arise.
#include "ctc.h"

#include <math.h>

namespace tfk {
namespace {

constexpr float NEG = -1e30f;

// log(e^a + e^b + e^c) on the hardware exp2 / log2 units: the arguments are <= 0 and the sum lies in [1, 3], where
// v_exp_f32 / v_log_f32 are good to ~1 ulp -- the recursion is a chain of Tn dependent evaluations of this on ONE
// wave, so its instruction count is the kernel's run time
// The largest argument contributes e^0 = 1 exactly, so only the other two (v_min3 / v_med3) go through v_exp_f32:
// 2 exp + 1 log per evaluation instead of 3 + 1.
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  const float lo = fminf(a, fminf(b, c));
  const float mid = __builtin_amdgcn_fmed3f(a, b, c);
  return m + __logf(1.f + __expf(mid - m) + __expf(lo - m));
}
// value of the previous / next lane (wave64 shift by one lane as a DPP move, no LDS round trip); the lane shifted
// in at the end gets `fill`
__device__ __forceinline__ float lane_prev(float v, float fill, int lane) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  return lane == 0 ? fill : __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float lane_next(float v, float fill, int lane) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  return lane == 63 ? fill : __builtin_bit_cast(float, r);
}
// maximum over the wave, identical in every lane: quad swaps and row rotations as DPP modifiers, then the four
// 16-lane rows through v_readlane
__device__ __forceinline__ float wave_max(float x) {
#define TFK_DPP(v, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
  x = fmaxf(x, TFK_DPP(x, 0xB1));   // quad_perm [1,0,3,2]
  x = fmaxf(x, TFK_DPP(x, 0x4E));   // quad_perm [2,3,0,1]
  x = fmaxf(x, TFK_DPP(x, 0x124));  // row_ror:4
  x = fmaxf(x, TFK_DPP(x, 0x128));  // row_ror:8
#undef TFK_DPP
  const int xi = __builtin_bit_cast(int, x);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48))));
}
__device__ __forceinline__ uint16_t to_bf16(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

// utterance of frame t: largest u with seg[u] <= t
__device__ __forceinline__ int utt_of(const int32_t* __restrict__ seg, int U, int t) {
  int lo = 0, hi = U;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
ctc_softmax_kernel(const float* __restrict__ logits, int O, int ld, float* __restrict__ post, float* __restrict__ lse) {
  __shared__ float sm[8];
  const int row = blockIdx.x;
  const float* zr = logits + (size_t)row * ld;
  float* pr = post + (size_t)row * ld;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < O; c += 256) mx = fmaxf(mx, zr[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float se = 0.f;
  for (int c = threadIdx.x; c < O; c += 256) se += expf(zr[c] - mx);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  if ((threadIdx.x & 63) == 0) sm[4 + (threadIdx.x >> 6)] = se;
  __syncthreads();
  se = (sm[4] + sm[5]) + (sm[6] + sm[7]);
  const float inv = 1.f / se;
  for (int c = threadIdx.x; c < ld; c += 256) pr[c] = c < O ? expf(zr[c] - mx) * inv : 0.f;
  if (threadIdx.x == 0) lse[row] = mx + logf(se);
}

__global__ void __launch_bounds__(256)
ctc_gather_kernel(CtcBatch b) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)b.T * b.sext) return;
  const int t = (int)(idx / b.sext), s = (int)(idx % b.sext);
  const int u = utt_of(b.seg, b.U, t);
  const int l0 = b.lab_off[u], n = 2 * (b.lab_off[u + 1] - l0) + 1;
  float v = NEG;
  if (s < n) {
    const int k = (s & 1) ? b.labels[l0 + (s >> 1)] : b.O - 1;
    v = b.logits[(size_t)t * b.ld + k] - b.lse[t];
  }
  b.lp[idx] = v;
}

template <int R>
__global__ void __launch_bounds__(64)
ctc_alpha_beta_kernel(CtcBatch b) {
  // blockIdx.y = 0: the forward variables and the loss; 1 (gradient only): the backward variables.  The two sweeps
  // need nothing from each other -- only the state posteriors do, and ctc_grad forms those -- so they run as two
  // independent waves and the chain of Tn dependent steps is paid once, not twice.
  const int u = blockIdx.x, lane = threadIdx.x;
  const bool backward_sweep = blockIdx.y != 0;
  const int r0 = b.seg[u], Tn = b.seg[u + 1] - r0;
  const int l0 = b.lab_off[u], S = b.lab_off[u + 1] - l0, n = 2 * S + 1;
  if (Tn <= 0) {  // an utterance without frames: only the empty labelling is possible
    if (lane == 0 && !backward_sweep) {
      b.utt_loss[u] = S == 0 ? 0.f : INFINITY;
      b.logz[u] = 0.0;
    }
    return;
  }
  const int s0 = lane * R;
  // transitions: `skip_in[r]`  s-2 -> s allowed (s is a label state whose label differs from the previous label)
  //              `skip_out[r]` s -> s+2 allowed (the same test for state s+2)
  bool skip_in[R], skip_out[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = s0 + r, j = s >> 1;
    const bool lab = (s & 1) && s < n;
    skip_in[r] = lab && j >= 1 && b.labels[l0 + j] != b.labels[l0 + j - 1];
    skip_out[r] = lab && j + 1 < S && b.labels[l0 + j + 1] != b.labels[l0 + j];
  }
  const float* lp = b.lp + (size_t)r0 * b.sext + s0;
  // The recursion is a chain of Tn dependent steps of ~100 cycles each, while a row of lp takes ~1000 cycles to
  // arrive: PFD rows are kept in flight in a register ring.  The ring is advanced with clamped row indices instead
  // of guards (a surplus step re-does the last row with the state held), so the unrolled body has no branches and
  // the compiler's wait counts stay exact.
  // Precision: log p of a long utterance runs into the thousands, where fp32 resolves only ~1e-4.  The state
  // vector is therefore kept RELATIVE to an offset: once per PFD steps its maximum is moved into `off` (double,
  // wave-uniform; stored per frame), so the per-state values stay small and exact to ~1e-6 whatever Tn is.
  constexpr int PFD = 8;
  float pre[PFD][R];
  double off = 0.0;
  if (!backward_sweep) {
    float* al = b.ab + (size_t)r0 * b.sext + s0;
    double* offs = b.off + r0;
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = lp[r];
      a[r] = (s0 + r < 2 && s0 + r < n) ? v : NEG;
      al[r] = a[r];
    }
    if (lane == 0) offs[0] = 0.0;
#pragma unroll
    for (int j = 0; j < PFD; ++j)
#pragma unroll
      for (int r = 0; r < R; ++r) pre[j][r] = lp[(size_t)min(1 + j, Tn - 1) * b.sext + r];
    for (int t0 = 1; t0 < Tn; t0 += PFD) {
#pragma unroll
      for (int j = 0; j < PFD; ++j) {
        const int t = t0 + j;
        const bool live = t < Tn;
        const int row = min(t, Tn - 1);
        float cur[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          cur[r] = pre[j][r];
          pre[j][r] = lp[(size_t)min(t + PFD, Tn - 1) * b.sext + r];
        }
        const float up1 = lane_prev(a[R - 1], NEG, lane), up2 = lane_prev(a[R - 2], NEG, lane);
        float na[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float p1 = r >= 1 ? a[r - 1] : up1;
          const float p2 = r >= 2 ? a[r - 2] : (r == 1 ? up1 : up2);
          const float v = (s0 + r < n) ? lse3(a[r], p1, skip_in[r] ? p2 : NEG) + cur[r] : NEG;
          na[r] = live ? v : a[r];
        }
        if (j == PFD - 1) {  // compile-time: re-centre the state vector on its maximum
          float m = NEG;
#pragma unroll
          for (int r = 0; r < R; ++r) m = fmaxf(m, na[r]);
          m = wave_max(m);
          if (live && m > -1e29f) {
            off += (double)m;
#pragma unroll
            for (int r = 0; r < R; ++r) na[r] = fmaxf(na[r] - m, NEG);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          a[r] = na[r];
          al[(size_t)row * b.sext + r] = a[r];
        }
        if (lane == 0 && live) offs[t] = off;
      }
    }
    // log p(labels) = alpha_T(n-1) (+) alpha_T(n-2)
    float m = NEG;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (s0 + r == n - 1 || s0 + r == n - 2) m = fmaxf(m, a[r]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (s0 + r == n - 1 || s0 + r == n - 2) se += expf(a[r] - m);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const float log_z_rel = m + logf(se);  // relative to the final offset
    const bool feasible = log_z_rel > -1e29f;
    const double log_z = off + (double)log_z_rel;
    if (lane == 0) {
      b.utt_loss[u] = feasible ? (float)-log_z : INFINITY;
      b.logz[u] = log_z;
    }
    return;
  }
  // backward variables beta~_t(s) (emission of frame t included), relative to their own offsets
  float* be = b.bb + (size_t)r0 * b.sext + s0;
  double* offs = b.offb + r0;
  float bt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int s = s0 + r;
    bt[r] = (s == n - 1 || s == n - 2) ? lp[(size_t)(Tn - 1) * b.sext + r] : NEG;
    be[(size_t)(Tn - 1) * b.sext + r] = bt[r];
  }
  if (lane == 0) offs[Tn - 1] = 0.0;
#pragma unroll
  for (int j = 0; j < PFD; ++j)
#pragma unroll
    for (int r = 0; r < R; ++r) pre[j][r] = lp[(size_t)max(Tn - 2 - j, 0) * b.sext + r];
  for (int t0 = Tn - 2; t0 >= 0; t0 -= PFD) {
#pragma unroll
    for (int j = 0; j < PFD; ++j) {
      const int t = t0 - j;
      const bool live = t >= 0;
      const int row = max(t, 0);
      float cur[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        cur[r] = pre[j][r];
        pre[j][r] = lp[(size_t)max(t - PFD, 0) * b.sext + r];
      }
      const float dn1 = lane_next(bt[0], NEG, lane), dn2 = lane_next(bt[1], NEG, lane);
      float nb[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float q1 = r + 1 < R ? bt[r + 1] : dn1;
        const float q2 = r + 2 < R ? bt[r + 2] : (r + 1 < R ? dn1 : dn2);
        const float v = (s0 + r < n) ? lse3(bt[r], q1, skip_out[r] ? q2 : NEG) + cur[r] : NEG;
        nb[r] = live ? v : bt[r];
      }
      if (j == PFD - 1) {  // re-centre beta
        float m = NEG;
#pragma unroll
        for (int r = 0; r < R; ++r) m = fmaxf(m, nb[r]);
        m = wave_max(m);
        if (live && m > -1e29f) {
          off += (double)m;
#pragma unroll
          for (int r = 0; r < R; ++r) nb[r] = fmaxf(nb[r] - m, NEG);
        }
      }
      // (a surplus step holds the state and rewrites row 0 with the value it already has)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        bt[r] = nb[r];
        be[(size_t)row * b.sext + r] = bt[r];
      }
      if (lane == 0 && live) offs[t] = off;
    }
  }
}

// one wave per frame; dynamic LDS: the class row, then the frame's state posteriors
//   gamma_t(s) = exp((alpha~ + beta~ - lp) + (off_alpha(t) + off_beta(t) - log Z))
// (the offsets cancel to a small number, which is exact enough in fp32)
__global__ void __launch_bounds__(64)
ctc_grad_kernel(CtcBatch b, float* __restrict__ dlogits, Twin tw) {
  extern __shared__ float row[];
  float* g = row + b.ld;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int u = utt_of(b.seg, b.U, t);
  const int l0 = b.lab_off[u], S = b.lab_off[u + 1] - l0, n = 2 * S + 1;
  const bool live = b.utt_loss[u] < INFINITY;
  const float* pr = b.post + (size_t)t * b.ld;
  for (int c = lane; c < b.ld; c += 64) row[c] = live ? pr[c] : 0.f;
  {
    const float shift = live ? (float)(b.off[t] + b.offb[t] - b.logz[u]) : 0.f;
    const float* al = b.ab + (size_t)t * b.sext;
    const float* be = b.bb + (size_t)t * b.sext;
    const float* lp = b.lp + (size_t)t * b.sext;
    for (int s = lane; s < n; s += 64) g[s] = live ? __expf(al[s] + be[s] - lp[s] + shift) : 0.f;
  }
  __syncthreads();
  float blank = 0.f;
  for (int s = 2 * lane; s < n; s += 128) blank += g[s];
  for (int o = 32; o > 0; o >>= 1) blank += __shfl_xor(blank, o);
  if (lane == 0 && live) {
    row[b.O - 1] -= blank;
    for (int j = 0; j < S; ++j) row[b.labels[l0 + j]] -= g[2 * j + 1];  // label order: deterministic for repeats
  }
  __syncthreads();
  float* dr = dlogits + (size_t)t * b.ld;
  for (int c = lane; c < b.ld; c += 64) dr[c] = row[c];
  if (tw.p)
    for (int c = lane; c < b.ld; c += 64) twin_put(tw, (size_t)t, c, row[c]);
}

__global__ void __launch_bounds__(256)
ctc_loss_reduce_kernel(const float* __restrict__ utt_loss, const int32_t* __restrict__ lab_off, int U,
                       float* __restrict__ scalars, int overwrite) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < U; i += 256) s += utt_loss[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    const float labels = (float)(lab_off[U] - lab_off[0]);
    scalars[0] = overwrite ? s : scalars[0] + s;
    scalars[1] = overwrite ? labels : scalars[1] + labels;
    scalars[2] = overwrite ? 1.f : scalars[2] + 1.f;
  }
}

int regs_for(int max_labels) {
  const int n = 2 * max_labels + 1;
  int r = 2;
  while (64 * r < n) r *= 2;
  return r;
}

}  // namespace

int ctc_state_stride(int max_labels) { return 64 * regs_for(max_labels); }

void ctc_loss_grad(hipStream_t s, const CtcBatch& b, float* dlogits, int with_grad, Twin tw) {
  if (b.T <= 0 || b.U <= 0) return;
  hipLaunchKernelGGL(ctc_softmax_kernel, dim3(b.T), dim3(256), 0, s, b.logits, b.O, b.ld, b.post, b.lse);
  const size_t n = (size_t)b.T * b.sext;
  hipLaunchKernelGGL(ctc_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b);
  const dim3 grid(b.U, with_grad ? 2 : 1);
  switch (b.sext / 64) {
    case 2: hipLaunchKernelGGL(ctc_alpha_beta_kernel<2>, grid, dim3(64), 0, s, b); break;
    case 4: hipLaunchKernelGGL(ctc_alpha_beta_kernel<4>, grid, dim3(64), 0, s, b); break;
    case 8: hipLaunchKernelGGL(ctc_alpha_beta_kernel<8>, grid, dim3(64), 0, s, b); break;
    default: hipLaunchKernelGGL(ctc_alpha_beta_kernel<16>, grid, dim3(64), 0, s, b); break;
  }
  if (with_grad)
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(b.T), dim3(64), (size_t)(b.ld + b.sext) * sizeof(float), s, b, dlogits, tw);
}

void ctc_loss_reduce(hipStream_t s, const float* utt_loss, const int32_t* lab_off, int U, float* scalars,
                     bool overwrite) {
  hipLaunchKernelGGL(ctc_loss_reduce_kernel, dim3(1), dim3(256), 0, s, utt_loss, lab_off, U, scalars, overwrite ? 1 : 0);
}

}  // namespace tfk
