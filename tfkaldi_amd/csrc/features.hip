// Feature computation on the device (gfx950): wav samples -> pre-emphasis -> frames -> power spectrum -> mel
// filterbank -> log / DCT + lifter / spectral centroids -> deltas, and the per-speaker CMVN sums.
//
// Reference semantics (vrenkens/tfkaldi, all float64 numpy):
//   processing/feat.py:42-69        FeatureComputer.__call__: features, optional log-energy column, dynamics
//   processing/base.py:39-154       mfcc / fbank / logfbank / ssc
//   processing/base.py:226-284      lifter, deriv (scipy.ndimage.convolve1d [2,1,0,-1,-2], 'reflect'), delta, ddelta
//   processing/sigproc.py:33-191    framesig (rectangular window, zero padding), magspec, powspec, preemphasis
//   processing/prepare_data.py:80-118  compute_cmvn (float32 row-after-row sums)
//
// Shape of the work: a frame is 400 samples in, 40 numbers out, ~16 kFLOP of float64 in between -- neither an HBM nor an
// MFMA problem; what bounds it is instruction issue (vector ALU busy 0.67) and the LDS traffic of the transform (0.51).
// One WAVEFRONT owns one frame: the 512-point real transform is a 256-point complex Stockham radix-4 transform over
// (even, odd) sample pairs, ping-ponging between two 4 KB buffers in that wave's slice of LDS, untangled into the half
// spectrum on the way to the power; the mel filterbank runs over the non-zero supports of its triangles, cut into equal
// pieces across the lanes; no intermediate leaves the CU except the static features of utterances that need deltas
// (float64, read back by the dynamics kernel with the 'reflect' boundary applied per utterance).  The frames of a whole
// batch of utterances go in one launch whose waves walk over them.  Measurements and history: DESIGN.md 4b.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/tfkaldi_hip.h"

namespace tfk {
int set_error(int code, const char* msg);  // engine.hip (thread-local message behind tfk_last_error)
}

namespace {

constexpr double kEps = 2.220446049250313e-16;  // numpy.finfo(float).eps: base.py:84,94
constexpr int kMaxFft = 4096;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return tfk::set_error(code ? code : -1, buf);
}

#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct Batch {
  const void* sig;
  const int64_t* sig_off;
  const int64_t* frame_off;
  int n_utts;
  int64_t n_frames;
};

// Where a frame starts, resolved once per batch by frame_meta_kernel: the transform kernel then begins with ONE
// 16-byte load instead of an utterance search and three dependent offset loads.
struct FrameMeta {
  int64_t start;   // index of the frame's first sample in the concatenated signal
  int32_t avail;   // samples of its utterance from there on (<= 0: none), clamped to int32
  int32_t first;   // the frame starts at sample 0 of its utterance (y[0] = x[0], sigproc.py:191)
};

// The mel filterbank is triangles: filter j is non-zero on the bins [lo[j], lo[j] + cnt[j]) only, and its weights
// there sit at val[off[j] ...] -- ~2 values per bin instead of nfilt.
struct FrameArgs {
  Batch b;
  int frame_len, frame_step, nfft, log2_n2, nfilt, ncep, kind, include_energy, stage;
  double preemph, inv_nfft;
  const int* fb_meta;    // [3][nfilt]: lo | cnt | off, then (n_items > 0) [4][64] items: filter | lo | cnt | off, [nfilt+1] first item
  const double* fb_val;  // [fb_nnz]
  int fb_nnz;
  int n_items;           // > 0: the supports are cut into <= 64 pieces of near-equal length, one per lane (see create)
  int n_meta;            // ints in fb_meta
  int dct_lds;           // the DCT matrix is small enough to be staged in LDS
  const double* binw;    // [nbins]
  const double* dct;     // [nfilt][ncep]
  const double* lift;    // [ncep]
  const double2* tw;     // e^{-2 pi i k / nfft}, k < nfft/2
  const FrameMeta* fmeta;  // [n_frames]
  void* out;
  int64_t ld_out;
  int out_f64;
};

// last utterance whose first frame (row) is <= f
__device__ __forceinline__ int find_utt(const int64_t* off, int n, int64_t f) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= f) lo = mid; else hi = mid;
  }
  return lo;
}

template <int SAMPLE>
__device__ __forceinline__ double raw_sample(const void* sig, int64_t i) {
  if (SAMPLE == TFK_SAMPLE_I16) return (double)((const int16_t*)sig)[i];
  if (SAMPLE == TFK_SAMPLE_F32) return (double)((const float*)sig)[i];
  return ((const double*)sig)[i];
}

// y = x - coeff * x_prev with numpy's roundings: in float64, except for float32 signals, where numpy casts the Python
// float down and both operations round to float32
template <int SAMPLE>
__device__ __forceinline__ double emphasise(double x, double xp, double coeff) {
#pragma clang fp contract(off)
  if (SAMPLE == TFK_SAMPLE_F32) {
    const float c = (float)coeff, scaled = c * (float)xp;
    return (double)((float)x - scaled);
  }
  const double scaled = coeff * xp;
  return x - scaled;
}

// pre-emphasised sample i of the utterance at [base, base + len): sigproc.py:180-191 -- two roundings, as numpy's
// `signal[1:] - coeff * signal[:-1]` has; zero past the end (the padding of sigproc.py:57-60 follows the filter)
template <int SAMPLE>
__device__ __forceinline__ double emph_sample(const void* sig, int64_t base, int64_t len, int64_t i, double coeff) {
#pragma clang fp contract(off)  // HIP's __dmul_rn / __dsub_rn are plain operators: without this the pair fuses into one FMA
  if (i >= len) return 0.0;
  const double x = raw_sample<SAMPLE>(sig, base + i);
  if (i == 0 || coeff == 0.0) return x;
  return emphasise<SAMPLE>(x, raw_sample<SAMPLE>(sig, base + i - 1), coeff);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ void put(void* out, int64_t idx, double v) { ((T*)out)[idx] = (T)v; }

// LDS of one frame belongs to one wavefront: its DS instructions execute in order, so a wave-level fence (for the
// compiler) is all that separates a transform stage from the next
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- sigproc.framesig of the pre-emphasised signal: out[f][n], n < frame_len ----
template <int SAMPLE>
__global__ __launch_bounds__(256) void frames_kernel(FrameArgs p) {
  const int64_t f = blockIdx.x;
  const int u = find_utt(p.b.frame_off, p.b.n_utts, f);
  const int64_t t = f - p.b.frame_off[u];
  const int64_t base = p.b.sig_off[u], len = p.b.sig_off[u + 1] - base;
  double* out = (double*)p.out + f * p.ld_out;
  for (int n = threadIdx.x; n < p.frame_len; n += blockDim.x)
    out[n] = emph_sample<SAMPLE>(p.b.sig, base, len, t * p.frame_step + n, p.preemph);
}

// ---- one wavefront per frame: samples -> spectrum -> features ----
struct alignas(16) Cplx { double x, y; };  // 16-byte LDS accesses: a wave's 64 points sweep all banks once
__device__ __forceinline__ Cplx cmul(Cplx a, double2 w) { return Cplx{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }

// (even, odd) pre-emphasised sample pair m of a frame: one complex point of the half-length transform.
// Branch-free: the three samples x[2m-1], x[2m], x[2m+1] are loaded unconditionally from clamped (always valid)
// addresses and the edge rules are applied with selects -- with a branch per rule the compiler waits for every load
// before it issues the next one, and a frame's 16 loads cost 16 round trips to the L2.
// Requires fm.avail >= 1 (the caller skips frames of empty utterances: all their points are zero).
template <int SAMPLE> struct RawOf { typedef int T; };        // int16 samples travel as sign-extended 32-bit registers
template <> struct RawOf<TFK_SAMPLE_F64> { typedef double T; };
template <> struct RawOf<TFK_SAMPLE_F32> { typedef float T; };

template <int SAMPLE>
__device__ __forceinline__ typename RawOf<SAMPLE>::T raw_load(const void* sig, int64_t i) {
  if (SAMPLE == TFK_SAMPLE_I16) return (typename RawOf<SAMPLE>::T)((const int16_t*)sig)[i];
  if (SAMPLE == TFK_SAMPLE_F32) return (typename RawOf<SAMPLE>::T)((const float*)sig)[i];
  return (typename RawOf<SAMPLE>::T)((const double*)sig)[i];
}

// the three loads of pair m (addresses only -- nothing here waits for memory)
template <int SAMPLE>
__device__ __forceinline__ void load_pair(const FrameArgs& p, const FrameMeta& fm, int m, typename RawOf<SAMPLE>::T& xa,
                                          typename RawOf<SAMPLE>::T& xb, typename RawOf<SAMPLE>::T& xc) {
  const int n0 = 2 * m, n1 = n0 + 1, last = fm.avail - 1;
  const bool head = n0 == 0 && fm.first;
  const int ia = head ? 0 : min(n0 - 1, last);            // (a frame that does not start its utterance has a sample before it)
  xa = raw_load<SAMPLE>(p.b.sig, fm.start + ia);
  xb = raw_load<SAMPLE>(p.b.sig, fm.start + min(n0, last));
  xc = raw_load<SAMPLE>(p.b.sig, fm.start + min(n1, last));
}

// pre-emphasis and the edge rules on the loaded samples
template <int SAMPLE>
__device__ __forceinline__ Cplx finish_pair(const FrameArgs& p, const FrameMeta& fm, int used, int m,
                                            typename RawOf<SAMPLE>::T ra, typename RawOf<SAMPLE>::T rb,
                                            typename RawOf<SAMPLE>::T rc) {
#pragma clang fp contract(off)  // y = x - coeff * x_prev: two roundings, as numpy's `signal[1:] - coeff * signal[:-1]`
  const int n0 = 2 * m, n1 = n0 + 1;
  const int lim = min(used, fm.avail);
  const bool head = n0 == 0 && fm.first;                  // y[0] = x[0] (sigproc.py:191)
  const double xa = (double)ra, xb = (double)rb, xc = (double)rc;
  const double y0 = emphasise<SAMPLE>(xb, xa, p.preemph), y1 = emphasise<SAMPLE>(xc, xb, p.preemph);
  const bool plain = p.preemph == 0.0;
  Cplx v;
  v.x = n0 < lim ? ((head || plain) ? xb : y0) : 0.0;
  v.y = n1 < lim ? (plain ? xc : y1) : 0.0;
  return v;
}

template <int SAMPLE>
__device__ __forceinline__ Cplx sample_pair(const FrameArgs& p, const FrameMeta& fm, int used, int m) {
  typename RawOf<SAMPLE>::T xa, xb, xc;
  load_pair<SAMPLE>(p, fm, m, xa, xb, xc);
  return finish_pair<SAMPLE>(p, fm, used, m, xa, xb, xc);
}

// radix-4 butterfly on already-twiddled inputs: outputs r = 0..3 of the 4-point transform
__device__ __forceinline__ void bfly4(const Cplx& v0, const Cplx& v1, const Cplx& v2, const Cplx& v3, Cplx* o) {
  const Cplx a{v0.x + v2.x, v0.y + v2.y}, b{v0.x - v2.x, v0.y - v2.y};
  const Cplx c{v1.x + v3.x, v1.y + v3.y}, d{v1.y - v3.y, -(v1.x - v3.x)};  // d = -i (v1 - v3)
  o[0] = Cplx{a.x + c.x, a.y + c.y};
  o[1] = Cplx{b.x + d.x, b.y + d.y};
  o[2] = Cplx{a.x - c.x, a.y - c.y};
  o[3] = Cplx{b.x - d.x, b.y - d.y};
}

__global__ __launch_bounds__(256) void frame_meta_kernel(Batch b, int frame_step, FrameMeta* out) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= b.n_frames) return;
  const int u = find_utt(b.frame_off, b.n_utts, f);
  const int64_t t = f - b.frame_off[u];
  const int64_t base = b.sig_off[u], len = b.sig_off[u + 1] - base, s0 = t * frame_step;
  const int64_t avail = len - s0;
  out[f] = FrameMeta{base + s0, (int32_t)(avail > 0x7fffffff ? 0x7fffffff : (avail < 0 ? 0 : avail)), t == 0 ? 1 : 0};
}

// twiddle e^{-2 pi i m / nfft} for m < nfft (the table holds the first half turn)
__device__ __forceinline__ double2 twiddle(const double2* tws, int N2, int m) {
  if (m < N2) return tws[m];
  const double2 w = tws[m - N2];
  return double2{-w.x, -w.y};
}

// LOG2N2 > 0: log2 of the half-length known at compile time (8 = the 512-point transform of the reference's
// configurations) -- pass count, strides and twiddle steps fold to constants and the passes unroll; 0: any size.
template <int SAMPLE, int LOG2N2>
__global__ __launch_bounds__(512) void feat_frames_kernel(FrameArgs p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // (the wave index through readfirstlane: the compiler then knows that everything per frame -- frame number, its
  // metadata, buffer addresses -- is uniform and keeps it in scalar registers and scalar instructions)
  const int waves = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int N2 = LOG2N2 ? (1 << LOG2N2) : (p.nfft >> 1), nbins = N2 + 1;
  const int log2_n2 = LOG2N2 ? LOG2N2 : p.log2_n2;
  // block-shared tables, then two [N2] complex buffers per wave
  double2* tws = (double2*)lds;
  double* fbv = lds + 2 * N2;
  double* binw = fbv + p.fb_nnz;
  double* dcts = binw + (p.kind == TFK_FEAT_SSC ? nbins : 0);
  int* meta = (int*)(dcts + (p.dct_lds ? p.nfilt * p.ncep : 0));
  const size_t table_bytes = ((size_t)((double*)meta - lds) * sizeof(double) + (size_t)p.n_meta * sizeof(int) + 15) & ~(size_t)15;
  double* mine = lds + table_bytes / sizeof(double) + (size_t)wave * 4 * N2;
  Cplx* bufA = (Cplx*)mine;
  Cplx* bufB = bufA + N2;
  for (int k = threadIdx.x; k < N2; k += blockDim.x) tws[k] = p.tw[k];
  for (int k = threadIdx.x; k < p.fb_nnz; k += blockDim.x) fbv[k] = p.fb_val[k];
  for (int k = threadIdx.x; k < p.n_meta; k += blockDim.x) meta[k] = p.fb_meta[k];
  if (p.kind == TFK_FEAT_SSC)
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) binw[k] = p.binw[k];
  if (p.dct_lds)
    for (int k = threadIdx.x; k < p.nfilt * p.ncep; k += blockDim.x) dcts[k] = p.dct[k];
  __syncthreads();
  // The grid is sized to what the chip holds at once (launch_frames); every wave then walks over its share of the
  // frames, so the tables above are staged once per block and not once per 8 frames.  No block-wide barrier below.
  const int used = min(p.frame_len, p.nfft);             // numpy.fft.rfft(frames, nfft) truncates / zero-pads
  const int64_t stride = (int64_t)gridDim.x * waves;
  for (int64_t f = (int64_t)blockIdx.x * waves + wave; f < p.b.n_frames; f += stride) {
  const FrameMeta fm = p.fmeta[f];

  // Half-length complex transform, Stockham autosort (natural order in and out, ping-pong between the two buffers),
  // radix 4 with one leading radix-2 pass when log2(N2) is odd.  The first pass has unit twiddles and takes its
  // points straight from global memory.
  Cplx* src = bufA;
  Cplx* dst = bufB;
  int Ns;
  const bool silent = fm.avail <= 0;                       // a frame of an empty utterance: nothing to read
  const Cplx zero{0.0, 0.0};
  if (log2_n2 & 1) {
    for (int j = lane; j < (N2 >> 1); j += 64) {
      Cplx a = zero, b = zero;
      if (!silent) {
        a = sample_pair<SAMPLE>(p, fm, used, j);
        b = sample_pair<SAMPLE>(p, fm, used, j + (N2 >> 1));
      }
      dst[2 * j] = Cplx{a.x + b.x, a.y + b.y};
      dst[2 * j + 1] = Cplx{a.x - b.x, a.y - b.y};
    }
    Ns = 2;
  } else {
    const int q = N2 >> 2;
    for (int j = lane; j < q; j += 64) {
      Cplx v0 = zero, v1 = zero, v2 = zero, v3 = zero;
      if (!silent) {                                       // one uniform branch: the twelve loads go out back to back
        v0 = sample_pair<SAMPLE>(p, fm, used, j);
        v1 = sample_pair<SAMPLE>(p, fm, used, j + q);
        v2 = sample_pair<SAMPLE>(p, fm, used, j + 2 * q);
        v3 = sample_pair<SAMPLE>(p, fm, used, j + 3 * q);
      }
      Cplx o[4];
      bfly4(v0, v1, v2, v3, o);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[4 * j + r] = o[r];
    }
    Ns = 4;
  }
  wave_sync();
  { Cplx* tmp = src; src = dst; dst = tmp; }
  for (; Ns < N2; Ns <<= 2) {
    const int q = N2 >> 2;
    const int tstep = N2 / (2 * Ns);                      // w = e^{-2 pi i k / (4 Ns)} = table[k * N2 / (2 Ns)]
    for (int j = lane; j < q; j += 64) {
      const int k = j & (Ns - 1);
      const int m1 = k * tstep;
      // (twiddles of a pass are strided table reads: many lanes on few LDS banks -- they come through the L1 instead,
      // where the 4 KB table stays resident; the loads do not depend on the data and are issued ahead of it)
      const double2 w1 = p.tw[m1], w2 = p.tw[2 * m1], w3 = twiddle(p.tw, N2, 3 * m1);
      Cplx o[4];
      bfly4(src[j], cmul(src[j + q], w1), cmul(src[j + 2 * q], w2), cmul(src[j + 3 * q], w3), o);
      const int at = ((j - k) << 2) + k;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[at + r * Ns] = o[r];
    }
    wave_sync();
    Cplx* tmp = src; src = dst; dst = tmp;
  }
  // Untangle the real spectrum: with E = (Z[k] + conj Z[N2-k]) / 2, O = -i (Z[k] - conj Z[N2-k]) / 2 and
  // T = W^k O:  X[k] = E + T,  X[N2-k] = conj(E - T).  One lane owns the pair (k, N2-k); the spectrum goes into the
  // buffer the transform left free.
  double* pw = (double*)dst;                              // [N2 + 1]
  double esum = 0.0;
  auto emit = [&](int k, double xr, double xi) {
    const double sq = fma(xr, xr, xi * xi);               // |X|^2 (sigproc.py:139,153 take the square of the modulus)
    const double pk = p.inv_nfft * sq;
    pw[k] = p.stage == TFK_STAGE_MAGSPEC ? sqrt(sq) : pk;
    esum += pk;
  };
  // (the three bins without a distinct partner -- 0, N2 and the middle one, where W^k O = -i Im Z -- are lane 0's extra
  // work in the first trip, so that the trips cover k = 1 .. N2/2 - 1 only: two instead of three at N2 = 256)
  for (int k = lane; k < (N2 >> 1); k += 64) {
    if (k == 0) {
      const Cplx z = src[0], mid = src[N2 >> 1];
      emit(0, z.x + z.y, 0.0);
      emit(N2, z.x - z.y, 0.0);
      emit(N2 >> 1, mid.x, -mid.y);
    } else {
      const Cplx a = src[k], bq = src[N2 - k];
      const double br = bq.x, bi = -bq.y;
      const double er = 0.5 * (a.x + br), ei = 0.5 * (a.y + bi);
      const Cplx o{0.5 * (a.y - bi), -0.5 * (a.x - br)};
      const Cplx tt = cmul(o, tws[k]);
      emit(k, er + tt.x, ei + tt.y);
      emit(N2 - k, er - tt.x, ei - tt.y);
    }
  }
  wave_sync();
  if (p.stage) {
    double* out = (double*)p.out + f * p.ld_out;
    for (int k = lane; k <= N2; k += 64) out[k] = pw[k];
    wave_sync();
    continue;
  }
  double energy = wave_sum(esum);                         // base.py:80-84
  if (energy == 0.0) energy = kEps;

  // mel filterbank (base.py:90).  A filter's support runs from 4 bins (low mels) to ~45 (high mels): with the filters
  // on the lanes the loop is as long as the widest and most lanes idle.  The supports are therefore cut into <= 64
  // pieces of near-equal length (create()), one per lane; the pieces of a filter are added up in order afterwards.
  double* logf = (double*)src;                            // the transform's other buffer is free now
  double* part = logf + p.nfilt;                          // [2][64] partial sums of the pieces
  const int d_static = (p.kind == TFK_FEAT_MFCC ? p.ncep : p.nfilt);
  const int64_t row = f * p.ld_out;
  if (p.n_items > 0) {
    const int* item = meta + 3 * p.nfilt;
    if (lane < p.n_items) {
      const int lo = item[64 + lane], cnt = item[128 + lane];
      const double* w = fbv + item[192 + lane];
      double acc = 0.0, num = 0.0;
      if (p.kind == TFK_FEAT_SSC) {
        for (int i = 0; i < cnt; ++i) {
          const double pk = pw[lo + i];
          acc = fma(pk, w[i], acc);
          num = fma(pk * binw[lo + i], w[i], num);         // numpy.dot(pspec * tiles, filterbank.T) (base.py:154)
        }
        part[64 + lane] = num;
      } else {
        for (int i = 0; i < cnt; ++i) acc = fma(pw[lo + i], w[i], acc);
      }
      part[lane] = acc;
    }
    wave_sync();
  }
  for (int j = lane; j < p.nfilt; j += 64) {
    double acc = 0.0, num = 0.0;
    if (p.n_items > 0) {
      const int* first = meta + 3 * p.nfilt + 256;
      for (int it = first[j]; it < first[j + 1]; ++it) {
        acc += part[it];
        if (p.kind == TFK_FEAT_SSC) num += part[64 + it];
      }
    } else {
      const int lo = meta[j], cnt = meta[p.nfilt + j];
      const double* w = fbv + meta[2 * p.nfilt + j];
      if (p.kind == TFK_FEAT_SSC) {
        for (int i = 0; i < cnt; ++i) {
          const double pk = pw[lo + i];
          acc = fma(pk, w[i], acc);
          num = fma(pk * binw[lo + i], w[i], num);
        }
      } else {
        for (int i = 0; i < cnt; ++i) acc = fma(pw[lo + i], w[i], acc);
      }
    }
    double v;
    if (p.kind == TFK_FEAT_SSC) {
      v = num / acc;                                       // base.py:154: the denominator is not guarded there
    } else {
      if (acc == 0.0) acc = kEps;                          // base.py:93-94
      v = p.kind == TFK_FEAT_FBANK_RAW ? acc : log(acc);
    }
    if (p.kind == TFK_FEAT_MFCC) logf[j] = v;
    else if (p.out_f64) put<double>(p.out, row + j, v);
    else put<float>(p.out, row + j, v);
  }
  if (p.kind == TFK_FEAT_MFCC) {
    wave_sync();
    const double* dct = p.dct_lds ? dcts : p.dct;
    for (int c = lane; c < p.ncep; c += 64) {
      double acc = 0.0;
#pragma unroll 8
      for (int j = 0; j < p.nfilt; ++j) acc = fma(logf[j], dct[(size_t)j * p.ncep + c], acc);
      const double v = p.lift[c] * acc;                    // base.py:56,243
      if (p.out_f64) put<double>(p.out, row + c, v); else put<float>(p.out, row + c, v);
    }
  }
  if (p.include_energy && lane == 0) {                     // feat.py:61-62: log-energy is the last static column
    const double v = p.kind == TFK_FEAT_FBANK_RAW ? energy : log(energy);
    if (p.out_f64) put<double>(p.out, row + d_static, v); else put<float>(p.out, row + d_static, v);
  }
  wave_sync();                                             // the next frame reuses both buffers
  }  // frames of this wave
}

// ---- base.deriv / delta / ddelta with scipy's 'reflect' boundary per utterance ----
struct DynArgs {
  const double* x;
  int64_t ld_x;
  int dim;
  const int64_t* row_off;
  int n_utts;
  int64_t n_rows;
  int dynamic, deriv_only;
  void* out;
  int64_t ld_out;
  int out_f64;
};

__device__ __forceinline__ int64_t reflect(int64_t i, int64_t n) {  // d c b a | a b c d | d c b a
  const int64_t period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - 1 - i : i;
}

// scipy's correlate1d on the anti-symmetric kernel: tmp = w0 x[t]; tmp += w[-2] (x[t-2] - x[t+2]); tmp += w[-1] (x[t-1] - x[t+1])
__device__ __forceinline__ double deriv5(double c, double m2, double m1, double p1, double p2) {
#pragma clang fp contract(off)
  double acc = 0.0 * c;
  acc = acc + -2.0 * (m2 - p2);
  acc = acc + -1.0 * (m1 - p1);
  return acc;
}

__global__ __launch_bounds__(256) void dynamic_kernel(DynArgs p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.n_rows * p.dim) return;
  const int64_t f = idx / p.dim;
  const int c = (int)(idx - f * p.dim);
  const int u = find_utt(p.row_off, p.n_utts, f);
  const int64_t r0 = p.row_off[u], n = p.row_off[u + 1] - r0, t = f - r0;
  auto X = [&](int64_t tt) { return p.x[(r0 + reflect(tt, n)) * p.ld_x + c]; };
  auto D1 = [&](int64_t tt) {
    const int64_t q = reflect(tt, n);                      // the first derivative AT a (reflected) frame
    return deriv5(X(q), X(q - 2), X(q - 1), X(q + 1), X(q + 2));
  };
  const double x0 = X(t), d1 = D1(t);
  const int64_t row = f * p.ld_out;
  auto store = [&](int64_t col, double v) {
    if (p.out_f64) put<double>(p.out, row + col, v); else put<float>(p.out, row + col, v);
  };
  if (p.deriv_only) { store(c, d1); return; }
  store(c, x0);
  if (p.dynamic >= 1) store(p.dim + c, d1);
  if (p.dynamic >= 2) {
    if (t >= 2 && t + 2 < n) {
      // the five first derivatives around an interior frame come from ONE window x[t-4 .. t+4] (each index reflected on
      // its own): 9 loads instead of 25
      double w[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) w[i] = X(t - 4 + i);
      const double m2 = deriv5(w[2], w[0], w[1], w[3], w[4]), m1 = deriv5(w[3], w[1], w[2], w[4], w[5]);
      const double p1 = deriv5(w[5], w[3], w[4], w[6], w[7]), p2 = deriv5(w[6], w[4], w[5], w[7], w[8]);
      store(2 * (int64_t)p.dim + c, deriv5(d1, m2, m1, p1, p2));
    } else {
      store(2 * (int64_t)p.dim + c, deriv5(d1, D1(t - 2), D1(t - 1), D1(t + 1), D1(t + 2)));
    }
  }
}

// ---- compute_cmvn: float32 sums accumulated row after row (numpy's axis-0 reduction of a C-contiguous matrix) ----
__global__ __launch_bounds__(64) void cmvn_stats_kernel(const float* __restrict__ feats, int64_t ld, int dim,
                                                        const int64_t* __restrict__ spk_off,
                                                        const int64_t* __restrict__ utt_row,
                                                        const int64_t* __restrict__ utt_len, double* __restrict__ stats) {
#pragma clang fp contract(off)  // numpy squares in float32 and then adds: two roundings
  const int s = blockIdx.x;
  const int c = blockIdx.y * 64 + threadIdx.x;
  double* out = stats + (size_t)s * 2 * (dim + 1);
  int64_t count = 0;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t q = spk_off[s]; q < spk_off[s + 1]; ++q) {
    const int64_t n = utt_len[q];
    count += n;
    if (c >= dim) continue;
    const float* x = feats + utt_row[q] * ld + c;
    int64_t r = 0;
    for (; r + 8 <= n; r += 8) {                           // loads are independent of the (serial) adds: batch them
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = x[(r + i) * ld];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float sq = v[i] * v[i];                        // numpy.square in float32, then the float32 sum
        s1 = s1 + v[i];
        s2 = s2 + sq;
      }
    }
    for (; r < n; ++r) {
      const float v = x[r * ld];
      const float sq = v * v;
      s1 = s1 + v;
      s2 = s2 + sq;
    }
  }
  if (c < dim) {
    out[c] = (double)s1;
    out[dim + 1 + c] = (double)s2;
  } else if (c == dim) {
    out[dim] = (double)count;                              // prepare_data.py:111
    out[2 * dim + 1] = 0.0;
  }
}

// ---- sigproc.deframesig: thread per output sample, the covering frames added in frame order ----
__global__ __launch_bounds__(256) void deframe_kernel(const double* __restrict__ frames, int64_t ld, int64_t n_frames,
                                                      int frame_len, int frame_step, const double* __restrict__ win,
                                                      int64_t padlen, double* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= padlen) return;
  int64_t lo = s - frame_len + 1;
  lo = lo <= 0 ? 0 : (lo + frame_step - 1) / frame_step;   // first frame that reaches sample s
  int64_t hi = s / frame_step;                              // last frame that starts at or before it
  if (hi > n_frames - 1) hi = n_frames - 1;
  double acc = 0.0, corr = 0.0;
  for (int64_t i = lo; i <= hi; ++i) {
    const int j = (int)(s - i * frame_step);
    acc = acc + frames[i * ld + j];
    corr = (corr + (win ? win[j] : 1.0)) + 1e-15;           // sigproc.py:113-115
  }
  out[s] = acc / corr;                                      // (a sample no frame covers is 0/0 there as well)
}

// ---- sigproc.logpowspec's tail: 10 log10(max(p, 1e-30)) [- global maximum] ----
__global__ __launch_bounds__(256) void logpow_kernel(double* __restrict__ p, int64_t n, double* __restrict__ block_max) {
  __shared__ double red[256];
  double m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = p[i];
    if (v <= 1e-30) v = 1e-30;                              // sigproc.py:172
    v = 10.0 * log10(v);
    p[i] = v;
    m = v > m ? v : m;                                      // (NaN never wins, as numpy.max would have it propagate: see below)
    if (v != v) m = v;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const double a = red[threadIdx.x], b = red[threadIdx.x + o];
      red[threadIdx.x] = (a != a || b != b) ? (a != a ? a : b) : (a > b ? a : b);   // NaN propagates like numpy.max
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) block_max[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sub_max_kernel(double* __restrict__ p, int64_t n, const double* __restrict__ block_max,
                                                      int blocks) {
  double m = block_max[0];
  for (int b = 1; b < blocks; ++b) {
    const double v = block_max[b];
    m = (m != m || v != v) ? (m != m ? m : v) : (v > m ? v : m);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = p[i] - m;
}

int check_batch(const void* signal, const int64_t* sig_off, const int64_t* frame_off, int32_t n_utts, int64_t n_frames,
                const void* out, int sample_type) {
  if (n_utts < 0 || n_frames < 0) return fail(-1, "negative batch size");
  if (n_frames == 0 || n_utts == 0) return 0;
  if (!signal || !sig_off || !frame_off || !out) return fail(-1, "NULL device pointer");
  if (sample_type < TFK_SAMPLE_I16 || sample_type > TFK_SAMPLE_F32) return fail(-1, "unknown sample type %d", sample_type);
  return 0;
}

}  // namespace

struct tfk_feat {
  tfk_feat_config cfg;
  int nbins, d0, dim, log2_n2, fb_nnz, dct_lds, waves, n_items, n_meta, num_cus;
  size_t lds_bytes;
  double* tables = nullptr;  // one allocation: filter values | binw | dct | lift | twiddles | filter meta (ints)
  double *fb_val = nullptr, *binw = nullptr, *dct = nullptr, *lift = nullptr;
  double2* tw = nullptr;
  int* fb_meta = nullptr;
  double* work = nullptr;    // static features of a batch that needs deltas, [frames][d0] float64
  size_t work_cap = 0;
  FrameMeta* fmeta = nullptr;  // [frames] of the batch in flight
  size_t fmeta_cap = 0;
};

namespace {

constexpr size_t kLdsBudget = 160 * 1024;

FrameArgs frame_args(const tfk_feat* f, const void* signal, const int64_t* sig_off, const int64_t* frame_off,
                     int n_utts, int64_t n_frames) {
  FrameArgs a;
  memset(&a, 0, sizeof(a));
  a.b = Batch{signal, sig_off, frame_off, n_utts, n_frames};
  a.frame_len = f->cfg.frame_len; a.frame_step = f->cfg.frame_step; a.nfft = f->cfg.nfft; a.log2_n2 = f->log2_n2;
  a.nfilt = f->cfg.nfilt; a.ncep = f->cfg.kind == TFK_FEAT_MFCC ? f->cfg.numcep : 0; a.kind = f->cfg.kind;
  a.include_energy = f->cfg.include_energy;
  a.preemph = f->cfg.preemph; a.inv_nfft = 1.0 / f->cfg.nfft;
  a.fb_meta = f->fb_meta; a.fb_val = f->fb_val; a.fb_nnz = f->fb_nnz; a.dct_lds = f->dct_lds;
  a.n_items = f->n_items; a.n_meta = f->n_meta;
  a.binw = f->binw; a.dct = f->dct; a.lift = f->lift; a.tw = f->tw;
  return a;
}

int launch_frames(tfk_feat* f, hipStream_t st, FrameArgs& a, int sample_type) {
  const size_t need = (size_t)a.b.n_frames;
  if (need > f->fmeta_cap) {                                // stream-ordered growth: earlier launches may still read it
    if (f->fmeta) HIPCHK(hipFreeAsync(f->fmeta, st));
    f->fmeta = nullptr; f->fmeta_cap = 0;
    const size_t cap = need + need / 4;
    HIPCHK(hipMallocAsync((void**)&f->fmeta, cap * sizeof(FrameMeta), st));
    f->fmeta_cap = cap;
  }
  hipLaunchKernelGGL(frame_meta_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, st, a.b, a.frame_step, f->fmeta);
  HIPCHK(hipGetLastError());
  a.fmeta = f->fmeta;
  const int waves = f->waves;
  const int64_t blocks_needed = (a.b.n_frames + waves - 1) / waves;
  const int64_t resident = (int64_t)f->num_cus * std::max<int64_t>(1, (int64_t)(kLdsBudget / f->lds_bytes));
  const unsigned grid = (unsigned)std::min<int64_t>(blocks_needed, resident);
  const bool fixed = f->cfg.nfft == 512;
#define TFK_LAUNCH_FRAMES(S)                                                                                              \
  do {                                                                                                                    \
    if (fixed) hipLaunchKernelGGL((feat_frames_kernel<S, 8>), dim3(grid), dim3(64 * waves), f->lds_bytes, st, a);        \
    else hipLaunchKernelGGL((feat_frames_kernel<S, 0>), dim3(grid), dim3(64 * waves), f->lds_bytes, st, a);              \
  } while (0)
  if (sample_type == TFK_SAMPLE_I16) TFK_LAUNCH_FRAMES(TFK_SAMPLE_I16);
  else if (sample_type == TFK_SAMPLE_F32) TFK_LAUNCH_FRAMES(TFK_SAMPLE_F32);
  else TFK_LAUNCH_FRAMES(TFK_SAMPLE_F64);
#undef TFK_LAUNCH_FRAMES
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int tfk_feat_create(const tfk_feat_config* cfg, const double* filterbank, const double* bin_weight, const double* dct,
                    const double* lifter, tfk_feat** out) {
  if (!cfg || !out) return fail(-1, "cfg / out is NULL");
  if (cfg->struct_size != (int32_t)sizeof(tfk_feat_config))
    return fail(-1, "tfk_feat_config.struct_size %d != %zu", cfg->struct_size, sizeof(tfk_feat_config));
  if (cfg->kind < TFK_FEAT_FBANK || cfg->kind > TFK_FEAT_FBANK_RAW) return fail(-1, "unknown feature type %d", cfg->kind);
  if (cfg->dynamic < TFK_DYN_NODELTA || cfg->dynamic > TFK_DYN_DDELTA) return fail(-1, "unknown dynamic type %d", cfg->dynamic);
  if (cfg->frame_len < 1 || cfg->frame_step < 1) return fail(-1, "frame_len / frame_step must be positive");
  int lg = 0;
  while ((1 << lg) < cfg->nfft) ++lg;
  if (cfg->nfft < 32 || cfg->nfft > kMaxFft || (1 << lg) != cfg->nfft)
    return fail(-1, "nfft %d: the device transform takes a power of two in [32, %d]", cfg->nfft, kMaxFft);
  if (cfg->nfilt < 1 || cfg->nfilt > cfg->nfft / 2) return fail(-1, "nfilt %d must be in [1, nfft/2]", cfg->nfilt);
  if (!filterbank) return fail(-1, "filterbank is NULL");
  if (cfg->kind == TFK_FEAT_MFCC && (cfg->numcep < 1 || cfg->numcep > cfg->nfilt || !dct || !lifter))
    return fail(-1, "mfcc needs 1 <= numcep <= nfilt, a DCT matrix and lifter weights");
  if (cfg->kind == TFK_FEAT_SSC && !bin_weight) return fail(-1, "ssc needs the bin weights");
  HIPCHK(hipSetDevice(cfg->device));
  tfk_feat* f = new tfk_feat();
  f->cfg = *cfg;
  f->nbins = cfg->nfft / 2 + 1;
  f->log2_n2 = lg - 1;
  f->d0 = (cfg->kind == TFK_FEAT_MFCC ? cfg->numcep : cfg->nfilt) + (cfg->include_energy ? 1 : 0);
  f->dim = f->d0 * (1 + cfg->dynamic);
  const int ncep = cfg->kind == TFK_FEAT_MFCC ? cfg->numcep : 0;
  const int nfilt = cfg->nfilt, nbins = f->nbins, N2 = cfg->nfft / 2;
  // the support of every filter (a triangle of base.get_filterbanks; any matrix works, zeros inside a support are kept)
  std::vector<int> meta(3 * nfilt + 1, 0);
  std::vector<double> vals;
  for (int j = 0; j < nfilt; ++j) {
    int lo = nbins, hi = 0;
    for (int k = 0; k < nbins; ++k)
      if (filterbank[(size_t)j * nbins + k] != 0.0) { lo = k < lo ? k : lo; hi = k + 1; }
    if (hi <= lo) { lo = 0; hi = 0; }
    meta[j] = lo; meta[nfilt + j] = hi - lo; meta[2 * nfilt + j] = (int)vals.size();
    for (int k = lo; k < hi; ++k) vals.push_back(filterbank[(size_t)j * nbins + k]);
  }
  f->fb_nnz = (int)vals.size();
  // pieces for the lanes: the smallest piece length C with sum_j ceil(cnt_j / C) <= 64; needs room for the partial
  // sums next to the log-energies in one transform buffer (2 N2 doubles)
  f->n_items = 0;
  if (nfilt <= 64 && 2 * N2 >= nfilt + 128) {
    int C = 1;
    for (;; ++C) {
      int items = 0;
      for (int j = 0; j < nfilt; ++j) items += meta[nfilt + j] ? (meta[nfilt + j] + C - 1) / C : 0;
      if (items <= 64) break;
    }
    std::vector<int> it(4 * 64, 0), first(nfilt + 1, 0);
    int n = 0;
    for (int j = 0; j < nfilt; ++j) {
      first[j] = n;
      for (int s0 = 0; s0 < meta[nfilt + j]; s0 += C, ++n) {
        it[n] = j; it[64 + n] = meta[j] + s0; it[128 + n] = std::min(C, meta[nfilt + j] - s0); it[192 + n] = meta[2 * nfilt + j] + s0;
      }
    }
    first[nfilt] = n;
    f->n_items = n;
    meta.resize(3 * nfilt);
    meta.insert(meta.end(), it.begin(), it.end());
    meta.insert(meta.end(), first.begin(), first.end());
  } else {
    meta.resize(3 * nfilt);
  }
  f->n_meta = (int)meta.size();
  const size_t n_dct = (size_t)nfilt * ncep, n_tw = (size_t)cfg->nfft /* nfft/2 double2 */, n_meta = (meta.size() + 1) / 2 + 1;
  std::vector<double> host(vals.size() + nbins + n_dct + ncep + n_tw + n_meta, 0.0);
  double* h = host.data();
  memcpy(h, vals.data(), vals.size() * sizeof(double));
  h += vals.size();
  if (bin_weight) memcpy(h, bin_weight, nbins * sizeof(double));
  h += nbins;
  if (ncep) { memcpy(h, dct, n_dct * sizeof(double)); memcpy(h + n_dct, lifter, ncep * sizeof(double)); }
  h += n_dct + ncep;
  for (int k = 0; k < N2; ++k) {
    const long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / cfg->nfft;
    h[2 * k] = (double)cosl(ang);
    h[2 * k + 1] = (double)sinl(ang);
  }
  h += n_tw;
  memcpy(h, meta.data(), meta.size() * sizeof(int));
  // LDS: twiddles + filter values (+ ssc weights, + the DCT matrix while it is small) + meta, then 4 N2 doubles per
  // wave; as many waves (frames) per block as fit next to a second block on the CU, eight at most
  f->dct_lds = n_dct * sizeof(double) <= 16 * 1024;
  const size_t shared_raw = ((size_t)2 * N2 + vals.size() + (cfg->kind == TFK_FEAT_SSC ? nbins : 0) + (f->dct_lds ? n_dct : 0)) * sizeof(double) +
                        meta.size() * sizeof(int);
  const size_t shared = (shared_raw + 15) & ~(size_t)15;
  const size_t per_wave = (size_t)4 * N2 * sizeof(double);
  if (shared + per_wave > kLdsBudget) {
    delete f;
    return fail(-1, "this configuration needs %zu bytes of LDS for one frame (160 KB per CU)", shared + per_wave);
  }
  int waves = 8;
  while (waves > 1 && 2 * (shared + waves * per_wave) > kLdsBudget) waves >>= 1;
  f->waves = waves;
  f->num_cus = 256;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) f->num_cus = cus;
  }
  f->lds_bytes = shared + waves * per_wave;
  if (f->lds_bytes > 64 * 1024) {
    hipError_t ea = hipSuccess;
    const void* variants[6] = {(const void*)feat_frames_kernel<TFK_SAMPLE_I16, 0>, (const void*)feat_frames_kernel<TFK_SAMPLE_I16, 8>,
                               (const void*)feat_frames_kernel<TFK_SAMPLE_F64, 0>, (const void*)feat_frames_kernel<TFK_SAMPLE_F64, 8>,
                               (const void*)feat_frames_kernel<TFK_SAMPLE_F32, 0>, (const void*)feat_frames_kernel<TFK_SAMPLE_F32, 8>};
    for (int v = 0; v < 6 && ea == hipSuccess; ++v)
      ea = hipFuncSetAttribute(variants[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->lds_bytes);
    if (ea != hipSuccess) { delete f; return fail((int)ea, "LDS size attribute: %s", hipGetErrorString(ea)); }
  }
  hipError_t e = hipMalloc((void**)&f->tables, host.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(f->tables, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (f->tables) hipFree(f->tables);
    delete f;
    return fail((int)e, "feature tables: %s", hipGetErrorString(e));
  }
  f->fb_val = f->tables;
  f->binw = f->fb_val + vals.size();
  f->dct = f->binw + nbins;
  f->lift = f->dct + n_dct;
  f->tw = (double2*)(f->lift + ncep);
  f->fb_meta = (int*)(f->lift + ncep + n_tw);
  *out = f;
  return 0;
}

int tfk_feat_destroy(tfk_feat* f) {
  if (!f) return 0;
  if (f->work) hipFree(f->work);
  if (f->fmeta) hipFree(f->fmeta);
  if (f->tables) hipFree(f->tables);
  delete f;
  return 0;
}

int tfk_feat_dim(const tfk_feat* f, int32_t* dim) {
  if (!f || !dim) return fail(-1, "NULL argument");
  *dim = f->dim;
  return 0;
}

int tfk_feat_compute(tfk_feat* f, void* stream, const void* signal, int sample_type, const int64_t* sig_off,
                     const int64_t* frame_off, int32_t n_utts, int64_t n_frames, void* out, int64_t ld_out, int out_f64) {
  if (!f) return fail(-1, "plan is NULL");
  if (int rc = check_batch(signal, sig_off, frame_off, n_utts, n_frames, out, sample_type)) return rc;
  if (n_frames == 0 || n_utts == 0) return 0;
  if (ld_out < f->dim) return fail(-1, "ld_out %lld < feature dimension %d", (long long)ld_out, f->dim);
  HIPCHK(hipSetDevice(f->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  FrameArgs a = frame_args(f, signal, sig_off, frame_off, n_utts, n_frames);
  if (f->cfg.dynamic == TFK_DYN_NODELTA) {
    a.out = out; a.ld_out = ld_out; a.out_f64 = out_f64;
    return launch_frames(f, st, a, sample_type);
  }
  const size_t need = (size_t)n_frames * f->d0;
  if (need > f->work_cap) {                                 // stream-ordered growth: earlier launches may still read it
    if (f->work) HIPCHK(hipFreeAsync(f->work, st));
    f->work = nullptr; f->work_cap = 0;
    const size_t cap = need + need / 4;
    HIPCHK(hipMallocAsync((void**)&f->work, cap * sizeof(double), st));
    f->work_cap = cap;
  }
  a.out = f->work; a.ld_out = f->d0; a.out_f64 = 1;
  if (int rc = launch_frames(f, st, a, sample_type)) return rc;
  return tfk_feat_dynamic(stream, f->work, f->d0, f->d0, frame_off, n_utts, n_frames, f->cfg.dynamic, 0, out, ld_out, out_f64);
}

int tfk_feat_stage(tfk_feat* f, void* stream, int stage, const void* signal, int sample_type, const int64_t* sig_off,
                   const int64_t* frame_off, int32_t n_utts, int64_t n_frames, double* out, int64_t ld_out) {
  if (!f) return fail(-1, "plan is NULL");
  if (stage < TFK_STAGE_FRAMES || stage > TFK_STAGE_POWSPEC) return fail(-1, "unknown stage %d", stage);
  if (int rc = check_batch(signal, sig_off, frame_off, n_utts, n_frames, out, sample_type)) return rc;
  if (n_frames == 0 || n_utts == 0) return 0;
  const int cols = stage == TFK_STAGE_FRAMES ? f->cfg.frame_len : f->nbins;
  if (ld_out < cols) return fail(-1, "ld_out %lld < %d columns of this stage", (long long)ld_out, cols);
  HIPCHK(hipSetDevice(f->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  FrameArgs a = frame_args(f, signal, sig_off, frame_off, n_utts, n_frames);
  a.out = out; a.ld_out = ld_out; a.out_f64 = 1; a.stage = stage;
  if (stage != TFK_STAGE_FRAMES) return launch_frames(f, st, a, sample_type);
  if (n_frames > 0x7fffffffLL) return fail(-1, "too many frames for one launch");
  if (sample_type == TFK_SAMPLE_I16)
    hipLaunchKernelGGL(frames_kernel<TFK_SAMPLE_I16>, dim3((unsigned)n_frames), dim3(256), 0, st, a);
  else if (sample_type == TFK_SAMPLE_F32)
    hipLaunchKernelGGL(frames_kernel<TFK_SAMPLE_F32>, dim3((unsigned)n_frames), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(frames_kernel<TFK_SAMPLE_F64>, dim3((unsigned)n_frames), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

int tfk_feat_dynamic(void* stream, const double* x, int64_t ld_x, int32_t dim, const int64_t* row_off, int32_t n_utts,
                     int64_t n_rows, int dynamic, int deriv_only, void* out, int64_t ld_out, int out_f64) {
  if (n_rows == 0 || n_utts == 0 || dim == 0) return 0;
  if (!x || !row_off || !out || n_rows < 0 || n_utts < 0 || dim < 0) return fail(-1, "bad argument");
  if (dynamic < TFK_DYN_NODELTA || dynamic > TFK_DYN_DDELTA) return fail(-1, "unknown dynamic type %d", dynamic);
  const int64_t cols = deriv_only ? dim : (int64_t)dim * (1 + dynamic);
  if (ld_x < dim || ld_out < cols) return fail(-1, "leading dimension smaller than the row");
  DynArgs a{x, ld_x, dim, row_off, n_utts, n_rows, dynamic, deriv_only, out, ld_out, out_f64};
  const int64_t total = n_rows * dim;
  hipLaunchKernelGGL(dynamic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(hipGetLastError());
  return 0;
}

int tfk_deframesig(void* stream, const double* frames, int64_t ld, int64_t n_frames, int32_t frame_len, int32_t frame_step,
                   const double* win, double* out) {
  if (n_frames <= 0) return 0;
  if (!frames || !out || frame_len < 1 || frame_step < 1 || ld < frame_len) return fail(-1, "bad argument");
  const int64_t padlen = (n_frames - 1) * frame_step + frame_len;
  hipLaunchKernelGGL(deframe_kernel, dim3((unsigned)((padlen + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, ld,
                     n_frames, frame_len, frame_step, win, padlen, out);
  HIPCHK(hipGetLastError());
  return 0;
}

int tfk_logpow(void* stream, double* p, int64_t n, int norm, double* scratch) {
  if (n <= 0) return 0;
  if (!p || !scratch) return fail(-1, "NULL argument");
  const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(logpow_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, scratch);
  if (norm) hipLaunchKernelGGL(sub_max_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, scratch, blocks);
  HIPCHK(hipGetLastError());
  return 0;
}

int tfk_cmvn_stats(void* stream, const float* feats, int64_t ld, int32_t dim, const int64_t* spk_off,
                   const int64_t* utt_row, const int64_t* utt_len, int32_t n_spk, double* stats) {
  if (n_spk == 0) return 0;
  if (!feats || !spk_off || !utt_row || !utt_len || !stats || dim < 1 || n_spk < 0 || ld < dim)
    return fail(-1, "bad argument");
  hipLaunchKernelGGL(cmvn_stats_kernel, dim3((unsigned)n_spk, (unsigned)(dim / 64 + 1)), dim3(64), 0,
                     (hipStream_t)stream, feats, ld, dim, spk_off, utt_row, utt_len, stats);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
