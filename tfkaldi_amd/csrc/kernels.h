// Non-GEMM kernels of the DNN hot path (gfx950): interface.  See kernels.hip for the reference
// citations of each op.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "x3_layout.h"

namespace tfk {

// The activation chain of one hidden layer (neuralNetworks/classifiers/activation.py:22-42 order:
// Batchnorm -> nonlin -> L2Norm -> Dropout), plus the dropout RNG coordinates.
struct ActDesc {
  int nonlin;      // TFK_NONLIN_*
  int bn;          // batch norm on
  int l2;          // L2Norm on
  float keep;      // dropout keep probability (>= 1: off)
  uint64_t seed;   // RNG key
  uint32_t call;   // RNG counter: accumulate-call index
  uint32_t layer;  // RNG counter: layer index
  int train;       // dropout only in training mode
};

constexpr int kMaxRowSplits = 256;

// Mixed-precision mode: kernels that produce a GEMM operand also store its bf16 twin (p == nullptr: fp32 mode).
// ld in bf16 elements, a multiple of 8; padding columns of the twin stay zero from its allocation.
// x3 != 0 ("bf16x3", gemm_bf16.h): the twin is THREE bf16 planes whose sum is the fp32 value exactly (twin_split3 below), in
// the tiled layout of x3_layout.h (2-row x 32-column units of three 128-byte lines; ld a multiple of 32; x3::elems(rows, ld)
// elements in all; a row offset into such a twin must be even); x3 == 0: one plane, round to nearest even.
struct Twin {
  uint16_t* p = nullptr;
  int ld = 0;
  int x3 = 0;
};
// x = p1 + p2 + p3 exactly, each a bf16, rounded to nearest: x3::split3 (x3_layout.h), the ONE split every producer uses
__device__ __forceinline__ void twin_split3(float x, uint16_t& p1, uint16_t& p2, uint16_t& p3) { x3::split3(x, p1, p2, p3); }
__device__ __forceinline__ void twin_put(const Twin& t, size_t row, int col, float v) {
  if (t.x3) {
    uint16_t a, b, c;
    twin_split3(v, a, b, c);
    const size_t at = x3::at(row, col, t.ld);
    t.p[at] = a; t.p[at + 64] = b; t.p[at + 128] = c;
  } else {
    t.p[row * t.ld + col] = __builtin_bit_cast(uint16_t, (__bf16)v);
  }
}

// ---- batch norm statistics ----
// train: per-column mean / biased variance of z[T,H] (two-level, Chan-merged), rstd = rsqrt(var+eps);
// the moving-average increments follow E <- decay*E + (1-decay)*stat.  ws: >= 2*kMaxRowSplits*ld floats.
void bn_stats_train(hipStream_t s, const float* z, int T, int H, int ld, float eps, float decay, float* mean,
                    float* rstd, float* e_mean, float* e_var, float* ws);
// eval: mean = moving_mean, rstd = rsqrt(moving_var + eps)
void bn_stats_eval(hipStream_t s, const float* mov_mean, const float* mov_var, int H, float eps, float* mean,
                   float* rstd);

// ---- fused training-mode batch norm + activation chain forward (no L2Norm) ----
// stats = per-chunk (mean, M2) of z's columns as written by the forward GEMM's EPI_COLSTATS epilogue
// ([2, nchunk, ld], chunk k = rows [k*chunk_rows, ...)).  Merges them (Chan), publishes mean / rstd for the
// backward pass, advances the moving-average increments E <- decay*E + (1-decay)*stat, and writes
// a = dropout(nonlin((z - mean) * rstd + beta)).
void bn_act_forward(hipStream_t s, const ActDesc& d, const float* z, float* a, const float* stats, int chunk_rows,
                    int T, int H, int ld, float eps, float decay, float* mean, float* rstd, float* e_mean,
                    float* e_var, const float* beta, Twin tw = Twin(), int T_apply = 0, int slab_chunks = 0);
// (a segment of a STACKED pass: T rows carry the statistics, rows [T, T_apply) behind them are padding that receives
// finite outputs and counts for nothing; slab_chunks = chunks per statistics slab when `stats` points into a larger table)

// ---- activation chain forward: z -> a (v: post-nonlin copy, rowscale: L2 mean-square; both only if l2) ----
void act_forward(hipStream_t s, const ActDesc& d, const float* z, float* a, float* v, float* rowscale,
                 const float* mean, const float* rstd, const float* beta, int T, int H, int ld, Twin tw = Twin());

// ---- activation chain backward ----
// L2 chains only: da -> du (gradient w.r.t. the BN output) in place, row-wise.
void act_backward_rows(hipStream_t s, const ActDesc& d, float* da, const float* v, const float* rowscale, int T,
                       int H, int ld);
// Column-tiled backward of {dropout, nonlin', batch norm}: da[T,H] -> dz in place.  Leaves the per-chunk
// partial column sums in ws (>= 3*kMaxRowSplits*ld floats): slab 0 = sum_t du (-> d beta), slab 1 =
// sum_t du*xhat, slab 2 = sum_t dz (-> d bias); grad_final() turns them into gradients.
// Slab k starts at row k*kMaxRowSplits of ws ([., ld]).
// `pre_du` != 0: da already holds du (after act_backward_rows, or from the dA GEMM's EPI_DACT epilogue).
// `stats_chunks` > 0: slabs 0 / 1 already hold that many chunks (EPI_DACT): the statistics pass is skipped.
void hidden_backward(hipStream_t s, const ActDesc& d, int pre_du, float* da, const float* a, const float* z,
                     const float* mean, const float* rstd, int T, int H, int ld, float* ws, int stats_chunks = 0,
                     Twin tw = Twin(), int T_apply = 0, float* ws_dz = nullptr);
// (stacked pass: rows [T, T_apply) are padding -- their dz is written as zero; ws_dz = where this launch's slab-2
// partial sums go, default slab 2 of ws)
// per-chunk partial column sums of x[T, ld] into slab 0 of ws (bias gradient of the output layer)
// Tall micro-batches (more than kMergeOnceChunks GEMM row tiles): merge the per-tile statistics / partial sums ONCE
// instead of in every block of the column-tiled kernels.
constexpr int kMergeOnceChunks = 32;
// stats = the EPI_COLSTATS output [2, ceil(T / chunk_rows), ld] -> mean, rstd and the moving-average increments
void bn_stats_from_chunks(hipStream_t s, const float* stats, int chunk_rows, int T, int H, int ld, float eps, float decay,
                          float* mean, float* rstd, float* e_mean, float* e_var);
// ws slabs 0 and 1 (EPI_DACT output): entry 0 <- sum of the first `chunks` entries
void chunk_totals(hipStream_t s, float* ws, int chunks, int ld);
void colsum_partial(hipStream_t s, const float* x, int T, int ld, float* ws);
// colsum_partial(x, T, ld, ws) and loss_reduce(row_loss, T_loss, scalars, overwrite, frames, microbatches) in ONE launch
void colsum_loss(hipStream_t s, const float* x, int T, int ld, float* ws, const float* row_loss, int T_loss, float* scalars,
                 bool overwrite, int frames = -1, int microbatches = 1);
int row_splits(int T);  // chunks the column-tiled kernels cut T rows into

// g[c] (+)= sum over the chunks of slab `which` of ws -- for MANY (layer, vector) pairs in one launch
struct FinalItem {
  const float* ws;
  float* g;
  int which, rs, N, ld;
};
constexpr int kMaxFinalItems = 48;
struct FinalBatch {
  FinalItem it[kMaxFinalItems];
  int n;
  int accumulate;  // 0: first micro-batch of a step overwrites G
};
void grad_final(hipStream_t s, const FinalBatch& b);

// ---- softmax cross-entropy (trainer.py:526-531): row_loss[t] = logsumexp(z_t) - z_t[y_t];
// with_grad: logits <- softmax(z) - onehot(y) in place (sum-reduced loss => no 1/T factor).
// y[t] < 0: the row holds no frame (padding of a stacked pass): loss 0, gradient row 0.
// (the variant that also sums the losses in the same launch -- last block by ticket -- was measured slower than the
// second launch and is archived: tools/experiments/r03_softmax_xent_ticket.patch)
void softmax_xent(hipStream_t s, float* logits, const int32_t* y, int T, int O, int ld, float* row_loss,
                  int with_grad, Twin tw = Twin());
// scalars[0] += sum(row_loss), scalars[1] += T, scalars[2] += 1
// overwrite: the accumulators were logically re-initialised since the last call (no memset needed)
// frames / microbatches: what the T rows stand for (a stacked pass sums k micro-batches whose rows include padding)
void loss_reduce(hipStream_t s, const float* row_loss, int T, float* scalars, bool overwrite, int frames = -1,
                 int microbatches = 1);
// decoder.py:44 softmax; prior != null: out = log(softmax / prior) (nnet.py:280-286)
void softmax_rows(hipStream_t s, const float* logits, int T, int O, int ld, float* out, int64_t ldo,
                  const float* prior);

// Where the weight matrices' x3 twins are, for the optimiser: matrix l occupies arena elements [begin[l], begin[l] + rows[l] *
// ld[l]) (row-major, leading dimension ld[l]) and its twin starts `twin[l]` elements behind the shadow's base with leading
// dimension ld_twin[l].  Up to kShadowMapMax matrices (deeper nets rebuild the twins after the update instead).
constexpr int kShadowMapMax = 16;
struct ShadowMap {
  int n;
  uint32_t begin[kShadowMapMax], rows[kShadowMapMax], ld[kShadowMapMax], ld_twin[kShadowMapMax];
  uint64_t twin[kShadowMapMax];
};

// ---- optimiser (trainer.py:174-184): g = clip(G / num_frames, -1, 1); TF Adam (G is left as is) ----
// grid_cap > 0 limits the number of blocks (grid-stride loop does the rest)
void adam_apply(hipStream_t s, float* w, float* g, float* m, float* v, size_t n, const float* scalars, float lr_t,
                float beta1, float beta2, float eps, int grid_cap, uint16_t* wb = nullptr, size_t n_wb = 0,
                const struct ShadowMap* map = nullptr, size_t first = 0);
// wb: bf16 shadow of the first n_wb parameters (the weight matrices), written with the update: element for element behind
// `wb` (mixed precision), or -- map != nullptr, x3 -- into the tiled three-plane twins the map describes; w[0] is then arena
// element `first`
// fp32 [rows, lds] -> bf16 [rows, ldd] with zero padding columns (x3: the tiled three-plane twin, ldd a multiple of 32)
void to_bf16_rows(hipStream_t s, const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols, int x3 = 0);
// end of a step in one launch: moving <- decay^{num_microbatches} * moving + E, E <- 0, and
// host[0..3] <- scalars[0..3] (host = device address of mapped pinned memory)
// snap (nullable): device copy of scalars[0..3] that stays valid until the next step_finish
// seq: written to word kStepSeqWord of `host` behind the scalars (system-scope release): what a polling host waits for
constexpr int kStepSeqWord = 12;
void step_finish(hipStream_t s, float* moving, float* e, size_t n, const float* scalars, float decay, float* host,
                 float* snap = nullptr, unsigned seq = 0);
void scale_inplace(hipStream_t s, float* x, size_t n, float factor);
void fill(hipStream_t s, float* x, size_t n, float value);
// +-context splicing on the device (reference processing/feature_reader.py:117-156): raw[T, ldr] holds the
// unspliced frames of the utterances back to back, seg[U+1] their start offsets; out[t, j*D + d] =
// raw[t + j - c, d] when that frame belongs to the same utterance, else 0.  Pad columns of out are zeroed.
// cmvn (nullable) = per-utterance [U, 2, D] (mean, standard deviation): the spliced value is
// (raw - mean) / std, IEEE-rounded like numpy's float32 subtract / divide (feature_reader.py:109-115).
// out_seg (nullable, [U]): utterance u's spliced rows go to rows out_seg[u] .. of `out` instead of seg[u] .. (a stacked
// pass leaves padding rows between its segments).
void splice_frames(hipStream_t s, const float* raw, int ldr, const int32_t* seg, int U, int T, int D, int context,
                   const float* cmvn, float* out, int ldo, const int32_t* out_seg = nullptr);
// *out += sum_i p[i] * ((i & 0xffff) + 1) over n 32-bit words (unsigned 64-bit arithmetic; *out zeroed by the caller)
void checksum_words(hipStream_t s, const uint32_t* p, size_t n, unsigned long long* out);
// debug: regenerate the keep mask of a layer as 0/1 floats [T, ld]
void dropout_mask(hipStream_t s, const ActDesc& d, float* out, int T, int H, int ld);

}  // namespace tfk
