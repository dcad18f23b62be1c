// C-ABI engine (include/tfkaldi_hip.h): owns the HBM layout, the HIP streams and the per-step kernel
// schedule of the DNN acoustic-model trainer.  Replaces what the reference executes inside
// tf.Session.run for neuralNetworks/trainer.py, decoder.py and classifiers/{dnn,layer,activation}.py.
//
// HBM layout (all fp32, every row 16-byte aligned, padding columns kept at zero):
//   persistent state (one allocation, optionally caller-owned so torch.distributed can reduce it):
//     [ params P ][ G (P) | scalars (64) | BN EMA increments (E) ][ Adam m (P) ][ Adam v (P) ][ BN moving (E) ]
//                  `--------------- reduce region ---------------'
//     per layer l the params are contiguous [ W_l (d_in x ld_out) | b_l | beta_l ] = one all-reduce bucket.
//   activations (engine-owned, grow on demand): X, per hidden layer z_l (pre-BN) and a_l (layer output),
//     logits, two ping-pong gradient buffers, BN batch statistics, reduction workspace.
#include "../../include/tfkaldi_hip.h"
#include "ctc.h"
#include "gemm_bf16.h"
#include "gemm_f32.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <chrono>

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

using namespace tfk;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code ? code : -1;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                        \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

inline size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct LayerLayout {
  int d_in, d_out, ld_in, ld_out;
  size_t w_off, b_off, beta_off;  // float offsets inside a P-sized region
  size_t w_sz, b_sz, beta_sz;     // span lengths (64-float aligned; beta_sz = 0 without batch norm)
};

enum KernelFamily {
  KF_GEMM_NN = 0, KF_GEMM_NT, KF_GEMM_TN, KF_GEMM_DUAL, KF_BN_STATS, KF_ACT_FWD, KF_HIDDEN_BWD, KF_COLSUM, KF_SOFTMAX_XENT,
  KF_LOSS_REDUCE, KF_SOFTMAX, KF_ADAM, KF_EMA, KF_MISC, KF_COUNT
};
const char* kFamilyName[KF_COUNT] = {"gemm_f32_nn(fwd affine)", "gemm_f32_nt(dA)",  "gemm_f32_tn(dW)",
                                     "gemm_f32_dual(dA+dW)",    "bn_stats",
                                     "act_forward",             "hidden_backward",  "colsum",          "softmax_xent",
                                     "loss_reduce",             "softmax_rows",     "adam_apply",      "bn_ema_apply",
                                     "misc"};

struct ProfRec {
  int family;
  hipEvent_t a, b;
  double flops, bytes;
};

}  // namespace

struct tfk_engine {
  tfk_config cfg;
  int F, L, H, O, ldF, ldH, ldO;
  float bn_decay, bn_eps, b1, b2, adam_eps;
  bool dropout;

  hipStream_t stream = nullptr, copy_stream = nullptr;
  bool own_stream = false;
  // Everything runs in order on ONE stream (plus the copy stream of the host-fed inputs).  Running the weight-
  // gradient GEMMs or the optimiser on a side stream was measured slower (profiles/r01_overlap_experiment.txt:
  // every GEMM already fills all 256 CUs, a concurrent kernel only evicts its L2 working set) and was removed;
  // the independent dA / dW pair of a layer shares one LAUNCH instead (gemm_f32_dual).
  // The optimiser of tfk_apply runs layer by layer on its OWN stream: the step is HBM-bound there (28 B per parameter)
  // while the next step's forward contractions are bound by the L2 -> LDS fill / the matrix pipes, and the next forward
  // pass only waits, layer by layer, for the update of the layer it is about to read (env TFK_ADAM_OVERLAP).
  hipStream_t opt_stream = nullptr;
  hipEvent_t ev_opt_begin = nullptr, ev_opt_vec = nullptr;
  std::vector<hipEvent_t> ev_adam;   // [L + 1]: the update of W_l (and its bf16 shadow) is complete
  bool opt_overlap = false;          // decided at create
  bool opt_pending = false;          // updates of the last tfk_apply may still be running on opt_stream
  float* d_snap = nullptr;           // (loss, frames, #micro-batches) of the step being applied (step_finish)
  hipEvent_t ev_loss = nullptr;
  // the loss hand-over without an event: step_finish writes loss_seq behind the scalars in mapped memory, tfk_apply_end polls it
  // (env TFK_LOSS_EVENT=1: the event record + synchronise of rounds 1-5 instead)
  unsigned loss_seq = 0;
  bool loss_event = false;
  bool slot_lazy = true;  // env TFK_SLOT_EVENT=1: record compute_done behind every micro-batch, as rounds 1-5 did
  hipEvent_t ev_grow = nullptr;          // orders the copy stream behind a stream-ordered (re)allocation
  std::vector<void*> host_garbage;       // outgrown pinned staging buffers: released at tfk_destroy (hipHostFree
                                         // synchronises the device, which growth must not do)

  // persistent state
  float* state = nullptr;
  bool own_state = false;
  size_t P = 0, E = 0, state_floats = 0;
  size_t off_param = 0, off_grad = 0, off_scalars = 0, off_ema = 0, off_m = 0, off_v = 0, off_mov = 0, off_shadow = 0;
  size_t reduce_floats = 0;
  std::vector<LayerLayout> lay;  // L + 1

  // activations
  int cap = 0;
  float* dX[2] = {nullptr, nullptr};
  int32_t* dY[2] = {nullptr, nullptr};
  std::vector<float*> z, a, v, rowscale, mean, rstd;
  float *logits = nullptr, *post = nullptr, *dA[2] = {nullptr, nullptr}, *row_loss = nullptr, *ws = nullptr;
  // device-side splice (tfk_*_raw): unspliced frames + utterance offsets, double-buffered like dX
  float* dRaw[2] = {nullptr, nullptr};
  int32_t* dSeg[2] = {nullptr, nullptr};
  float* hRaw[2] = {nullptr, nullptr};
  int32_t* hSeg[2] = {nullptr, nullptr};
  float* dCmvn[2] = {nullptr, nullptr};  // per-utterance (mean, std) tables of the raw entry points, grown on demand
  float* hCmvn[2] = {nullptr, nullptr};
  size_t cmvn_cap = 0;
  int seg_cap = 0;
  // stacked passes (tfk_accumulate_stacked*): batch mean / rstd of every SEGMENT, [kMaxStack, ldH] per hidden layer
  std::vector<float*> seg_mean, seg_rstd;
  float* ws_stats = nullptr;  // [2, ceil(cap / 64), ldH] per-tile BN statistics from the forward GEMM epilogue
  float* ws_bwd = nullptr;  // per-layer partial column sums of backward (finalised by one kernel)
  float* ws_splitk = nullptr;  // split-K partials of narrow weight-gradient GEMMs (grown on demand)
  size_t ws_splitk_floats = 0;
  size_t ws_bwd_stride = 0;
  float* prior = nullptr;
  bool have_prior = false;

  // host staging (pinned)
  float* hX[2] = {nullptr, nullptr};
  int32_t* hY[2] = {nullptr, nullptr};
  float* h_post = nullptr;
  size_t h_post_floats = 0;
  float* h_scalars = nullptr;    // mapped pinned memory: kernels write the step's (loss, frames, #mb) here
  float* h_scalars_dev = nullptr;
  bool apply_open = false, apply_direct = false;  // between tfk_apply_begin and tfk_apply_end
  float cur_lr_t = 0.f;
  bool scalars_fresh = true;     // batch_loss / num_frames / #mb are logically zero (next loss_reduce overwrites)
  bool colsum_done = false;      // the output layer's bias-gradient partial sums of THIS micro-batch are in the workspace already
                                 // (colsum_loss, launched with the loss sum right behind softmax_xent)
  bool fuse_hb_enabled = true;   // env TFK_FUSE_HB=0: separate statistics pass (experiments)
  bool dual_gemm = true;         // env TFK_DUAL_GEMM=0: dA and dW of a layer as two launches
  bool stack_enabled = true;     // env TFK_STACK=0: tfk_accumulate_stacked* run their micro-batches one after the other
  bool fuse_eval = true;         // env TFK_FUSE_EVAL=0: evaluation-mode layers as GEMM + bn_stats_eval + act_forward
  int post_chunk = 2048;         // env TFK_POST_CHUNK: rows per chunk of a pipelined tfk_posteriors pass (0: off)
  std::vector<hipEvent_t> post_ev;

  // CTC loss (tfk_accumulate_ctc): device copies of the utterance / label offsets and the state workspaces,
  // grown on demand
  int32_t *ctc_seg = nullptr, *ctc_lab_off = nullptr, *ctc_lab = nullptr;
  float *ctc_lp = nullptr, *ctc_ab = nullptr, *ctc_bb = nullptr, *ctc_utt_loss = nullptr, *ctc_lse = nullptr;
  double *ctc_off = nullptr, *ctc_offb = nullptr, *ctc_logz = nullptr;
  size_t ctc_cap_bb = 0, ctc_cap_offb = 0, ctc_cap_logz = 0;
  int32_t* h_ctc[2] = {nullptr, nullptr};  // pinned staging of (seg, lab_off, labels)
  size_t h_ctc_cap[2] = {0, 0};
  hipEvent_t ctc_staged[2] = {nullptr, nullptr};
  int ctc_stage_slot = 0;
  size_t ctc_cap_offrows = 0;
  size_t ctc_cap_seg = 0, ctc_cap_off = 0, ctc_cap_loss = 0, ctc_cap_lab = 0, ctc_cap_lp = 0, ctc_cap_ab = 0,
         ctc_cap_rows = 0;

  // mixed precision (cfg.compute_dtype == TFK_DTYPE_BF16): every fp32 buffer that is a GEMM operand has a bf16
  // twin written by its producer; master parameters, statistics, gradients and the optimiser stay fp32
  bool bf16 = false;                 // GEMM operands have bf16 twins (compute_dtype bf16 or f32x3)
  bool x3 = false;                   // TFK_DTYPE_F32X3: every twin is THREE bf16 planes summing to the fp32 value, in the tiled
                                     // layout of x3_layout.h (gemm_bf16x3)
  // elements the twin of an [rows, ld] matrix occupies / the offset of its row r (even in x3 mode) inside it
  size_t tw_elems(size_t rows, int ld) const { return x3 ? x3::elems(rows, ld) : rows * ld; }
  size_t tw_row(size_t r, int ld) const { return x3 ? x3::at(r, 0, ld) : r * ld; }
  ShadowMap wb_map;                  // x3: where the optimiser finds the twin of every weight matrix (n == 0: it cannot --
                                     // more than kShadowMapMax matrices -- and the twins are rebuilt after the update)
  int ldFb = 0, ldHb = 0, ldOb = 0;  // leading dimensions of the twins (multiples of 8 elements; of 32 in x3 mode)
  bf16_t* Xb[2] = {nullptr, nullptr};
  std::vector<bf16_t*> ab;
  bf16_t* logb = nullptr;
  bf16_t* dAb[2] = {nullptr, nullptr};
  bf16_t* Wb = nullptr;              // shadow of the weight matrices
  std::vector<size_t> wb_off;
  std::vector<int> wb_ld;
  bool wb_aligned = false;           // shadow offsets == fp32 arena offsets: Adam writes it with the update
  bool own_wb = false;               // packed shadow in its own allocation (else it is the tail of the state arena)
  unsigned long long* d_checksum = nullptr;
  bool shadow_dirty = true;
  hipEvent_t copy_done[2] = {nullptr, nullptr}, compute_done[2] = {nullptr, nullptr};
  bool slot_used[2] = {false, false};
  // When the LAST micro-batch of a step has read an input slot, no `compute_done` event is recorded behind it (a record between two
  // kernels costs the stream ~6 us, profiles/r06_loss_seq.txt): the slot is free once the HOST has seen that step's loss, which it
  // normally has long before the slot comes round again.  slot_step[s] = the optimiser step (count of tfk_apply_begin calls) whose
  // loss frees slot s, 0 = `compute_done[s]` was recorded as before;  steps_begun / steps_seen: apply_begin / apply_end (loss seen)
  unsigned long long slot_step[2] = {0, 0}, steps_begun = 0, steps_seen = 0;
  int slot = 0;

  // host-side scalars (trainer.py:98-106, dnn.py:85-89)
  int64_t global_step = 0, adam_t = 0;
  double lr_fact = 1.0;
  int initialised_layers = 0;
  uint32_t call_counter = 0;
  int later_mb = 0;
  // G is logically zero after apply (init_grads, trainer.py:350) but is not rewritten: the first
  // micro-batch of the next step overwrites it instead of accumulating (saves 3 x 4P bytes per step).
  bool grads_fresh = true;

  tfk_bucket_fn cb = nullptr;
  void* cb_user = nullptr;
  tfk_layer_fn layer_cb = nullptr;  // "the parameters of this layer are about to be read" (tfk_set_layer_callback)
  void* layer_user = nullptr;

  // last call (debug fetch)
  int last_T = 0, last_nfw = 0;
  uint32_t last_call = 0;
  const float* last_in = nullptr;

  // profiling
  bool profiling = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;

  int nact() const {
    if (!cfg.layerwise_init) return L;
    int n = initialised_layers + 1;
    if (n > L || initialised_layers < 0) n = L;
    return n;
  }
  float* p_param() const { return state + off_param; }
  float* p_grad() const { return state + off_grad; }
  float* p_scalars() const { return state + off_scalars; }
  float* p_ema() const { return state + off_ema; }
  float* p_m() const { return state + off_m; }
  float* p_v() const { return state + off_v; }
  float* p_mov() const { return state + off_mov; }
  float* ema_mean(int l) const { return p_ema() + (size_t)2 * l * ldH; }
  float* ema_var(int l) const { return p_ema() + (size_t)(2 * l + 1) * ldH; }
  float* mov_mean(int l) const { return p_mov() + (size_t)2 * l * ldH; }
  float* mov_var(int l) const { return p_mov() + (size_t)(2 * l + 1) * ldH; }
};

namespace {

int validate(const tfk_config* c) {
  if (!c) return fail(-1, "config is NULL");
  if (c->struct_size != (int32_t)sizeof(tfk_config))
    return fail(-1, "tfk_config.struct_size %d != %zu (ABI mismatch)", c->struct_size, sizeof(tfk_config));
  if (c->input_dim <= 0 || c->num_units <= 0 || c->output_dim <= 0 || c->num_layers < 1)
    return fail(-1, "bad dimensions F=%d L=%d H=%d O=%d", c->input_dim, c->num_layers, c->num_units, c->output_dim);
  if (c->nonlin < 0 || c->nonlin > 3) return fail(-1, "unkown nonlinearity %d", c->nonlin);
  if (!(c->keep_prob > 0.f)) return fail(-1, "dropout keep probability must be in (0, 1], got %g", c->keep_prob);
  if (c->compute_dtype != TFK_DTYPE_F32 && c->compute_dtype != TFK_DTYPE_BF16 && c->compute_dtype != TFK_DTYPE_F32X3)
    return fail(-1, "unknown compute_dtype %d", c->compute_dtype);
  return 0;
}

void compute_layout(const tfk_config* c, std::vector<LayerLayout>& lay, size_t& P, size_t& E) {
  const int L = c->num_layers;
  const int ldF = (int)up(c->input_dim, 4), ldH = (int)up(c->num_units, 4), ldO = (int)up(c->output_dim, 4);
  lay.resize(L + 1);
  size_t off = 0;
  // all weight matrices first (one all-reduce bucket each), then every bias / beta vector: the vectors'
  // gradients are finalised by ONE kernel at the end of backward and travel in the last bucket, right in
  // front of the scalar + BN tail of the reduce region
  for (int l = 0; l <= L; ++l) {
    LayerLayout& x = lay[l];
    x.d_in = l == 0 ? c->input_dim : c->num_units;
    x.ld_in = l == 0 ? ldF : ldH;
    x.d_out = l == L ? c->output_dim : c->num_units;
    x.ld_out = l == L ? ldO : ldH;
    x.w_off = off;
    x.w_sz = up((size_t)x.d_in * x.ld_out, 64);
    off += x.w_sz;
  }
  for (int l = 0; l <= L; ++l) {
    lay[l].b_off = off;
    lay[l].b_sz = up((size_t)lay[l].ld_out, 64);
    off += lay[l].b_sz;
  }
  for (int l = 0; l <= L; ++l) {
    lay[l].beta_off = off;
    lay[l].beta_sz = (c->batch_norm && l < L) ? up((size_t)ldH, 64) : 0;
    off += lay[l].beta_sz;
  }
  P = off;
  E = c->batch_norm ? up((size_t)2 * L * ldH, 64) : 0;
}

constexpr size_t kScalarFloats = 64;
constexpr int kErrWord = 8;  // index into h_scalars (mapped pinned memory): a kernel-reported failure, see check_kernel_errors
// Mixed precision: when every weight matrix has a leading dimension that is a multiple of 8 the bf16 shadow mirrors the
// fp32 arena element for element and lives at the END of the state arena (w_end bf16 values = w_end / 2 floats), so a
// host that owns the arena (torch.distributed) can all-gather SHARDS OF THE SHADOW ITSELF -- half the bytes of the
// fp32 parameters, and the next forward pass waits for them layer by layer.  Otherwise it is packed in its own allocation.
bool shadow_mirrors(const tfk_config* c, const std::vector<LayerLayout>& lay) {
  if (c->compute_dtype != TFK_DTYPE_BF16) return false;
  for (const LayerLayout& y : lay)
    if (y.ld_out % 8) return false;
  return true;
}
size_t shadow_floats(const tfk_config* c, const std::vector<LayerLayout>& lay) {
  return shadow_mirrors(c, lay) ? up(lay[0].b_off, 128) / 2 : 0;
}
size_t total_state_floats(size_t P, size_t E, size_t S) { return 4 * P + kScalarFloats + 2 * E + S; }

// ---- profiling helpers ----
hipEvent_t get_event(tfk_engine* e) {
  if (e->ev_next == e->ev_pool.size()) {
    hipEvent_t ev;
    hipEventCreate(&ev);
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_next++];
}
struct ProfScope {
  tfk_engine* e;
  ProfRec r;
  hipStream_t st;
  ProfScope(tfk_engine* e_, int family, double flops, double bytes, hipStream_t st_ = nullptr)
      : e(e_), st(st_ ? st_ : e_->stream) {
    if (!e->profiling) return;
    r.family = family; r.flops = flops; r.bytes = bytes;
    r.a = get_event(e); r.b = get_event(e);
    hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!e->profiling) return;
    hipEventRecord(r.b, st);
    e->prof.push_back(r);
  }
};

// everything this engine has in flight (main and copy streams)
int sync_streams(tfk_engine* e) {
  HIPCHK(hipStreamSynchronize(e->copy_stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->opt_stream) HIPCHK(hipStreamSynchronize(e->opt_stream));
  e->opt_pending = false;
  return 0;
}
// the engine stream behind every optimiser launch still in flight on the optimiser stream (anything that reads or writes
// parameters, moments or gradient sums outside the layer-by-layer forward pass)
int join_optimizer(tfk_engine* e) {
  if (e->opt_pending) {
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_adam[e->L], 0));  // recorded last: implies every earlier launch
    e->opt_pending = false;
  }
  return 0;
}
// the forward pass is about to read the parameters of `layer`
int wait_layer_update(tfk_engine* e, int layer, bool first) {
  if (!e->opt_pending) return 0;
  if (first) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_opt_vec, 0));  // bias / beta vectors of every layer
  HIPCHK(hipStreamWaitEvent(e->stream, e->ev_adam[layer], 0));
  if (layer == e->L) e->opt_pending = false;  // updates are launched in layer order: the last one covers all
  return 0;
}

// Device buffers that grow on demand are allocated and released in STREAM ORDER on the engine stream
// (hipMallocAsync / hipFreeAsync): growing for an unusually long micro-batch neither waits for the work already
// enqueued nor stalls the host -- the old buffers are released behind the kernels that still read them.
void dev_free(tfk_engine* e, void* p, bool async) {
  if (!p) return;
  if (async) (void)hipFreeAsync(p, e->stream);
  else (void)hipFree(p);
}
int dev_alloc(tfk_engine* e, void** p, size_t bytes, bool zero) {
  HIPCHK(hipMallocAsync(p, bytes, e->stream));
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, e->stream));
  return 0;
}
void host_retire(tfk_engine* e, void* p, bool async) {
  if (!p) return;
  if (async) e->host_garbage.push_back(p);
  else (void)hipHostFree(p);
}

// The host may still be writing parameters asynchronously (the sharded exchange step all-gathers the updated
// parameters while the next step's forward pass is already being enqueued): tell it which layer's parameters the
// next kernels read (-1: all of them), so that it can make the engine stream wait for exactly that write.
inline void need_params(tfk_engine* e, int layer) {
  if (e->layer_cb) e->layer_cb(e->layer_user, layer);
}

// bf16 twin of an fp32 GEMM operand buffer (mixed-precision mode)
// (x3 mode: the twin is the tiled three-plane array of x3_layout.h, x3::elems(rows, ld) elements)
const bf16_t* twin_of(tfk_engine* e, const float* p, int* ld) {
  for (int s = 0; s < 2; ++s) {
    if (p == e->dX[s]) { *ld = e->ldFb; return e->Xb[s]; }
    if (p == e->dA[s]) { *ld = e->ldHb; return e->dAb[s]; }
  }
  if (p == e->logits) { *ld = e->ldOb; return e->logb; }
  for (size_t l = 0; l < e->a.size(); ++l)
    if (p == e->a[l]) { *ld = e->ldHb; return e->ab[l]; }
  for (int l = 0; l <= e->L; ++l)
    if (p == e->p_param() + e->lay[l].w_off) { *ld = e->wb_ld[l]; return e->Wb + e->wb_off[l]; }
  return nullptr;
}
// rows a GEMM's per-tile statistics (EPI_COLSTATS / EPI_DACT) are chunked by
int gemm_chunk_rows(tfk_engine* e, GemmLayout layout, int M, int N, int K, int* cfg) {
  if (e->x3) { *cfg = -1; return kGemmBf16x3TileRows; }
  if (e->bf16) { *cfg = -1; return gemm_bf16_tile_rows(M, N); }
  *cfg = gemm_f32_pick_config(layout, M, N, K);
  return gemm_f32_config_bm(*cfg);
}
// (re)build the bf16 shadow of the weight matrices after the parameters were written from outside the optimiser
int refresh_shadow(tfk_engine* e) {
  if (!e->bf16 || !e->shadow_dirty) return 0;
  for (int l = 0; l <= e->L; ++l) {
    const LayerLayout& y = e->lay[l];
    to_bf16_rows(e->stream, e->p_param() + y.w_off, y.ld_out, e->Wb + e->wb_off[l], e->wb_ld[l], y.d_in, y.d_out, e->x3);
  }
  HIPCHK(hipGetLastError());
  e->shadow_dirty = false;
  return 0;
}

struct ActEpi {  // EPI_DACT operands: the hidden layer whose output gradient the GEMM produces
  const float *a, *z, *mean, *rstd;
  int nonlin;
  // EPI_EVAL_ACT (a == z == nullptr): mean / rstd = moving mean / moving VARIANCE (nullptr without batch norm)
  const float* beta = nullptr;
  float eps = 0.f;
  bf16_t* twin = nullptr;  // mixed precision: bf16 copy of the result
  int ld_twin = 0;
  float scale = 1.f;       // EPI_DACT: 1 / keep_prob of a ReLU + dropout chain
};
int run_gemm(tfk_engine* e, GemmLayout layout, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             int M, int N, int K, const float* bias, int epi, hipStream_t st = nullptr, float* stats = nullptr,
             int cfg = -1, const ActEpi* act = nullptr, const int* row_vend = nullptr) {
  if (!st) st = e->stream;
  if (e->bf16) {
    GemmArgsB b = {};
    int lda8 = 0, ldb8 = 0;
    b.A = twin_of(e, A, &lda8);
    b.B = twin_of(e, B, &ldb8);
    if (!b.A || !b.B) return fail(-1, "internal: GEMM operand without a bf16 twin");
    b.C = C; b.bias = bias; b.stats = stats;
    b.act_a = act ? act->a : nullptr; b.act_z = act ? act->z : nullptr;
    b.act_mean = act ? act->mean : nullptr; b.act_rstd = act ? act->rstd : nullptr;
    b.act_nonlin = act ? act->nonlin : 0;
    b.stats_stride = kMaxRowSplits;
    b.act_beta = act ? act->beta : nullptr; b.bn_eps = act ? act->eps : 0.f;
    b.C_twin = act ? act->twin : nullptr; b.ldct = act ? act->ld_twin : 0;
    b.ct_x3 = e->x3;
    b.act_scale = act ? act->scale : 1.f;
    b.act_keep = act ? 1.f / act->scale : 1.f;
    b.row_vend = row_vend;
    b.M = M; b.N = N; b.K = K; b.lda = lda8; b.ldb = ldb8; b.ldc = ldc; b.epi = epi;
    if (e->x3) {
      // the split-K form of the 1024-frame contractions (gemm_bf16.h); its flag words start out as zeros and every launch
      // leaves them so.  (The workspace is shared with the fp32 split-K form, which an x3 engine never runs.)
      const size_t need = gemm_bf16x3_splitk_floats(layout, M, N, K);
      if (need > e->ws_splitk_floats) {  // stream-ordered: behind the GEMMs that still use the old workspace
        dev_free(e, e->ws_splitk, true);
        e->ws_splitk = nullptr;
        e->ws_splitk_floats = 0;
        CHK(dev_alloc(e, (void**)&e->ws_splitk, need * sizeof(float), false));
        e->ws_splitk_floats = need;
        HIPCHK(hipMemsetAsync(e->ws_splitk, 0, need * sizeof(float), st));
      }
      b.splitk_ws = e->ws_splitk;
      b.splitk_ws_floats = e->ws_splitk_floats;
      b.err = reinterpret_cast<unsigned*>(e->h_scalars_dev + kErrWord);
    }
    const int fam = layout == GEMM_NN ? KF_GEMM_NN : layout == GEMM_NT ? KF_GEMM_NT : KF_GEMM_TN;
    ProfScope ps(e, fam, 2.0 * M * N * K,
                 (e->x3 ? 6.0 : 2.0) * ((double)M * K + (double)K * N) + 4.0 * (double)M * N * ((epi & EPI_ACCUM) ? 2 : 1), st);
    const int rc = e->x3 ? gemm_bf16x3(layout, b, st) : gemm_bf16(layout, b, st);
    if (rc != 0) return fail(rc, "gemm_bf16%s launch failed: %s", e->x3 ? "x3" : "", hipGetErrorString((hipError_t)rc));
    return 0;
  }
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.stats = stats;
  g.act_a = act ? act->a : nullptr; g.act_z = act ? act->z : nullptr;
  g.act_mean = act ? act->mean : nullptr; g.act_rstd = act ? act->rstd : nullptr;
  g.act_nonlin = act ? act->nonlin : 0;
  g.stats_stride = kMaxRowSplits;
  g.act_beta = act ? act->beta : nullptr; g.bn_eps = act ? act->eps : 0.f;
  g.act_scale = act ? act->scale : 1.f;
  g.act_keep = act ? 1.f / act->scale : 1.f;
  g.row_vend = row_vend;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = epi;
  if (layout == GEMM_TN && K >= gemm_f32_splitk_min_k() && (size_t)M * N < ((size_t)1 << 20)) {
    // narrow layer, many frames: the weight gradient may run split-K (gemm_f32.h) -- partials of up to 32 chunks
    const size_t need = (size_t)32 * M * ldc;
    if (need > e->ws_splitk_floats) {  // stream-ordered: behind the GEMMs that still read the old workspace
      dev_free(e, e->ws_splitk, true);
      e->ws_splitk = nullptr;
      e->ws_splitk_floats = 0;
      CHK(dev_alloc(e, (void**)&e->ws_splitk, need * sizeof(float), false));
      e->ws_splitk_floats = need;
    }
    g.splitk_ws = e->ws_splitk;
    g.splitk_ws_floats = e->ws_splitk_floats;
  }
  const int fam = layout == GEMM_NN ? KF_GEMM_NN : layout == GEMM_NT ? KF_GEMM_NT : KF_GEMM_TN;
  ProfScope ps(e, fam, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N * ((epi & EPI_ACCUM) ? 2 : 1)), st);
  const int rc = gemm_f32(layout, g, cfg, st);
  if (rc != 0) return fail(rc, "gemm_f32 launch failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

// dA = dZ . W^T (optionally with the EPI_DACT epilogue) and dW (+)= in^T . dZ -- independent, both reading dZ -- in
// ONE launch (gemm_f32_dual).  Returns 1 when the pair is not eligible (the caller launches them separately).
int run_gemm_dual(tfk_engine* e, const float* dz, int ld_dz, const float* W, int ldw, float* da_out, int ld_da, int T,
                  int N_da, int K_da, const ActEpi* act, float* stats, const float* in, int ld_in, float* Gw, int ld_g,
                  int d_in, int d_out, int epi_w, int* chunk_rows) {
  if (!e->dual_gemm) return 1;
  const double flops = 2.0 * T * N_da * K_da + 2.0 * d_in * d_out * T;
  if (e->bf16) {
    const int dcfg = e->x3 ? -1 : gemm_bf16_dual_config(T, N_da, d_in, d_out);
    const int bm = e->x3 ? kGemmBf16x3TileRows : gemm_bf16_dual_tile_rows(dcfg);
    if (!dcfg || (act && (T + bm - 1) / bm > kMaxRowSplits)) return 1;
    GemmArgsB a = {}, w = {};
    int ld_a = 0, ld_b = 0, ld_c = 0, ld_d = 0;
    a.A = twin_of(e, dz, &ld_a);
    a.B = twin_of(e, W, &ld_b);
    w.A = twin_of(e, in, &ld_c);
    w.B = twin_of(e, dz, &ld_d);
    if (!a.A || !a.B || !w.A || !w.B) return fail(-1, "internal: GEMM operand without a bf16 twin");
    a.C = da_out; a.stats = stats;
    a.act_a = act ? act->a : nullptr; a.act_z = act ? act->z : nullptr;
    a.act_mean = act ? act->mean : nullptr; a.act_rstd = act ? act->rstd : nullptr;
    a.act_nonlin = act ? act->nonlin : 0;
    a.act_scale = act ? act->scale : 1.f;
    a.act_keep = act ? 1.f / act->scale : 1.f;
    a.act_beta = act ? act->beta : nullptr;
    a.stats_stride = kMaxRowSplits;
    a.M = T; a.N = N_da; a.K = K_da; a.lda = ld_a; a.ldb = ld_b; a.ldc = ld_da; a.epi = act ? EPI_DACT : 0;
    w.C = Gw;
    w.M = d_in; w.N = d_out; w.K = T; w.lda = ld_c; w.ldb = ld_d; w.ldc = ld_g; w.epi = epi_w;
    const double ob = e->x3 ? 6.0 : 2.0;  // operand bytes per element: one bf16 value, or three planes
    const double bytes = ob * ((double)T * K_da + (double)K_da * N_da) + 4.0 * (double)T * N_da +
                         ob * ((double)T * d_in + (double)T * d_out) + 4.0 * (double)d_in * d_out * ((epi_w & EPI_ACCUM) ? 2 : 1);
    hipEvent_t pa = nullptr, pb = nullptr;
    if (e->profiling) {
      pa = get_event(e); pb = get_event(e);
      hipEventRecord(pa, e->stream);
    }
    const int rc = e->x3 ? gemm_bf16x3_dual(a, w, e->stream) : gemm_bf16_dual(a, w, e->stream);
    if (rc == -1) return 1;
    if (rc != 0) return fail(rc, "gemm_bf16%s_dual launch failed: %s", e->x3 ? "x3" : "", hipGetErrorString((hipError_t)rc));
    if (e->profiling) {
      hipEventRecord(pb, e->stream);
      ProfRec r;
      r.family = KF_GEMM_DUAL; r.flops = flops; r.bytes = bytes; r.a = pa; r.b = pb;
      e->prof.push_back(r);
    }
    if (chunk_rows) *chunk_rows = bm;
    return 0;
  }
  GemmArgs a, w;
  a.A = dz; a.B = W; a.C = da_out; a.bias = nullptr; a.stats = stats;
  a.act_a = act ? act->a : nullptr; a.act_z = act ? act->z : nullptr;
  a.act_mean = act ? act->mean : nullptr; a.act_rstd = act ? act->rstd : nullptr;
  a.act_nonlin = act ? act->nonlin : 0;
  a.act_scale = act ? act->scale : 1.f;
  a.act_keep = act ? 1.f / act->scale : 1.f;
  a.act_beta = act ? act->beta : nullptr;
  a.stats_stride = kMaxRowSplits;
  a.M = T; a.N = N_da; a.K = K_da; a.lda = ld_dz; a.ldb = ldw; a.ldc = ld_da; a.epi = act ? EPI_DACT : 0;
  w.A = in; w.B = dz; w.C = Gw; w.bias = nullptr; w.stats = nullptr;
  w.act_a = w.act_z = w.act_mean = w.act_rstd = nullptr; w.act_nonlin = 0; w.stats_stride = 0;
  w.M = d_in; w.N = d_out; w.K = T; w.lda = ld_in; w.ldb = ld_dz; w.ldc = ld_g; w.epi = epi_w;
  const double bytes = 4.0 * ((double)T * K_da + (double)K_da * N_da + (double)T * N_da) +
                       4.0 * ((double)T * d_in + (double)T * d_out + (double)d_in * d_out * ((epi_w & EPI_ACCUM) ? 2 : 1));
  hipEvent_t pa = nullptr, pb = nullptr;
  if (e->profiling) {
    pa = get_event(e); pb = get_event(e);
    hipEventRecord(pa, e->stream);
  }
  const int rc = gemm_f32_dual(a, w, e->stream);
  if (rc == -1) return 1;  // (an unused event pair stays in the pool)
  if (rc != 0) return fail(rc, "gemm_f32_dual launch failed: %s", hipGetErrorString((hipError_t)rc));
  if (e->profiling) {
    hipEventRecord(pb, e->stream);
    ProfRec r;
    r.family = KF_GEMM_DUAL; r.flops = flops; r.bytes = bytes; r.a = pa; r.b = pb;
    e->prof.push_back(r);
  }
  return 0;
}

void free_activations(tfk_engine* e, bool async = false) {
  auto fr = [&](float*& p) { dev_free(e, p, async); p = nullptr; };
  for (int s = 0; s < 2; ++s) {
    fr(e->dX[s]);
    dev_free(e, e->dY[s], async); e->dY[s] = nullptr;
    fr(e->dA[s]);
    host_retire(e, e->hX[s], async); e->hX[s] = nullptr;
    host_retire(e, e->hY[s], async); e->hY[s] = nullptr;
    fr(e->dRaw[s]);
    dev_free(e, e->dSeg[s], async); e->dSeg[s] = nullptr;
    host_retire(e, e->hRaw[s], async); e->hRaw[s] = nullptr;
    host_retire(e, e->hSeg[s], async); e->hSeg[s] = nullptr;
    fr(e->dCmvn[s]);
    host_retire(e, e->hCmvn[s], async); e->hCmvn[s] = nullptr;
  }
  e->cmvn_cap = 0;
  auto frb = [&](bf16_t*& p) { dev_free(e, p, async); p = nullptr; };
  for (int s = 0; s < 2; ++s) { frb(e->Xb[s]); frb(e->dAb[s]); }
  for (auto& p : e->ab) frb(p);
  frb(e->logb);
  for (auto& p : e->z) fr(p);
  for (auto& p : e->a) fr(p);
  for (auto& p : e->v) fr(p);
  for (auto& p : e->rowscale) fr(p);
  fr(e->logits); fr(e->post); fr(e->row_loss); fr(e->ws); fr(e->ws_bwd); fr(e->ws_stats);
  e->cap = 0;
}

// synchronous allocations of create (small, once)
int alloc_zero(float** p, size_t floats) {
  HIPCHK(hipMalloc((void**)p, floats * sizeof(float)));
  HIPCHK(hipMemset(*p, 0, floats * sizeof(float)));
  return 0;
}
int alloc_zero_b(bf16_t** p, size_t elems) {
  HIPCHK(hipMalloc((void**)p, elems * sizeof(bf16_t)));
  HIPCHK(hipMemset(*p, 0, elems * sizeof(bf16_t)));
  return 0;
}
// stream-ordered, zero-filled
int grow_zero(tfk_engine* e, float** p, size_t floats) { return dev_alloc(e, (void**)p, floats * sizeof(float), true); }
int grow_zero_b(tfk_engine* e, bf16_t** p, size_t elems) { return dev_alloc(e, (void**)p, elems * sizeof(bf16_t), true); }

inline size_t seg_ints(int cap) { return (size_t)2 * cap + cap / 64 + 8; }

int reserve(tfk_engine* e, int T) {
  if (T <= e->cap) return 0;
  int cap = e->cap + e->cap / 2;
  if (cap < T) cap = T;
  cap = (int)up(cap, 64);
  // The pinned staging buffers of the last two micro-batches may still be the source of their H2D copies: wait for
  // those copies (the copy stream only -- compute keeps running); everything on the device is released and
  // re-allocated in stream order behind the kernels already enqueued, without a host or device synchronisation.
  for (int s = 0; s < 2; ++s)
    if (e->slot_used[s]) HIPCHK(hipEventSynchronize(e->copy_done[s]));
  free_activations(e, /*async=*/true);
  const int L = e->L;
  for (int s = 0; s < 2; ++s) {
    CHK(grow_zero(e, &e->dX[s], (size_t)cap * e->ldF));
    CHK(dev_alloc(e, (void**)&e->dY[s], (size_t)cap * sizeof(int32_t), false));
    CHK(grow_zero(e, &e->dA[s], (size_t)cap * e->ldH));
    HIPCHK(hipHostMalloc((void**)&e->hX[s], (size_t)cap * e->F * sizeof(float), hipHostMallocDefault));
    // (zeroed: a stacked pass copies whole padded slots to the device, and a padding row of the host buffer that happened to
    //  hold a NaN bit pattern would poison dW = X^T dZ although its dZ row is zero -- pinned pages are not cleared by the driver)
    memset(e->hX[s], 0, (size_t)cap * e->F * sizeof(float));
    HIPCHK(hipHostMalloc((void**)&e->hY[s], (size_t)cap * sizeof(int32_t), hipHostMallocDefault));
    // raw frames: at most F columns (context 0); one utterance per frame at worst
    CHK(grow_zero(e, &e->dRaw[s], (size_t)cap * e->ldF));
    // utterance offsets [U + 1]; stacked passes add the utterances' output rows [U] and the row_vend table [cap / 64 + 1]
    CHK(dev_alloc(e, (void**)&e->dSeg[s], seg_ints(cap) * sizeof(int32_t), false));
    HIPCHK(hipHostMalloc((void**)&e->hRaw[s], (size_t)cap * e->F * sizeof(float), hipHostMallocDefault));
    memset(e->hRaw[s], 0, (size_t)cap * e->F * sizeof(float));
    HIPCHK(hipHostMalloc((void**)&e->hSeg[s], seg_ints(cap) * sizeof(int32_t), hipHostMallocDefault));
    e->slot_used[s] = false;
  }
  e->z.assign(L, nullptr); e->a.assign(L, nullptr); e->v.assign(L, nullptr); e->rowscale.assign(L, nullptr);
  for (int l = 0; l < L; ++l) {
    CHK(grow_zero(e, &e->z[l], (size_t)cap * e->ldH));
    CHK(grow_zero(e, &e->a[l], (size_t)cap * e->ldH));
    if (e->cfg.l2_norm) {
      CHK(grow_zero(e, &e->v[l], (size_t)cap * e->ldH));
      CHK(grow_zero(e, &e->rowscale[l], (size_t)cap));
    }
  }
  if (e->bf16) {
    for (int s = 0; s < 2; ++s) {
      CHK(grow_zero_b(e, &e->Xb[s], e->tw_elems(cap, e->ldFb)));
      CHK(grow_zero_b(e, &e->dAb[s], e->tw_elems(cap, e->ldHb)));
    }
    e->ab.assign(L, nullptr);
    for (int l = 0; l < L; ++l) CHK(grow_zero_b(e, &e->ab[l], e->tw_elems(cap, e->ldHb)));
    CHK(grow_zero_b(e, &e->logb, e->tw_elems(cap, e->ldOb)));
  }
  CHK(grow_zero(e, &e->logits, (size_t)cap * e->ldO));
  CHK(grow_zero(e, &e->post, (size_t)cap * e->ldO));
  CHK(grow_zero(e, &e->row_loss, (size_t)cap));
  const int ldmax = e->ldH > e->ldO ? e->ldH : e->ldO;
  CHK(grow_zero(e, &e->ws, (size_t)3 * kMaxRowSplits * ldmax));
  CHK(grow_zero(e, &e->ws_stats, (size_t)2 * ((cap + 63) / 64) * e->ldH));
  e->ws_bwd_stride = (size_t)3 * kMaxRowSplits * ldmax;
  CHK(grow_zero(e, &e->ws_bwd, e->ws_bwd_stride * (L + 1)));
  // the copy stream writes the new input slots: not before their allocation + zero fill in engine-stream order
  HIPCHK(hipEventRecord(e->ev_grow, e->stream));
  HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_grow, 0));
  e->cap = cap;
  return 0;
}

// ---- stacked passes: several micro-batches of one optimiser step behind each other in ONE pass of the GEMMs ----
// The reference runs update_gradients_op once per micro-batch (trainer.py:310-332): the contractions of a micro-batch of
// a thousand frames leave the matrix pipes half idle (one tile per CU, weights re-read k times per step).  A stacked pass
// multiplies all k micro-batches ("segments") at once -- rows are independent in the affine maps, and dW over the stacked
// rows IS G += g -- while everything that couples the rows of a micro-batch stays per segment: batch-norm statistics and
// their moving-average updates (in segment order), BN backward, the dropout stream (call index + row inside the segment).
// Segment i starts at row r0[i], a multiple of every GEMM tile height, holds rows[i] frames and is followed by padding
// up to span[i] rows: padding carries label -1 (zero loss, zero gradient rows) and finite activations, and is excluded
// from the statistics by the row_vend table (gemm_f32.h).
constexpr int kMaxStack = 32;
struct Stack {
  int k = 0;
  int r0[kMaxStack], rows[kMaxStack], span[kMaxStack];
  int T_pad = 0, T_valid = 0;
  const int* d_vend = nullptr;  // device: row_vend table of the pass
};

// Bring one micro-batch to HBM (or adopt device pointers).  Returns the GEMM-ready X (ld in *ldx_out).
int wait_slot_free(tfk_engine* e, int s);  // (below, next to finish_slot)
int stage_input(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int T, int flags, const float** Xd,
                int* ldx_out, const int32_t** yd) {
  if (ldx < e->F) return fail(-1, "ldx %lld < input_dim %d", (long long)ldx, e->F);
  if (flags & TFK_DEVICE_PTRS) {
    if ((ldx % 4) == 0 && (e->F % 4) == 0 && (((uintptr_t)X) % 16) == 0) {
      *Xd = X;
      *ldx_out = (int)ldx;
    } else {
      const int s = e->slot;
      HIPCHK(hipMemcpy2DAsync(e->dX[s], (size_t)e->ldF * 4, X, (size_t)ldx * 4, (size_t)e->F * 4, T,
                              hipMemcpyDeviceToDevice, e->stream));
      *Xd = e->dX[s];
      *ldx_out = e->ldF;
      e->slot ^= 1;
    }
    *yd = y;
    return 0;
  }
  const int s = e->slot;
  if (e->slot_used[s]) HIPCHK(hipEventSynchronize(e->copy_done[s]));  // pinned slot free again
  if (ldx == e->F) {
    memcpy(e->hX[s], X, (size_t)T * e->F * sizeof(float));
  } else {
    for (int t = 0; t < T; ++t) memcpy(e->hX[s] + (size_t)t * e->F, X + (size_t)t * ldx, (size_t)e->F * sizeof(float));
  }
  if (y) memcpy(e->hY[s], y, (size_t)T * sizeof(int32_t));
  // the device slot may still be read by the compute enqueued two calls ago
  CHK(wait_slot_free(e, s));
  HIPCHK(hipMemcpy2DAsync(e->dX[s], (size_t)e->ldF * 4, e->hX[s], (size_t)e->F * 4, (size_t)e->F * 4, T,
                          hipMemcpyHostToDevice, e->copy_stream));
  if (y) HIPCHK(hipMemcpyAsync(e->dY[s], e->hY[s], (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(hipEventRecord(e->copy_done[s], e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done[s], 0));
  e->slot_used[s] = true;
  *Xd = e->dX[s];
  *ldx_out = e->ldF;
  *yd = e->dY[s];
  e->slot ^= 1;
  return 0;
}

// Bring unspliced frames + utterance offsets to HBM and splice them into dX[slot] there.
// `st` (stacked pass): the utterances form st->k segments of seg_utts[.] utterances each; segment i's spliced rows start at
// row st->r0[i] of the slot, the rows in between are padding (label -1), and the row_vend table of the pass is staged with
// the utterance offsets (st->d_vend is set).
int stage_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int T, const int32_t* utt_len, int U,
              int context, const float* cmvn, const float** Xd, int* ldx_out, const int32_t** yd, bool raw_on_device = false,
              Stack* st = nullptr, const int32_t* seg_utts = nullptr) {
  if (context < 0) return fail(-1, "context_width %d < 0", context);
  const int win = 2 * context + 1;
  if (e->F % win != 0) return fail(-1, "input_dim %d is not a multiple of 2*context_width+1 = %d", e->F, win);
  const int D = e->F / win;
  if (ldraw < D) return fail(-1, "ldraw %lld < raw dimension %d", (long long)ldraw, D);
  if (U <= 0 || !utt_len) return fail(-1, "no utterances");
  if (U > T) return fail(-1, "more utterances (%d) than frames (%d)", U, T);
  long total = 0;
  for (int u = 0; u < U; ++u) {
    if (utt_len[u] < 0) return fail(-1, "negative utterance length");
    total += utt_len[u];
  }
  if (total != T) return fail(-1, "utterance lengths sum to %ld, expected T = %d", total, T);
  const int s = e->slot;
  const size_t cmvn_floats = cmvn ? (size_t)2 * U * D : 0;
  if (cmvn_floats > e->cmvn_cap) {
    // the tables of the last two micro-batches: their H2D copies (copy stream) must be over before the pinned side
    // is retired; the device side is released / re-allocated in engine-stream order (the splice kernels that read it
    // are on that stream, and the copies were awaited by it)
    for (int k = 0; k < 2; ++k)
      if (e->slot_used[k]) HIPCHK(hipEventSynchronize(e->copy_done[k]));
    for (int k = 0; k < 2; ++k) {
      dev_free(e, e->dCmvn[k], true); e->dCmvn[k] = nullptr;
      host_retire(e, e->hCmvn[k], true); e->hCmvn[k] = nullptr;
      CHK(dev_alloc(e, (void**)&e->dCmvn[k], 2 * cmvn_floats * sizeof(float), false));
      HIPCHK(hipHostMalloc((void**)&e->hCmvn[k], 2 * cmvn_floats * sizeof(float), hipHostMallocDefault));
    }
    HIPCHK(hipEventRecord(e->ev_grow, e->stream));
    HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_grow, 0));
    e->cmvn_cap = 2 * cmvn_floats;
  }
  if (e->slot_used[s]) HIPCHK(hipEventSynchronize(e->copy_done[s]));
  if (!raw_on_device) {
    if (ldraw == D) memcpy(e->hRaw[s], raw, (size_t)T * D * sizeof(float));
    else for (int t = 0; t < T; ++t) memcpy(e->hRaw[s] + (size_t)t * D, raw + (size_t)t * ldraw, (size_t)D * sizeof(float));
  }
  if (cmvn) memcpy(e->hCmvn[s], cmvn, cmvn_floats * sizeof(float));
  e->hSeg[s][0] = 0;
  for (int u = 0; u < U; ++u) e->hSeg[s][u + 1] = e->hSeg[s][u] + utt_len[u];
  size_t seg_words = (size_t)U + 1;
  int y_rows = T;
  if (st) {
    int32_t* oseg = e->hSeg[s] + U + 1;
    int32_t* vend = oseg + U;
    int u = 0;
    for (int i = 0; i < st->k; ++i) {
      int row = st->r0[i];
      for (int j = 0; j < seg_utts[i]; ++j, ++u) {
        oseg[u] = row;
        row += utt_len[u];
      }
      for (int unit = st->r0[i] / 64; unit < (st->r0[i] + st->span[i]) / 64; ++unit) vend[unit] = st->r0[i] + st->rows[i];
    }
    seg_words += (size_t)U + st->T_pad / 64;
    y_rows = st->T_pad;
    if (y) {
      for (int t = 0; t < st->T_pad; ++t) e->hY[s][t] = -1;
      for (int i = 0, t0 = 0; i < st->k; t0 += st->rows[i], ++i)
        memcpy(e->hY[s] + st->r0[i], y + t0, (size_t)st->rows[i] * sizeof(int32_t));
    }
  } else if (y) {
    memcpy(e->hY[s], y, (size_t)T * sizeof(int32_t));
  }
  CHK(wait_slot_free(e, s));
  const int ldD = (D + 3) & ~3;
  if (!raw_on_device)
    HIPCHK(hipMemcpy2DAsync(e->dRaw[s], (size_t)ldD * 4, e->hRaw[s], (size_t)D * 4, (size_t)D * 4, T,
                            hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(hipMemcpyAsync(e->dSeg[s], e->hSeg[s], seg_words * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
  if (y) HIPCHK(hipMemcpyAsync(e->dY[s], e->hY[s], (size_t)y_rows * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
  if (cmvn)
    HIPCHK(hipMemcpyAsync(e->dCmvn[s], e->hCmvn[s], cmvn_floats * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(hipEventRecord(e->copy_done[s], e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done[s], 0));
  {
    ProfScope ps(e, KF_MISC, 0, 4.0 * T * (D + e->F));
    // (TFK_RAW_DEVICE: the caller's matrix is spliced where it lies -- the kernel reads rows of any leading dimension)
    splice_frames(e->stream, raw_on_device ? raw : e->dRaw[s], raw_on_device ? (int)ldraw : ldD, e->dSeg[s], U, T, D, context,
                  cmvn ? e->dCmvn[s] : nullptr, e->dX[s], e->ldF, st ? e->dSeg[s] + U + 1 : nullptr);
  }
  if (st) st->d_vend = e->dSeg[s] + 2 * U + 1;
  e->slot_used[s] = true;
  *Xd = e->dX[s];
  *ldx_out = e->ldF;
  *yd = e->dY[s];
  e->slot ^= 1;
  return 0;
}

// Mixed precision: the bf16 twin of the staged input.  A caller-owned device matrix adopted in place has no
// slot of its own: it is converted into the current slot's twin and that slot's (unused) fp32 buffer becomes
// the key under which the GEMMs find it.
int twin_input(tfk_engine* e, const float** Xd, int* ld, int T) {
  int s = (*Xd == e->dX[0]) ? 0 : (*Xd == e->dX[1]) ? 1 : -1;
  const float* src = *Xd;
  const int ld_src = *ld;
  if (s < 0) {
    s = e->slot;
    *Xd = e->dX[s];
    *ld = e->ldF;
  }
  to_bf16_rows(e->stream, src, ld_src, e->Xb[s], e->ldFb, T, e->F, e->x3);
  HIPCHK(hipGetLastError());
  return 0;
}

int finish_slot(tfk_engine* e, int flags, int slot_before) {
  if ((flags & TFK_DEVICE_PTRS) || e->slot == slot_before) return 0;
  if ((flags & TFK_LAST_MICROBATCH) && e->slot_lazy) {
    e->slot_step[slot_before] = e->steps_begun + 1;  // the step the coming tfk_apply_begin opens
    return 0;
  }
  e->slot_step[slot_before] = 0;
  HIPCHK(hipEventRecord(e->compute_done[slot_before], e->stream));
  return 0;
}
// the copy stream may overwrite input slot s once the compute that read it last is over
int wait_slot_free(tfk_engine* e, int s) {
  if (!e->slot_used[s]) return 0;
  if (e->slot_step[s] == 0) {
    HIPCHK(hipStreamWaitEvent(e->copy_stream, e->compute_done[s], 0));
  } else if (e->steps_seen < e->slot_step[s]) {
    // the host has not collected that step's loss (a caller that accumulates again without tfk_apply in between): the slow way
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

ActDesc act_desc(const tfk_engine* e, int layer, int train, uint32_t call) {
  ActDesc d;
  d.nonlin = e->cfg.nonlin; d.bn = e->cfg.batch_norm ? 1 : 0; d.l2 = e->cfg.l2_norm ? 1 : 0;
  d.keep = e->dropout ? e->cfg.keep_prob : 1.f;
  d.seed = e->cfg.seed; d.call = call; d.layer = (uint32_t)layer; d.train = train;
  return d;
}

// Forward through `nfw` hidden layers; logits from the output of hidden layer `nact - 1`
// (dnn.py:73-108; the tf.case of dnn.py:97-102 is the choice of nact).
int forward(tfk_engine* e, const float* Xd, int ldx, int T, int train, int nact, int nfw, uint32_t call) {
  const float* in = Xd;
  int ld_in = ldx;
  const int H = e->H, ldH = e->ldH;
  if (e->bf16 && e->shadow_dirty) {
    need_params(e, -1);  // the shadow is rebuilt from every weight matrix
    CHK(join_optimizer(e));
    CHK(refresh_shadow(e));
  }
  auto twin_a = [&](int l) { Twin t; if (e->bf16) { t.p = e->ab[l]; t.ld = e->ldHb; t.x3 = e->x3; } return t; };
  for (int l = 0; l < nfw; ++l) {
    const LayerLayout& y = e->lay[l];
    need_params(e, l);
    CHK(wait_layer_update(e, l, l == 0));
    if (train && e->cfg.batch_norm && !e->cfg.l2_norm) {
      // fused path: the GEMM epilogue emits the per-tile column statistics, ONE column-tiled kernel merges them
      // and applies BN + nonlinearity + dropout (4 kernels per layer -> 2)
      int cfg;
      const int chunk = gemm_chunk_rows(e, GEMM_NN, T, H, y.d_in, &cfg);
      CHK(run_gemm(e, GEMM_NN, in, ld_in, e->p_param() + y.w_off, y.ld_out, e->z[l], ldH, T, H, y.d_in,
                   e->p_param() + y.b_off, EPI_BIAS | EPI_COLSTATS, nullptr, e->ws_stats, cfg));
      {
        ProfScope ps(e, KF_ACT_FWD, 0, 8.0 * T * H);
        const ActDesc d = act_desc(e, l, train, call);
        const int nchunk = (T + chunk - 1) / chunk;
        if (nchunk > kMergeOnceChunks && nchunk <= kMaxRowSplits) {
          // tall micro-batch: the statistics are merged once, the row-wise kernel applies them
          bn_stats_from_chunks(e->stream, e->ws_stats, chunk, T, H, ldH, e->bn_eps, e->bn_decay, e->mean[l], e->rstd[l],
                               e->ema_mean(l), e->ema_var(l));
          act_forward(e->stream, d, e->z[l], e->a[l], nullptr, nullptr, e->mean[l], e->rstd[l],
                      e->p_param() + y.beta_off, T, H, ldH, twin_a(l));
        } else {
          bn_act_forward(e->stream, d, e->z[l], e->a[l], e->ws_stats, chunk, T, H, ldH, e->bn_eps,
                         e->bn_decay, e->mean[l], e->rstd[l], e->ema_mean(l), e->ema_var(l),
                         e->p_param() + y.beta_off, twin_a(l));
        }
      }
      in = e->a[l];
      ld_in = ldH;
      continue;
    }
    if (!train && !e->cfg.l2_norm && e->fuse_eval) {
      // evaluation mode (decoder.py:36-44, trainer.py:77-79): affine, moving-statistics batch norm and the
      // nonlinearity in ONE launch -- the GEMM epilogue writes the layer output (dropout is the identity here)
      ActEpi ev = {nullptr, nullptr, e->cfg.batch_norm ? e->mov_mean(l) : nullptr,
                   e->cfg.batch_norm ? e->mov_var(l) : nullptr, e->cfg.nonlin};
      ev.beta = e->cfg.batch_norm ? e->p_param() + y.beta_off : nullptr;
      ev.eps = e->bn_eps;
      if (e->bf16) { ev.twin = e->ab[l]; ev.ld_twin = e->ldHb; }
      CHK(run_gemm(e, GEMM_NN, in, ld_in, e->p_param() + y.w_off, y.ld_out, e->a[l], ldH, T, H, y.d_in,
                   e->p_param() + y.b_off, EPI_BIAS | EPI_EVAL_ACT, nullptr, nullptr, -1, &ev));
      in = e->a[l];
      ld_in = ldH;
      continue;
    }
    CHK(run_gemm(e, GEMM_NN, in, ld_in, e->p_param() + y.w_off, y.ld_out, e->z[l], ldH, T, H, y.d_in,
                 e->p_param() + y.b_off, EPI_BIAS));
    if (e->cfg.batch_norm) {
      ProfScope ps(e, KF_BN_STATS, 0, 8.0 * T * H);
      if (train)
        bn_stats_train(e->stream, e->z[l], T, H, ldH, e->bn_eps, e->bn_decay, e->mean[l], e->rstd[l], e->ema_mean(l),
                       e->ema_var(l), e->ws);
      else
        bn_stats_eval(e->stream, e->mov_mean(l), e->mov_var(l), H, e->bn_eps, e->mean[l], e->rstd[l]);
    }
    {
      ProfScope ps(e, KF_ACT_FWD, 0, 8.0 * T * H);
      const ActDesc d = act_desc(e, l, train, call);
      act_forward(e->stream, d, e->z[l], e->a[l], e->cfg.l2_norm ? e->v[l] : nullptr,
                  e->cfg.l2_norm ? e->rowscale[l] : nullptr, e->mean[l], e->rstd[l],
                  e->cfg.batch_norm ? e->p_param() + y.beta_off : nullptr, T, H, ldH, twin_a(l));
    }
    in = e->a[l];
    ld_in = ldH;
  }
  const LayerLayout& o = e->lay[e->L];
  need_params(e, e->L);
  CHK(wait_layer_update(e, e->L, nfw == 0));
  CHK(run_gemm(e, GEMM_NN, e->a[nact - 1], ldH, e->p_param() + o.w_off, o.ld_out, e->logits, e->ldO, T, e->O, H,
               e->p_param() + o.b_off, EPI_BIAS));
  HIPCHK(hipGetLastError());
  return 0;
}

int backward(tfk_engine* e, const float* Xd, int ldx, int T, int nact, uint32_t call, bool fire) {
  const int L = e->L, H = e->H, ldH = e->ldH;
  const LayerLayout& o = e->lay[L];
  float* G = e->p_grad();
  const int acc = e->grads_fresh ? 0 : 1;
  const int epi_w = acc ? EPI_ACCUM : 0;
  if (!acc)  // layers above the active depth are not visited below: their (logically zero) G must be zero
    for (int l = nact; l < L; ++l) {
      const LayerLayout& q = e->lay[l];
      HIPCHK(hipMemsetAsync(G + q.w_off, 0, q.w_sz * sizeof(float), e->stream));
      HIPCHK(hipMemsetAsync(G + q.b_off, 0, q.b_sz * sizeof(float), e->stream));
      if (q.beta_sz) HIPCHK(hipMemsetAsync(G + q.beta_off, 0, q.beta_sz * sizeof(float), e->stream));
    }
  // output layer: dZ = softmax - onehot sits in `logits`
  // (when eligible the output layer's dW runs in one launch with the dA that follows: see below)
  bool out_dw_done = false;
  FinalBatch fin;
  fin.n = 0;
  fin.accumulate = acc;
  const int rs = row_splits(T);
  auto ws_of = [&](int l) { return e->ws_bwd + (size_t)l * e->ws_bwd_stride; };
  {
    if (!e->colsum_done) {  // (CTC: the loss has a reduction of its own)
      ProfScope ps(e, KF_COLSUM, 0, 4.0 * T * e->O);
      colsum_partial(e->stream, e->logits, T, e->ldO, ws_of(L));
    }
    e->colsum_done = false;
    fin.it[fin.n++] = {ws_of(L), G + o.b_off, 0, rs, e->O, e->ldO};
  }
  int pp = 0;
  // BN chains without L2Norm / dropout: the GEMM that produces a layer's output gradient also applies f' and
  // reduces the two column sums of batch-norm's backward in its epilogue (EPI_DACT), which removes one pass
  // over [T, H] per layer.  The per-tile partial sums must fit the kMaxRowSplits chunk slots of the workspace.
  int cfg_h, cfg_o;
  const int rows_h = gemm_chunk_rows(e, GEMM_NT, T, H, H, &cfg_h), rows_o = gemm_chunk_rows(e, GEMM_NT, T, H, e->O, &cfg_o);
  const int chunks_h = (T + rows_h - 1) / rows_h, chunks_o = (T + rows_o - 1) / rows_o;
  // Dropout behind a ReLU fuses as well: a = relu(u) * mask / keep, so d a / d u = (a > 0) / keep exactly -- the
  // epilogue reads it off the stored layer output without regenerating the mask.  (Behind sigmoid / tanh the kept
  // value would have to be un-scaled first; those chains keep the separate pass.)
  const bool drop = e->cfg.keep_prob < 1.f;
  const float dscale = drop ? 1.f / e->cfg.keep_prob : 1.f;
  const bool fuse_hb = e->cfg.batch_norm && !e->cfg.l2_norm && (!drop || e->cfg.nonlin == TFK_NONLIN_RELU) &&
                       e->fuse_hb_enabled && chunks_h <= kMaxRowSplits && chunks_o <= kMaxRowSplits;
  int chunks_first = chunks_o;  // row chunks of the EPI_DACT partial sums of the GEMM that produced the current `da`
  auto dact_gemm = [&](const float* dz, int ld_dz, const float* W, int ldw, float* out, int K, int target, int cfg) {
    if (!fuse_hb) return run_gemm(e, GEMM_NT, dz, ld_dz, W, ldw, out, ldH, T, H, K, nullptr, 0);
    ActEpi act = {e->a[target], e->z[target], e->mean[target], e->rstd[target], e->cfg.nonlin};
    act.scale = dscale;
    act.beta = e->p_param() + e->lay[target].beta_off;
    return run_gemm(e, GEMM_NT, dz, ld_dz, W, ldw, out, ldH, T, H, K, nullptr, EPI_DACT, nullptr, ws_of(target), cfg,
                    &act);
  };
  {
    ActEpi act = {e->a[nact - 1], e->z[nact - 1], e->mean[nact - 1], e->rstd[nact - 1], e->cfg.nonlin};
    act.scale = dscale;
    act.beta = e->cfg.batch_norm ? e->p_param() + e->lay[nact - 1].beta_off : nullptr;
    int bm = rows_o;
    const int rc = run_gemm_dual(e, e->logits, e->ldO, e->p_param() + o.w_off, o.ld_out, e->dA[pp], ldH, T, H, e->O,
                                 fuse_hb ? &act : nullptr, fuse_hb ? ws_of(nact - 1) : nullptr, e->a[nact - 1], ldH,
                                 G + o.w_off, o.ld_out, H, e->O, epi_w, &bm);
    if (rc < 0) return rc;
    out_dw_done = rc == 0;
    if (out_dw_done) chunks_first = (T + bm - 1) / bm;
  }
  if (!out_dw_done) {
    CHK(run_gemm(e, GEMM_TN, e->a[nact - 1], ldH, e->logits, e->ldO, G + o.w_off, o.ld_out, H, e->O, T, nullptr,
                 epi_w));
    CHK(dact_gemm(e->logits, e->ldO, e->p_param() + o.w_off, o.ld_out, e->dA[pp], e->O, nact - 1, cfg_o));
  }
  if (fire && e->cb) {
    e->cb(e->cb_user, 0);  // the output layer's weight gradient is enqueued
    // Layers above the active depth (layer-wise growth) receive no gradient -- the zero branch of the tf.case,
    // dnn.py:97-104; their G is zero already.  Their buckets are announced HERE, so that the order is 0, 1, .. L
    // whatever the active depth is: a rank without micro-batches replays exactly that order (dataparallel.py).
    for (int l = L - 1; l >= nact; --l) e->cb(e->cb_user, L - l);
  }
  int chunks_in = chunks_first;
  for (int l = nact - 1; l >= 0; --l) {
    const LayerLayout& y = e->lay[l];
    float* da = e->dA[pp];
    const ActDesc d = act_desc(e, l, 1, call);
    int pre_du = fuse_hb ? 1 : 0;
    if (e->cfg.l2_norm) {
      ProfScope ps(e, KF_HIDDEN_BWD, 0, 12.0 * T * H);
      act_backward_rows(e->stream, d, da, e->v[l], e->rowscale[l], T, H, ldH);
      pre_du = 1;
    }
    {
      ProfScope ps(e, KF_HIDDEN_BWD, 0, (e->cfg.batch_norm ? 28.0 : 12.0) * T * H);
      Twin tw;
      if (e->bf16) { tw.p = e->dAb[pp]; tw.ld = e->ldHb; tw.x3 = e->x3; }
      int chunks_eff = fuse_hb ? chunks_in : 0;
      if (chunks_eff > kMergeOnceChunks) {  // tall micro-batch: reduce the EPI_DACT partial sums once
        chunk_totals(e->stream, ws_of(l), chunks_eff, ldH);
        chunks_eff = 1;
      }
      hidden_backward(e->stream, d, pre_du, da, e->a[l], e->z[l], e->mean[l], e->rstd[l], T, H, ldH, ws_of(l),
                      chunks_eff, tw);
      if (e->cfg.batch_norm) fin.it[fin.n++] = {ws_of(l), G + y.beta_off, 0, fuse_hb ? chunks_eff : rs, H, ldH};
      fin.it[fin.n++] = {ws_of(l), G + y.b_off, 2, rs, H, ldH};
      if (fin.n + 2 > kMaxFinalItems) {  // very deep nets: flush
        grad_final(e->stream, fin);
        fin.n = 0;
      }
    }
    const float* in = l == 0 ? Xd : e->a[l - 1];
    const int ld_in = l == 0 ? ldx : ldH;
    bool fused = false;
    int chunks_next = chunks_h;
    if (l > 0) {  // dW_l and the dA that feeds layer l - 1 both read dz_l: one launch when eligible
      ActEpi act = {e->a[l - 1], e->z[l - 1], e->mean[l - 1], e->rstd[l - 1], e->cfg.nonlin};
      act.scale = dscale;
      act.beta = e->cfg.batch_norm ? e->p_param() + e->lay[l - 1].beta_off : nullptr;
      int bm = rows_h;
      const int rc = run_gemm_dual(e, da, ldH, e->p_param() + y.w_off, y.ld_out, e->dA[pp ^ 1], ldH, T, H, H,
                                   fuse_hb ? &act : nullptr, fuse_hb ? ws_of(l - 1) : nullptr, in, ld_in, G + y.w_off,
                                   y.ld_out, y.d_in, H, epi_w, &bm);
      if (rc < 0) return rc;
      fused = rc == 0;
      chunks_next = (T + bm - 1) / bm;
    }
    if (!fused) {
      CHK(run_gemm(e, GEMM_TN, in, ld_in, da, ldH, G + y.w_off, y.ld_out, y.d_in, H, T, nullptr, epi_w));
      if (l > 0) CHK(dact_gemm(da, ldH, e->p_param() + y.w_off, y.ld_out, e->dA[pp ^ 1], H, l - 1, cfg_h));
    }
    if (l > 0) chunks_in = fused ? chunks_next : chunks_h;
    if (fire && e->cb) e->cb(e->cb_user, L - l);
    pp ^= 1;
  }
  {  // one kernel turns every layer's partial column sums into the bias / beta gradient sums
    ProfScope ps(e, KF_COLSUM, 0, 0);
    grad_final(e->stream, fin);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// materialise the logical zeros of the scalar accumulators (only when something reads them before a micro-batch)
// A kernel reported a failure the host could not see at launch time (today: a split-K block of the fp32-emulating
// contraction whose partner's partial sums never arrived, gemm_bf16.h): the word sits in mapped pinned memory, so reading it
// behind a synchronisation costs nothing.  Called wherever the engine hands results to the host.
int check_kernel_errors(tfk_engine* e) {
  volatile unsigned* w = reinterpret_cast<volatile unsigned*>(e->h_scalars + kErrWord);
  if (*w == 0) return 0;
  *w = 0;
  if (e->ws_splitk)  // (flag words of the interrupted exchange may be left set: start the next launch from zeros)
    (void)hipMemsetAsync(e->ws_splitk, 0, e->ws_splitk_floats * sizeof(float), e->stream);
  return fail(-1, "split-K exchange timed out: a block of the fp32-emulating contraction waited ~1 s for its partner's partial "
                  "sums (workspace not zeroed, or the partner block failed); the results of this step are invalid");
}

int settle_scalars(tfk_engine* e) {
  if (e->scalars_fresh) {
    HIPCHK(hipMemsetAsync(e->p_scalars(), 0, kScalarFloats * sizeof(float), e->stream));
    e->scalars_fresh = false;
  }
  return 0;
}
int read_scalars(tfk_engine* e) {
  CHK(settle_scalars(e));
  HIPCHK(hipMemcpyAsync(e->h_scalars, e->p_scalars(), 4 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->scalars_fresh = true;  // init_loss / init_num_frames
  return check_kernel_errors(e);
}

struct TensorRef {
  float* ptr;
  int rows, cols, ld;
};
int tensor_ref(tfk_engine* e, int kind, int slot, int layer, TensorRef* t) {
  const int L = e->L;
  if (kind == TFK_WEIGHTS || kind == TFK_BIASES) {
    if (layer < 0 || layer > L) return fail(-1, "layer %d out of range [0, %d]", layer, L);
  } else if (kind >= TFK_BN_BETA && kind <= TFK_BN_MOVING_VAR) {
    if (layer < 0 || layer >= L) return fail(-1, "layer %d out of range [0, %d)", layer, L);
    if (!e->cfg.batch_norm) return fail(-1, "batch norm tensors requested but batch_norm is off");
  } else {
    return fail(-1, "unknown tensor kind %d", kind);
  }
  float* base;
  switch (slot) {
    case TFK_SLOT_PARAM: base = e->p_param(); break;
    case TFK_SLOT_GRAD: base = e->p_grad(); break;
    case TFK_SLOT_ADAM_M: base = e->p_m(); break;
    case TFK_SLOT_ADAM_V: base = e->p_v(); break;
    default: return fail(-1, "unknown tensor slot %d", slot);
  }
  const LayerLayout& y = e->lay[layer < 0 ? 0 : layer];
  switch (kind) {
    case TFK_WEIGHTS: *t = {base + y.w_off, y.d_in, y.d_out, y.ld_out}; break;
    case TFK_BIASES: *t = {base + y.b_off, 1, y.d_out, y.ld_out}; break;
    case TFK_BN_BETA: *t = {base + y.beta_off, 1, e->H, e->ldH}; break;
    case TFK_BN_MOVING_MEAN:
    case TFK_BN_MOVING_VAR:
      if (slot != TFK_SLOT_PARAM) return fail(-1, "moving statistics only have the PARAM slot");
      *t = {kind == TFK_BN_MOVING_MEAN ? e->mov_mean(layer) : e->mov_var(layer), 1, e->H, e->ldH};
      break;
  }
  return 0;
}

int create_impl(const tfk_config* cfg, void* state, size_t state_bytes, void* stream, tfk_engine** out) {
  if (!out) return fail(-1, "out is NULL");
  *out = nullptr;
  CHK(validate(cfg));
  HIPCHK(hipSetDevice(cfg->device));
  tfk_engine* e = new tfk_engine();
  e->cfg = *cfg;
  e->F = cfg->input_dim; e->L = cfg->num_layers; e->H = cfg->num_units; e->O = cfg->output_dim;
  e->ldF = (int)up(e->F, 4); e->ldH = (int)up(e->H, 4); e->ldO = (int)up(e->O, 4);
  e->bf16 = cfg->compute_dtype != TFK_DTYPE_F32;
  e->x3 = cfg->compute_dtype == TFK_DTYPE_F32X3;
  // (x3: the tiled twins consist of 2-row x 32-column units, x3_layout.h)
  const size_t ldq = e->x3 ? 32 : 8;
  e->ldFb = (int)up(e->F, ldq); e->ldHb = (int)up(e->H, ldq); e->ldOb = (int)up(e->O, ldq);
  e->bn_decay = cfg->bn_decay > 0.f ? cfg->bn_decay : 0.999f;
  e->bn_eps = cfg->bn_epsilon > 0.f ? cfg->bn_epsilon : 1e-3f;
  e->b1 = cfg->adam_beta1 > 0.f ? cfg->adam_beta1 : 0.9f;
  e->b2 = cfg->adam_beta2 > 0.f ? cfg->adam_beta2 : 0.999f;
  e->adam_eps = cfg->adam_epsilon > 0.f ? cfg->adam_epsilon : 1e-8f;
  e->dropout = cfg->keep_prob < 1.f;
  compute_layout(cfg, e->lay, e->P, e->E);
  e->state_floats = total_state_floats(e->P, e->E, shadow_floats(cfg, e->lay));
  e->off_param = 0;
  e->off_grad = e->P;
  e->off_scalars = 2 * e->P;
  e->off_ema = 2 * e->P + kScalarFloats;
  e->reduce_floats = e->P + kScalarFloats + e->E;
  e->off_m = e->off_ema + e->E;
  e->off_v = e->off_m + e->P;
  e->off_mov = e->off_v + e->P;
  e->off_shadow = e->off_mov + e->E;

  auto bail = [&](int rc) { tfk_destroy(e); return rc; };
#define HIPB(expr)                                                                                          \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return bail(fail((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_)));        \
  } while (0)
  if (stream) {
    e->stream = (hipStream_t)stream;
  } else {
    HIPB(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }
  HIPB(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  {
    const char* v;
    if ((v = getenv("TFK_FUSE_HB"))) e->fuse_hb_enabled = atoi(v) != 0;
    if ((v = getenv("TFK_POST_CHUNK"))) e->post_chunk = atoi(v) > 0 ? (int)up((size_t)atoi(v), 64) : 0;
    if ((v = getenv("TFK_DUAL_GEMM"))) e->dual_gemm = atoi(v) != 0;
    if ((v = getenv("TFK_STACK"))) e->stack_enabled = atoi(v) != 0;
    if ((v = getenv("TFK_LOSS_EVENT"))) e->loss_event = atoi(v) != 0;
    if ((v = getenv("TFK_SLOT_EVENT"))) e->slot_lazy = atoi(v) == 0;
    if ((v = getenv("TFK_FUSE_EVAL"))) e->fuse_eval = atoi(v) != 0;
  }
  {
    // Measured (profiles/r03_ablation.txt) and NOT adopted as the default: the optimiser's 28 B per parameter and the forward
    // contractions share the L2 -> CU path (the forward GEMMs of the 1024-frame configurations are bound by exactly that
    // fill), so running them side by side slows both -- cfg2 fp32 1.480 -> 1.522 ms, cfg3/GPU bf16 0.591 -> 0.636 ms; only at
    // cfg4/GPU size (MFMA-bound 256x128 tiles) it gains, 3.02 -> 2.94 ms.  TFK_ADAM_OVERLAP=1 switches it on.
    const char* v = getenv("TFK_ADAM_OVERLAP");
    e->opt_overlap = v ? atoi(v) != 0 : false;
  }
  if (e->opt_overlap) {
    HIPB(hipStreamCreateWithFlags(&e->opt_stream, hipStreamNonBlocking));
    HIPB(hipEventCreateWithFlags(&e->ev_opt_begin, hipEventDisableTiming));
    HIPB(hipEventCreateWithFlags(&e->ev_opt_vec, hipEventDisableTiming));
    e->ev_adam.assign(e->L + 1, nullptr);
    for (int l = 0; l <= e->L; ++l) HIPB(hipEventCreateWithFlags(&e->ev_adam[l], hipEventDisableTiming));
  }
  HIPB(hipMalloc((void**)&e->d_snap, 16 * sizeof(float)));
  HIPB(hipEventCreateWithFlags(&e->ev_loss, hipEventDisableTiming));
  HIPB(hipEventCreateWithFlags(&e->ev_grow, hipEventDisableTiming));
  for (int s = 0; s < 2; ++s) {
    HIPB(hipEventCreateWithFlags(&e->copy_done[s], hipEventDisableTiming));
    HIPB(hipEventCreateWithFlags(&e->compute_done[s], hipEventDisableTiming));
  }
  if (state) {
    if (state_bytes < e->state_floats * sizeof(float))
      return bail(fail(-1, "state arena too small: %zu < %zu bytes", state_bytes, e->state_floats * sizeof(float)));
    if (((uintptr_t)state) % 256) return bail(fail(-1, "state arena must be 256-byte aligned"));
    e->state = (float*)state;
  } else {
    HIPB(hipMalloc((void**)&e->state, e->state_floats * sizeof(float)));
    e->own_state = true;
  }
  HIPB(hipMemsetAsync(e->state, 0, e->state_floats * sizeof(float), e->stream));
  if (cfg->batch_norm)  // moving_variance initialises to 1, moving_mean to 0
    for (int l = 0; l < e->L; ++l) fill(e->stream, e->mov_var(l), (size_t)e->H, 1.0f);
  // (coherent = fine-grained: the host polls the sequence word while the stream is still running -- wait_loss)
  HIPB(hipHostMalloc((void**)&e->h_scalars, 16 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
  memset(e->h_scalars, 0, 16 * sizeof(float));
  HIPB(hipHostGetDevicePointer((void**)&e->h_scalars_dev, e->h_scalars, 0));
  e->mean.assign(e->L, nullptr);
  e->rstd.assign(e->L, nullptr);
  for (int l = 0; l < e->L; ++l) {
    if (alloc_zero(&e->mean[l], (size_t)e->ldH) || alloc_zero(&e->rstd[l], (size_t)e->ldH)) return bail(-1);
  }
  if (alloc_zero(&e->prior, (size_t)e->ldO)) return bail(-1);
  if (e->bf16) {
    // shadow: mirrors the fp32 arena element for element inside the state arena when every leading dimension is a
    // multiple of 8 (the optimiser then writes it with the update and the sharded exchange gathers it), else packed
    // (x3: one tiled three-plane twin per weight matrix -- x3_layout.h -- in an allocation of their own; the optimiser writes
    // them with the update through a map of the arena, `wb_map`; a data-parallel exchange gathers the fp32 parameters in that
    // mode and the twins are rebuilt from them)
    e->wb_aligned = e->x3 ? (e->L + 1 <= kShadowMapMax) : shadow_mirrors(cfg, e->lay);
    e->wb_off.assign(e->L + 1, 0);
    e->wb_ld.assign(e->L + 1, 0);
    e->wb_map.n = 0;
    size_t off = 0;
    for (int l = 0; l <= e->L; ++l) {
      const LayerLayout& y = e->lay[l];
      e->wb_ld[l] = (int)up(y.d_out, e->x3 ? 32 : 8);
      e->wb_off[l] = (e->wb_aligned && !e->x3) ? y.w_off : off;
      off += e->x3 ? up(x3::elems(y.d_in, e->wb_ld[l]), 64) : up((size_t)y.d_in * e->wb_ld[l], 64);
      if (e->x3 && e->wb_aligned) {
        ShadowMap& m = e->wb_map;
        m.begin[l] = (uint32_t)y.w_off; m.rows[l] = (uint32_t)y.d_in; m.ld[l] = (uint32_t)y.ld_out;
        m.ld_twin[l] = (uint32_t)e->wb_ld[l]; m.twin[l] = e->wb_off[l];
        m.n = l + 1;
      }
    }
    if (e->x3) {
      if (alloc_zero_b(&e->Wb, up(off, 128))) return bail(-1);
      e->own_wb = true;
    } else if (e->wb_aligned) {
      e->Wb = reinterpret_cast<bf16_t*>(e->state + e->off_shadow);  // (zeroed with the arena)
    } else {
      if (alloc_zero_b(&e->Wb, off)) return bail(-1);
      e->own_wb = true;
    }
    e->shadow_dirty = true;
  }
  const int cap0 = cfg->max_frames > 0 ? cfg->max_frames : 1024;
  { const int rc = reserve(e, cap0); if (rc) return bail(rc); }
  HIPB(hipStreamSynchronize(e->stream));
#undef HIPB
  *out = e;
  return 0;
}

struct CtcSpec {  // CTC loss instead of the frame-level cross-entropy
  const int32_t* utt_len;    // [U] frames per utterance (sum = T)
  int U;
  const int32_t* labels;     // concatenated label sequences
  const int32_t* label_len;  // [U]
};
template <class Tp>
int grow(tfk_engine* e, Tp** p, size_t* cap, size_t need) {
  if (need <= *cap) return 0;
  dev_free(e, *p, true);  // stream-ordered: kernels of the previous micro-batch may still read it
  *p = nullptr;
  *cap = 0;
  const size_t n = need + need / 2;
  CHK(dev_alloc(e, (void**)p, n * sizeof(Tp), false));
  *cap = n;
  return 0;
}
// Loss + dLogits of one CTC micro-batch (logits already in e->logits).  Replaces compute_loss of the reference's
// CTCTrainer (trainer.py:533-570); batch_loss += sum of -log p, num_frames += number of labels (trainer.py:126-133).
int ctc_loss(tfk_engine* e, const CtcSpec& c, int T, int train) {
  if (c.U <= 0 || !c.utt_len || !c.label_len) return fail(-1, "CTC: no utterances");
  std::vector<int32_t> seg(c.U + 1, 0), off(c.U + 1, 0);
  int max_labels = 0;
  for (int u = 0; u < c.U; ++u) {
    if (c.utt_len[u] < 0 || c.label_len[u] < 0) return fail(-1, "CTC: negative length");
    seg[u + 1] = seg[u] + c.utt_len[u];
    off[u + 1] = off[u] + c.label_len[u];
    max_labels = c.label_len[u] > max_labels ? c.label_len[u] : max_labels;
  }
  if (seg[c.U] != T) return fail(-1, "CTC: utterance lengths sum to %d, expected T = %d", seg[c.U], T);
  if (max_labels > kCtcMaxLabels) return fail(-1, "CTC: %d labels in one utterance (limit %d)", max_labels, kCtcMaxLabels);
  const int total = off[c.U];
  if (total > 0 && !c.labels) return fail(-1, "CTC: labels is NULL");
  for (int i = 0; i < total; ++i)  // the blank is the LAST class (tf.nn.ctc_loss): labels live in [0, O - 1)
    if (c.labels[i] < 0 || c.labels[i] >= e->O - 1) return fail(-1, "CTC: label %d outside [0, %d)", c.labels[i], e->O - 1);
  const int sext = ctc_state_stride(max_labels);
  CHK(grow(e, &e->ctc_seg, &e->ctc_cap_seg, (size_t)c.U + 1));
  CHK(grow(e, &e->ctc_lab_off, &e->ctc_cap_off, (size_t)c.U + 1));
  CHK(grow(e, &e->ctc_utt_loss, &e->ctc_cap_loss, (size_t)c.U));
  CHK(grow(e, &e->ctc_lab, &e->ctc_cap_lab, (size_t)(total > 0 ? total : 1)));
  CHK(grow(e, &e->ctc_lp, &e->ctc_cap_lp, (size_t)T * sext));
  CHK(grow(e, &e->ctc_ab, &e->ctc_cap_ab, (size_t)T * sext));
  CHK(grow(e, &e->ctc_lse, &e->ctc_cap_rows, (size_t)T));
  CHK(grow(e, &e->ctc_off, &e->ctc_cap_offrows, (size_t)T));
  CHK(grow(e, &e->ctc_logz, &e->ctc_cap_logz, (size_t)c.U));
  if (train) {
    CHK(grow(e, &e->ctc_bb, &e->ctc_cap_bb, (size_t)T * sext));
    CHK(grow(e, &e->ctc_offb, &e->ctc_cap_offb, (size_t)T));
  }
  {
    // The three small tables go through pinned staging, two buffers used in turn: the host never waits for the
    // stream here (the event guards the buffer's use two micro-batches ago).
    const int k = e->ctc_stage_slot ^= 1;
    const size_t need = 2 * ((size_t)c.U + 1) + (size_t)total;
    if (!e->ctc_staged[k]) HIPCHK(hipEventCreateWithFlags(&e->ctc_staged[k], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(e->ctc_staged[k]));
    if (need > e->h_ctc_cap[k]) {
      if (e->h_ctc[k]) HIPCHK(hipHostFree(e->h_ctc[k]));
      e->h_ctc[k] = nullptr;
      e->h_ctc_cap[k] = need + need / 2;
      HIPCHK(hipHostMalloc((void**)&e->h_ctc[k], e->h_ctc_cap[k] * sizeof(int32_t), hipHostMallocDefault));
    }
    int32_t* h = e->h_ctc[k];
    memcpy(h, seg.data(), ((size_t)c.U + 1) * sizeof(int32_t));
    memcpy(h + c.U + 1, off.data(), ((size_t)c.U + 1) * sizeof(int32_t));
    if (total > 0) memcpy(h + 2 * (c.U + 1), c.labels, (size_t)total * sizeof(int32_t));
    HIPCHK(hipMemcpyAsync(e->ctc_seg, h, (size_t)(c.U + 1) * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->ctc_lab_off, h + c.U + 1, (size_t)(c.U + 1) * sizeof(int32_t), hipMemcpyHostToDevice,
                          e->stream));
    if (total > 0)
      HIPCHK(hipMemcpyAsync(e->ctc_lab, h + 2 * (c.U + 1), (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice,
                            e->stream));
    HIPCHK(hipEventRecord(e->ctc_staged[k], e->stream));
  }
  CtcBatch b;
  b.logits = e->logits; b.ld = e->ldO; b.post = e->post; b.lse = e->ctc_lse;
  b.seg = e->ctc_seg; b.labels = e->ctc_lab; b.lab_off = e->ctc_lab_off;
  b.U = c.U; b.T = T; b.O = e->O; b.sext = sext;
  b.lp = e->ctc_lp; b.ab = e->ctc_ab; b.bb = e->ctc_bb; b.utt_loss = e->ctc_utt_loss;
  b.off = e->ctc_off; b.offb = e->ctc_offb; b.logz = e->ctc_logz;
  {
    ProfScope ps(e, KF_SOFTMAX_XENT, 0, 8.0 * T * e->O + 16.0 * T * sext);
    Twin tw;
    if (e->bf16 && train) { tw.p = e->logb; tw.ld = e->ldOb; tw.x3 = e->x3; }
    ctc_loss_grad(e->stream, b, e->logits, train, tw);
  }
  {
    ProfScope ps(e, KF_LOSS_REDUCE, 0, 4.0 * c.U);
    ctc_loss_reduce(e->stream, e->ctc_utt_loss, e->ctc_lab_off, c.U, e->p_scalars(), e->scalars_fresh);
    e->scalars_fresh = false;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

struct RawSpec {  // non-null utt_len selects the device-side splice
  const int32_t* utt_len;
  int U, context;
  const float* cmvn;  // nullable [U, 2, raw_dim]
};
int train_or_eval(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, int flags, int train,
                  const RawSpec* raw = nullptr, const CtcSpec* ctc = nullptr) {
  if (!e) return fail(-1, "engine is NULL");
  if (T <= 0) return fail(-1, "empty micro-batch (T = %d)", T);
  if (!X || (!y && !ctc)) return fail(-1, "X / y is NULL");
  if (raw && (flags & TFK_DEVICE_PTRS)) return fail(-1, "the raw entry points take host pointers (TFK_RAW_DEVICE: raw alone on the device)");
  if (!raw && (flags & TFK_RAW_DEVICE)) return fail(-1, "TFK_RAW_DEVICE belongs to the *_raw entry points");
  HIPCHK(hipSetDevice(e->cfg.device));
  // (a call that failed between the fused loss / column-sum launch and its backward pass must not leave the flag behind for a
  // later micro-batch -- a CTC one has no fused column sums: round-5 advisor finding)
  e->colsum_done = false;
  CHK(reserve(e, T));
  const float* Xd; const int32_t* yd; int ld;
  const int slot_before = e->slot;
  if (raw) CHK(stage_raw(e, X, ldx, y, T, raw->utt_len, raw->U, raw->context, raw->cmvn, &Xd, &ld, &yd, (flags & TFK_RAW_DEVICE) != 0));
  else CHK(stage_input(e, X, ldx, y, T, flags, &Xd, &ld, &yd));
  if (e->bf16) CHK(twin_input(e, &Xd, &ld, T));
  const uint32_t call = e->call_counter++;
  const int nact = e->nact();
  // train mode evaluates every hidden layer when BN is on: the UPDATE_OPS of all batch-norm layers are
  // fetched by update_gradients_op (trainer.py:164-169) even for layers the tf.case does not select.
  const int nfw = (train && e->cfg.layerwise_init && e->cfg.batch_norm) ? e->L : nact;
  CHK(forward(e, Xd, ld, T, train, nact, nfw, call));
  if (ctc) {
    CHK(ctc_loss(e, *ctc, T, train));
  } else {
    {
      ProfScope ps(e, KF_SOFTMAX_XENT, 0, (train ? 8.0 : 4.0) * T * e->O);
      Twin tw;
      if (e->bf16 && train) { tw.p = e->logb; tw.ld = e->ldOb; tw.x3 = e->x3; }
      softmax_xent(e->stream, e->logits, yd, T, e->O, e->ldO, e->row_loss, train, tw);
    }
    {
      // (folding this sum into softmax_xent -- the block that finishes last adds up the frames' losses -- was measured in
      // round 3 and is SLOWER than the second launch: 21.4 vs 9.3 + 6.3 us at cfg2, profiles/r03_fusion_experiments.txt)
      // training: ONE launch with the column sums of dLogits the backward pass starts with (kernels.hip: colsum_loss_kernel)
      ProfScope ps(e, train ? KF_COLSUM : KF_LOSS_REDUCE, 0, train ? 4.0 * T * e->O : 4.0 * T);
      if (train) {
        colsum_loss(e->stream, e->logits, T, e->ldO, e->ws_bwd + (size_t)e->L * e->ws_bwd_stride, e->row_loss, T, e->p_scalars(),
                    e->scalars_fresh);
        e->colsum_done = true;
      } else {
        loss_reduce(e->stream, e->row_loss, T, e->p_scalars(), e->scalars_fresh);
      }
      e->scalars_fresh = false;
    }
  }
  if (train) {
    const bool fire = (flags & TFK_LAST_MICROBATCH) != 0;
    if (fire) {
      // loss, frame count and the BN moving-average increments of the step are final once the last micro-batch's
      // forward + loss have run: their (tiny) bucket is announced FIRST, so that its all-reduce is long done when
      // the optimiser needs the frame count
      if (e->cfg.batch_norm && e->later_mb > 0) {
        ProfScope ps(e, KF_MISC, 0, 8.0 * e->E);
        scale_inplace(e->stream, e->p_ema(), e->E, (float)pow((double)e->bn_decay, (double)e->later_mb));
      }
      if (e->cb) e->cb(e->cb_user, e->L + 2);
    }
    CHK(backward(e, Xd, ld, T, nact, call, fire));
    if (fire && e->cb) e->cb(e->cb_user, e->L + 1);  // bias / beta gradients: finalised by backward's last kernel
    e->grads_fresh = false;
  }
  HIPCHK(hipGetLastError());
  CHK(finish_slot(e, flags, slot_before));
  e->last_T = T; e->last_nfw = nfw; e->last_call = call; e->last_in = Xd;
  return 0;
}


// ================= stacked passes (struct Stack above) =================

// the chains a stacked pass covers: batch norm + ReLU (+ dropout), cross-entropy loss, full depth, the fused BN paths on --
// BASELINE cfg2 / cfg3 / cfg4.  Everything else runs its micro-batches one after the other (same results).
bool stack_eligible(const tfk_engine* e) {
  return e->cfg.batch_norm && !e->cfg.l2_norm && e->cfg.nonlin == TFK_NONLIN_RELU && !e->cfg.layerwise_init &&
         e->fuse_hb_enabled && e->stack_enabled;
}
// segments start at multiples of the tallest GEMM tile of the arithmetic (fp32: 128, bf16: 256 rows)
int stack_align(const tfk_engine* e) { return (e->bf16 && !e->x3) ? 256 : 128; }
// rows of a stacked pass are bounded by the chunk slots of the backward workspaces: one slab-2 slot per row split (32 rows)
// of every segment, kMaxRowSplits in all
constexpr int kMaxStackRows = 8192;
// a segment this tall fills the chip on its own (and would take the merge-once statistics path): not stacked
constexpr int kMaxSegmentRows = 2048;

int seg_stats(tfk_engine* e) {
  if (!e->seg_mean.empty()) return 0;
  e->seg_mean.assign(e->L, nullptr);
  e->seg_rstd.assign(e->L, nullptr);
  for (int l = 0; l < e->L; ++l)
    if (alloc_zero(&e->seg_mean[l], (size_t)kMaxStack * e->ldH) || alloc_zero(&e->seg_rstd[l], (size_t)kMaxStack * e->ldH))
      return -1;
  return 0;
}

int forward_stacked(tfk_engine* e, const float* Xd, int ldx, const Stack& st, uint32_t call0) {
  const int H = e->H, ldH = e->ldH, T = st.T_pad;
  const float* in = Xd;
  int ld_in = ldx;
  if (e->bf16 && e->shadow_dirty) {
    need_params(e, -1);
    CHK(join_optimizer(e));
    CHK(refresh_shadow(e));
  }
  for (int l = 0; l < e->L; ++l) {
    const LayerLayout& y = e->lay[l];
    need_params(e, l);
    CHK(wait_layer_update(e, l, l == 0));
    int cfg;
    const int chunk = gemm_chunk_rows(e, GEMM_NN, T, H, y.d_in, &cfg);
    if (stack_align(e) % chunk) return fail(-1, "internal: GEMM tile of %d rows does not divide the stack alignment", chunk);
    CHK(run_gemm(e, GEMM_NN, in, ld_in, e->p_param() + y.w_off, y.ld_out, e->z[l], ldH, T, H, y.d_in,
                 e->p_param() + y.b_off, EPI_BIAS | EPI_COLSTATS, nullptr, e->ws_stats, cfg, nullptr, st.d_vend));
    const int tiles = (T + chunk - 1) / chunk;
    for (int i = 0; i < st.k; ++i) {  // statistics, moving averages (in segment order) and the activation chain per segment
      ProfScope ps(e, KF_ACT_FWD, 0, 8.0 * st.rows[i] * H);
      const ActDesc d = act_desc(e, l, 1, call0 + (uint32_t)i);
      Twin tw;
      if (e->bf16) { tw.p = e->ab[l] + e->tw_row((size_t)st.r0[i], e->ldHb); tw.ld = e->ldHb; tw.x3 = e->x3; }
      bn_act_forward(e->stream, d, e->z[l] + (size_t)st.r0[i] * ldH, e->a[l] + (size_t)st.r0[i] * ldH,
                     e->ws_stats + (size_t)(st.r0[i] / chunk) * ldH, chunk, st.rows[i], H, ldH, e->bn_eps, e->bn_decay,
                     e->seg_mean[l] + (size_t)i * ldH, e->seg_rstd[l] + (size_t)i * ldH, e->ema_mean(l), e->ema_var(l),
                     e->p_param() + y.beta_off, tw, st.span[i], tiles);
    }
    in = e->a[l];
    ld_in = ldH;
  }
  const LayerLayout& o = e->lay[e->L];
  need_params(e, e->L);
  CHK(wait_layer_update(e, e->L, false));
  CHK(run_gemm(e, GEMM_NN, e->a[e->L - 1], ldH, e->p_param() + o.w_off, o.ld_out, e->logits, e->ldO, T, e->O, H,
               e->p_param() + o.b_off, EPI_BIAS));
  HIPCHK(hipGetLastError());
  return 0;
}

int backward_stacked(tfk_engine* e, const float* Xd, int ldx, const Stack& st, uint32_t call0, bool fire) {
  const int L = e->L, H = e->H, ldH = e->ldH, T = st.T_pad;
  const LayerLayout& o = e->lay[L];
  float* G = e->p_grad();
  const int acc = e->grads_fresh ? 0 : 1;
  const int epi_w = acc ? EPI_ACCUM : 0;
  FinalBatch fin;
  fin.n = 0;
  fin.accumulate = acc;
  auto ws_of = [&](int l) { return e->ws_bwd + (size_t)l * e->ws_bwd_stride; };
  {
    if (!e->colsum_done) {
      ProfScope ps(e, KF_COLSUM, 0, 4.0 * T * e->O);
      colsum_partial(e->stream, e->logits, T, e->ldO, ws_of(L));  // (padding rows of dLogits are zero)
    }
    e->colsum_done = false;
    fin.it[fin.n++] = {ws_of(L), G + o.b_off, 0, row_splits(T), e->O, e->ldO};
  }
  int cfg_h, cfg_o;
  const int rows_h = gemm_chunk_rows(e, GEMM_NT, T, H, H, &cfg_h), rows_o = gemm_chunk_rows(e, GEMM_NT, T, H, e->O, &cfg_o);
  const float dscale = e->cfg.keep_prob < 1.f ? 1.f / e->cfg.keep_prob : 1.f;
  // the EPI_DACT operands of the layer whose output gradient a GEMM produces.  ReLU chains take the normalised
  // pre-activation from the layer output (a * keep - beta): the epilogue needs no per-segment mean / rstd.
  auto act_of = [&](int l) {
    ActEpi act = {e->a[l], e->z[l], e->seg_mean[l], e->seg_rstd[l], e->cfg.nonlin};
    act.scale = dscale;
    act.beta = e->p_param() + e->lay[l].beta_off;
    return act;
  };
  int pp = 0;
  int bm_in = rows_o;  // rows per chunk of the EPI_DACT partial sums of the GEMM that produced the current `da`
  {
    ActEpi act = act_of(L - 1);
    int bm = rows_o;
    const int rc = run_gemm_dual(e, e->logits, e->ldO, e->p_param() + o.w_off, o.ld_out, e->dA[pp], ldH, T, H, e->O, &act,
                                 ws_of(L - 1), e->a[L - 1], ldH, G + o.w_off, o.ld_out, H, e->O, epi_w, &bm);
    if (rc < 0) return rc;
    if (rc == 0) {
      bm_in = bm;
    } else {
      CHK(run_gemm(e, GEMM_TN, e->a[L - 1], ldH, e->logits, e->ldO, G + o.w_off, o.ld_out, H, e->O, T, nullptr, epi_w));
      CHK(run_gemm(e, GEMM_NT, e->logits, e->ldO, e->p_param() + o.w_off, o.ld_out, e->dA[pp], ldH, T, H, e->O, nullptr,
                   EPI_DACT, nullptr, ws_of(L - 1), cfg_o, &act));
    }
  }
  if (fire && e->cb) e->cb(e->cb_user, 0);
  for (int l = L - 1; l >= 0; --l) {
    const LayerLayout& y = e->lay[l];
    float* da = e->dA[pp];
    if (stack_align(e) % bm_in || (T + bm_in - 1) / bm_in > kMaxRowSplits)
      return fail(-1, "internal: EPI_DACT chunks of %d rows do not fit a stacked pass of %d rows", bm_in, T);
    int slot = 0;
    for (int i = 0; i < st.k; ++i) {  // BN backward per segment: its own column means, its own row count
      ProfScope ps(e, KF_HIDDEN_BWD, 0, 28.0 * st.rows[i] * H);
      const ActDesc d = act_desc(e, l, 1, call0 + (uint32_t)i);
      Twin tw;
      if (e->bf16) { tw.p = e->dAb[pp] + e->tw_row((size_t)st.r0[i], e->ldHb); tw.ld = e->ldHb; tw.x3 = e->x3; }
      const size_t r = (size_t)st.r0[i] * ldH;
      hidden_backward(e->stream, d, 1, da + r, e->a[l] + r, e->z[l] + r, e->seg_mean[l] + (size_t)i * ldH,
                      e->seg_rstd[l] + (size_t)i * ldH, st.rows[i], H, ldH, ws_of(l) + (size_t)(st.r0[i] / bm_in) * ldH,
                      (st.rows[i] + bm_in - 1) / bm_in, tw, st.span[i], ws_of(l) + ((size_t)2 * kMaxRowSplits + slot) * ldH);
      slot += row_splits(st.span[i]);
    }
    if (slot > kMaxRowSplits) return fail(-1, "internal: %d partial-sum slots in a stacked pass", slot);
    // d beta = sum of du over ALL rows (the chunks of every segment; padding chunks hold zeros), d bias = sum of dz
    fin.it[fin.n++] = {ws_of(l), G + y.beta_off, 0, (T + bm_in - 1) / bm_in, H, ldH};
    fin.it[fin.n++] = {ws_of(l), G + y.b_off, 2, slot, H, ldH};
    if (fin.n + 2 > kMaxFinalItems) {
      grad_final(e->stream, fin);
      fin.n = 0;
    }
    const float* in = l == 0 ? Xd : e->a[l - 1];
    const int ld_in = l == 0 ? ldx : ldH;
    bool fused = false;
    if (l > 0) {
      ActEpi act = act_of(l - 1);
      int bm = rows_h;
      const int rc = run_gemm_dual(e, da, ldH, e->p_param() + y.w_off, y.ld_out, e->dA[pp ^ 1], ldH, T, H, H, &act,
                                   ws_of(l - 1), in, ld_in, G + y.w_off, y.ld_out, y.d_in, H, epi_w, &bm);
      if (rc < 0) return rc;
      fused = rc == 0;
      if (fused) bm_in = bm;
    }
    if (!fused) {
      CHK(run_gemm(e, GEMM_TN, in, ld_in, da, ldH, G + y.w_off, y.ld_out, y.d_in, H, T, nullptr, epi_w));
      if (l > 0) {
        ActEpi act = act_of(l - 1);
        CHK(run_gemm(e, GEMM_NT, da, ldH, e->p_param() + y.w_off, y.ld_out, e->dA[pp ^ 1], ldH, T, H, H, nullptr, EPI_DACT,
                     nullptr, ws_of(l - 1), cfg_h, &act));
        bm_in = rows_h;
      }
    }
    if (fire && e->cb) e->cb(e->cb_user, L - l);
    pp ^= 1;
  }
  {
    ProfScope ps(e, KF_COLSUM, 0, 0);
    grad_final(e->stream, fin);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// padded layout of `k` segments of seg_rows[.] rows
void stack_layout(const tfk_engine* e, const int32_t* seg_rows, int k, Stack* st) {
  const int align = stack_align(e);
  st->k = k;
  int row = 0, valid = 0;
  for (int i = 0; i < k; ++i) {
    st->r0[i] = row;
    st->rows[i] = seg_rows[i];
    st->span[i] = (int)up((size_t)seg_rows[i], (size_t)align);
    row += st->span[i];
    valid += seg_rows[i];
  }
  st->T_pad = row;
  st->T_valid = valid;
  st->d_vend = nullptr;
}

// forward + loss + backward of one stacked pass whose input is staged (Xd: [T_pad, ld], yd: labels with -1 on padding)
int run_stacked(tfk_engine* e, const float* Xd, int ld, const int32_t* yd, const Stack& st, int flags, int slot_before) {
  e->colsum_done = false;  // (as train_or_eval: never inherited from a call that failed half way)
  if (e->bf16) {
    const float* x = Xd;
    CHK(twin_input(e, &x, &ld, st.T_pad));
    Xd = x;
  }
  CHK(seg_stats(e));
  const uint32_t call0 = e->call_counter;
  e->call_counter += (uint32_t)st.k;
  CHK(forward_stacked(e, Xd, ld, st, call0));
  {
    ProfScope ps(e, KF_SOFTMAX_XENT, 0, 8.0 * st.T_pad * e->O);
    Twin tw;
    if (e->bf16) { tw.p = e->logb; tw.ld = e->ldOb; tw.x3 = e->x3; }
    softmax_xent(e->stream, e->logits, yd, st.T_pad, e->O, e->ldO, e->row_loss, 1, tw);
  }
  {
    ProfScope ps(e, KF_COLSUM, 0, 4.0 * st.T_pad * e->O);
    colsum_loss(e->stream, e->logits, st.T_pad, e->ldO, e->ws_bwd + (size_t)e->L * e->ws_bwd_stride, e->row_loss, st.T_pad,
                e->p_scalars(), e->scalars_fresh, st.T_valid, st.k);
    e->colsum_done = true;
    e->scalars_fresh = false;
  }
  const bool fire = (flags & TFK_LAST_MICROBATCH) != 0;
  if (fire) {
    if (e->later_mb > 0) {
      ProfScope ps(e, KF_MISC, 0, 8.0 * e->E);
      scale_inplace(e->stream, e->p_ema(), e->E, (float)pow((double)e->bn_decay, (double)e->later_mb));
    }
    if (e->cb) e->cb(e->cb_user, e->L + 2);
  }
  CHK(backward_stacked(e, Xd, ld, st, call0, fire));
  if (fire && e->cb) e->cb(e->cb_user, e->L + 1);
  e->grads_fresh = false;
  HIPCHK(hipGetLastError());
  CHK(finish_slot(e, flags, slot_before));
  e->last_T = st.T_pad; e->last_nfw = e->L; e->last_call = call0; e->last_in = Xd;
  return 0;
}

// stage a stacked pass from a [T, F] matrix (host, or device with TFK_DEVICE_PTRS) whose segments lie back to back
int stage_stacked(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int flags, Stack* st, const float** Xd,
                  int* ld_out, const int32_t** yd) {
  if (ldx < e->F) return fail(-1, "ldx %lld < input_dim %d", (long long)ldx, e->F);
  const int s = e->slot;
  const bool dev = (flags & TFK_DEVICE_PTRS) != 0;
  // the row_vend table always goes through the pinned side of the slot
  if (e->slot_used[s]) HIPCHK(hipEventSynchronize(e->copy_done[s]));
  int32_t* vend = e->hSeg[s];
  for (int i = 0; i < st->k; ++i)
    for (int unit = st->r0[i] / 64; unit < (st->r0[i] + st->span[i]) / 64; ++unit) vend[unit] = st->r0[i] + st->rows[i];
  CHK(wait_slot_free(e, s));
  HIPCHK(hipMemcpyAsync(e->dSeg[s], vend, (size_t)(st->T_pad / 64) * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
  st->d_vend = e->dSeg[s];
  const bool dense = st->T_pad == st->T_valid;  // every segment already a multiple of the alignment: no padding rows
  if (dev) {
    HIPCHK(hipEventRecord(e->copy_done[s], e->copy_stream));
    HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done[s], 0));
    if (dense && (ldx % 4) == 0 && (e->F % 4) == 0 && (((uintptr_t)X) % 16) == 0) {
      *Xd = X; *ld_out = (int)ldx; *yd = y;  // multiplied where it lies
    } else {
      HIPCHK(hipMemsetAsync(e->dY[s], 0xFF, (size_t)st->T_pad * sizeof(int32_t), e->stream));  // label -1
      for (int i = 0, t0 = 0; i < st->k; t0 += st->rows[i], ++i) {
        HIPCHK(hipMemcpy2DAsync(e->dX[s] + (size_t)st->r0[i] * e->ldF, (size_t)e->ldF * 4, X + (size_t)t0 * ldx, (size_t)ldx * 4,
                                (size_t)e->F * 4, st->rows[i], hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->dY[s] + st->r0[i], y + t0, (size_t)st->rows[i] * sizeof(int32_t), hipMemcpyDeviceToDevice,
                              e->stream));
      }
      *Xd = e->dX[s]; *ld_out = e->ldF; *yd = e->dY[s];
    }
    e->slot_used[s] = true;
    e->slot ^= 1;
    return 0;
  }
  for (int t = 0; t < st->T_pad; ++t) e->hY[s][t] = -1;
  for (int i = 0, t0 = 0; i < st->k; t0 += st->rows[i], ++i) {
    memcpy(e->hY[s] + st->r0[i], y + t0, (size_t)st->rows[i] * sizeof(int32_t));
    for (int t = 0; t < st->rows[i]; ++t)
      memcpy(e->hX[s] + (size_t)(st->r0[i] + t) * e->F, X + (size_t)(t0 + t) * ldx, (size_t)e->F * sizeof(float));
  }
  // (padding rows of the pinned image are whatever an earlier batch left there: finite, and multiplied by zero gradients)
  HIPCHK(hipMemcpy2DAsync(e->dX[s], (size_t)e->ldF * 4, e->hX[s], (size_t)e->F * 4, (size_t)e->F * 4, st->T_pad,
                          hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(hipMemcpyAsync(e->dY[s], e->hY[s], (size_t)st->T_pad * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(hipEventRecord(e->copy_done[s], e->copy_stream));
  HIPCHK(hipStreamWaitEvent(e->stream, e->copy_done[s], 0));
  e->slot_used[s] = true;
  *Xd = e->dX[s]; *ld_out = e->ldF; *yd = e->dY[s];
  e->slot ^= 1;
  return 0;
}

// Cut the k micro-batches of a call into runs: consecutive segments that fit one stacked pass, and single segments that
// go through the ordinary path.  fn(first, count, stacked) is called for every run in order.
template <class Fn>
int for_each_run(const tfk_engine* e, const int32_t* seg_rows, int k, Fn fn) {
  const bool can = stack_eligible(e);
  const int align = stack_align(e);
  int i = 0;
  while (i < k) {
    int j = i, rows = 0;
    if (can)
      while (j < k && j - i < kMaxStack && seg_rows[j] > 0 && seg_rows[j] <= kMaxSegmentRows &&
             rows + (int)up((size_t)seg_rows[j], (size_t)align) <= kMaxStackRows) {
        rows += (int)up((size_t)seg_rows[j], (size_t)align);
        ++j;
      }
    if (j - i >= 2) {
      CHK(fn(i, j - i, true));
      i = j;
    } else {
      CHK(fn(i, 1, false));
      i += 1;
    }
  }
  return 0;
}

}  // namespace

namespace tfk {
// the other translation units of the library (features.hip) report through the same thread-local message
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
}  // namespace tfk

extern "C" {

int tfk_abi_version(void) { return TFK_ABI_VERSION; }
#ifndef TFK_BUILD_ID
#error "compile engine.hip through tfkaldi_amd/build.py: it passes -DTFK_BUILD_ID=<hash of the sources>"
#endif
// (the marker makes the id findable in the file without loading it: tfkaldi_amd/build.py library_id)
static const char kBuildId[] = "TFK_BUILD_ID=" TFK_BUILD_ID;
const char* tfk_build_id(void) { return kBuildId + 13; }
const char* tfk_last_error(void) { return g_err.c_str(); }

int tfk_state_bytes(const tfk_config* cfg, size_t* bytes) {
  CHK(validate(cfg));
  if (!bytes) return fail(-1, "bytes is NULL");
  std::vector<LayerLayout> lay;
  size_t P, E;
  compute_layout(cfg, lay, P, E);
  *bytes = total_state_floats(P, E, shadow_floats(cfg, lay)) * sizeof(float);
  return 0;
}

int tfk_create(const tfk_config* cfg, tfk_engine** out) { return create_impl(cfg, nullptr, 0, nullptr, out); }
int tfk_create_ex(const tfk_config* cfg, void* state, size_t state_bytes, void* stream, tfk_engine** out) {
  return create_impl(cfg, state, state_bytes, stream, out);
}

int tfk_destroy(tfk_engine* e) {
  if (!e) return 0;
  hipSetDevice(e->cfg.device);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->copy_stream) hipStreamSynchronize(e->copy_stream);
  if (e->opt_stream) hipStreamSynchronize(e->opt_stream);
  free_activations(e);
  for (void* p : e->host_garbage) (void)hipHostFree(p);
  e->host_garbage.clear();
  if (e->ev_loss) hipEventDestroy(e->ev_loss);
  if (e->ev_grow) hipEventDestroy(e->ev_grow);
  if (e->ev_opt_begin) hipEventDestroy(e->ev_opt_begin);
  if (e->ev_opt_vec) hipEventDestroy(e->ev_opt_vec);
  for (hipEvent_t ev : e->ev_adam) if (ev) hipEventDestroy(ev);
  if (e->opt_stream) hipStreamDestroy(e->opt_stream);
  if (e->d_snap) hipFree(e->d_snap);
  for (auto p : e->mean) if (p) hipFree(p);
  for (auto p : e->seg_mean) if (p) hipFree(p);
  for (auto p : e->seg_rstd) if (p) hipFree(p);
  for (auto p : e->rstd) if (p) hipFree(p);
  if (e->prior) hipFree(e->prior);
  if (e->Wb && e->own_wb) hipFree(e->Wb);
  if (e->d_checksum) hipFree(e->d_checksum);
  if (e->ws_splitk) hipFree(e->ws_splitk);
  for (void* p : {(void*)e->ctc_seg, (void*)e->ctc_lab_off, (void*)e->ctc_lab, (void*)e->ctc_lp, (void*)e->ctc_ab,
                  (void*)e->ctc_utt_loss, (void*)e->ctc_lse, (void*)e->ctc_off, (void*)e->ctc_bb, (void*)e->ctc_offb,
                  (void*)e->ctc_logz})
    if (p) hipFree(p);
  for (hipEvent_t ev : e->post_ev) hipEventDestroy(ev);
  for (int k = 0; k < 2; ++k) {
    if (e->h_ctc[k]) hipHostFree(e->h_ctc[k]);
    if (e->ctc_staged[k]) hipEventDestroy(e->ctc_staged[k]);
  }
  if (e->h_scalars) hipHostFree(e->h_scalars);
  if (e->h_post) hipHostFree(e->h_post);
  if (e->own_state && e->state) hipFree(e->state);
  for (int s = 0; s < 2; ++s) {
    if (e->copy_done[s]) hipEventDestroy(e->copy_done[s]);
    if (e->compute_done[s]) hipEventDestroy(e->compute_done[s]);
  }
  for (auto ev : e->ev_pool) hipEventDestroy(ev);
  if (e->copy_stream) hipStreamDestroy(e->copy_stream);
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

int tfk_tensor_count(tfk_engine* e, int kind, int layer, size_t* count) {
  if (!e || !count) return fail(-1, "NULL argument");
  TensorRef t;
  CHK(tensor_ref(e, kind, TFK_SLOT_PARAM, layer, &t));
  *count = (size_t)t.rows * t.cols;
  return 0;
}

int tfk_tensor_get(tfk_engine* e, int kind, int slot, int layer, float* host, size_t count) {
  if (!e || !host) return fail(-1, "NULL argument");
  TensorRef t;
  CHK(tensor_ref(e, kind, slot, layer, &t));
  if (count != (size_t)t.rows * t.cols) return fail(-1, "count %zu != %d x %d", count, t.rows, t.cols);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (slot == TFK_SLOT_GRAD && e->grads_fresh) {  // logically zero (not yet rewritten in memory)
    memset(host, 0, count * sizeof(float));
    return 0;
  }
  if (slot == TFK_SLOT_PARAM) need_params(e, -1);
  CHK(sync_streams(e));
  HIPCHK(hipMemcpy2D(host, (size_t)t.cols * 4, t.ptr, (size_t)t.ld * 4, (size_t)t.cols * 4, t.rows, hipMemcpyDeviceToHost));
  return 0;
}

int tfk_tensor_set(tfk_engine* e, int kind, int slot, int layer, const float* host, size_t count) {
  if (!e || !host) return fail(-1, "NULL argument");
  TensorRef t;
  CHK(tensor_ref(e, kind, slot, layer, &t));
  if (count != (size_t)t.rows * t.cols) return fail(-1, "count %zu != %d x %d", count, t.rows, t.cols);
  HIPCHK(hipSetDevice(e->cfg.device));
  if (slot == TFK_SLOT_PARAM) need_params(e, -1);
  CHK(sync_streams(e));
  if (slot == TFK_SLOT_GRAD && e->grads_fresh) {
    HIPCHK(hipMemsetAsync(e->p_grad(), 0, e->P * sizeof(float), e->stream));
    e->grads_fresh = false;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy2D(t.ptr, (size_t)t.ld * 4, host, (size_t)t.cols * 4, (size_t)t.cols * 4, t.rows, hipMemcpyHostToDevice));
  if (slot == TFK_SLOT_PARAM) e->shadow_dirty = true;
  return 0;
}

static double current_lr(const tfk_engine* e) {
  // tf.train.exponential_decay(init, global_step, num_steps, decay) * learning_rate_fact  (trainer.py:110-112)
  const double frac = e->cfg.num_steps > 0 ? (double)e->global_step / (double)e->cfg.num_steps : 0.0;
  return (double)e->cfg.init_learning_rate * pow((double)e->cfg.learning_rate_decay, frac) * e->lr_fact;
}

int tfk_scalar_get(tfk_engine* e, int which, double* value) {
  if (!e || !value) return fail(-1, "NULL argument");
  switch (which) {
    case TFK_GLOBAL_STEP: *value = (double)e->global_step; return 0;
    case TFK_LEARNING_RATE_FACT: *value = e->lr_fact; return 0;
    case TFK_INITIALISED_LAYERS: *value = (double)e->initialised_layers; return 0;
    case TFK_ADAM_STEPS: *value = (double)e->adam_t; return 0;
    case TFK_LEARNING_RATE: *value = current_lr(e); return 0;
    case TFK_BATCH_LOSS:
    case TFK_NUM_FRAMES: {
      HIPCHK(hipSetDevice(e->cfg.device));
      if (e->scalars_fresh) { *value = 0.0; return 0; }
      CHK(sync_streams(e));
      float h[2];
      HIPCHK(hipMemcpy(h, e->p_scalars(), sizeof(h), hipMemcpyDeviceToHost));
      *value = which == TFK_BATCH_LOSS ? h[0] : h[1];
      return 0;
    }
  }
  return fail(-1, "unknown scalar %d", which);
}

int tfk_scalar_set(tfk_engine* e, int which, double value) {
  if (!e) return fail(-1, "engine is NULL");
  switch (which) {
    case TFK_GLOBAL_STEP: e->global_step = (int64_t)value; return 0;
    case TFK_LEARNING_RATE_FACT: e->lr_fact = value; return 0;
    case TFK_INITIALISED_LAYERS: e->initialised_layers = (int)value; return 0;
    case TFK_ADAM_STEPS: e->adam_t = (int64_t)value; return 0;
  }
  return fail(-1, "scalar %d is read-only or unknown", which);
}

int tfk_accumulate(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, int flags) {
  return train_or_eval(e, X, ldx, y, T, flags, 1);
}
int tfk_eval_accumulate(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, int flags) {
  return train_or_eval(e, X, ldx, y, T, flags & ~TFK_LAST_MICROBATCH, 0);
}
int tfk_accumulate_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                       const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn, int flags) {
  const RawSpec r = {utt_len, U, context_width, cmvn};
  if (!utt_len) return fail(-1, "utt_len is NULL");
  return train_or_eval(e, raw, ldraw, y, T, flags, 1, &r);
}
int tfk_eval_accumulate_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                            const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn, int flags) {
  const RawSpec r = {utt_len, U, context_width, cmvn};
  if (!utt_len) return fail(-1, "utt_len is NULL");
  return train_or_eval(e, raw, ldraw, y, T, flags & ~TFK_LAST_MICROBATCH, 0, &r);
}

int tfk_accumulate_stacked(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, const int32_t* seg_rows,
                           int32_t k, int flags) {
  if (!e) return fail(-1, "engine is NULL");
  if (!X || !y || !seg_rows) return fail(-1, "X / y / seg_rows is NULL");
  if (k <= 0) return fail(-1, "no micro-batch (k = %d)", k);
  long total = 0;
  for (int i = 0; i < k; ++i) {
    if (seg_rows[i] <= 0) return fail(-1, "micro-batch %d of the stack is empty (%d rows)", i, seg_rows[i]);
    total += seg_rows[i];
  }
  if (total != T) return fail(-1, "the micro-batches hold %ld rows, expected T = %d", total, T);
  HIPCHK(hipSetDevice(e->cfg.device));
  std::vector<int> first(k + 1, 0);
  for (int i = 0; i < k; ++i) first[i + 1] = first[i] + seg_rows[i];
  return for_each_run(e, seg_rows, k, [&](int i0, int n, bool stacked) -> int {
    const int last = (i0 + n == k) ? (flags & TFK_LAST_MICROBATCH) : 0;
    const int fl = (flags & ~TFK_LAST_MICROBATCH) | last;
    const float* Xr = X + (size_t)first[i0] * ldx;
    const int32_t* yr = y + first[i0];
    if (!stacked) return train_or_eval(e, Xr, ldx, yr, seg_rows[i0], fl, 1);
    Stack st;
    stack_layout(e, seg_rows + i0, n, &st);
    CHK(reserve(e, st.T_pad));
    const float* Xd; const int32_t* yd; int ld;
    const int slot_before = e->slot;
    CHK(stage_stacked(e, Xr, ldx, yr, fl, &st, &Xd, &ld, &yd));
    return run_stacked(e, Xd, ld, yd, st, fl, slot_before);
  });
}

int tfk_accumulate_stacked_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                               const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn,
                               const int32_t* seg_utts, int32_t k, int flags) {
  if (!e) return fail(-1, "engine is NULL");
  if (!raw || !y || !utt_len || !seg_utts) return fail(-1, "raw / y / utt_len / seg_utts is NULL");
  if (flags & (TFK_DEVICE_PTRS | TFK_RAW_DEVICE)) return fail(-1, "tfk_accumulate_stacked_raw takes host pointers");
  if (k <= 0) return fail(-1, "no micro-batch (k = %d)", k);
  const int win = 2 * context_width + 1;
  if (context_width < 0 || e->F % win) return fail(-1, "input_dim %d is not a multiple of 2*context_width+1 = %d", e->F, win);
  const int D = e->F / win;
  std::vector<int> first_utt(k + 1, 0), first_row(k + 1, 0);
  std::vector<int32_t> seg_rows(k, 0);
  for (int i = 0; i < k; ++i) {
    if (seg_utts[i] <= 0) return fail(-1, "micro-batch %d of the stack holds no utterance", i);
    first_utt[i + 1] = first_utt[i] + seg_utts[i];
    if (first_utt[i + 1] > U) return fail(-1, "the micro-batches hold more than U = %d utterances", U);
    for (int u = first_utt[i]; u < first_utt[i + 1]; ++u) {
      if (utt_len[u] < 0) return fail(-1, "negative utterance length");
      seg_rows[i] += utt_len[u];
    }
    if (seg_rows[i] <= 0) return fail(-1, "micro-batch %d of the stack is empty", i);
    first_row[i + 1] = first_row[i] + seg_rows[i];
  }
  if (first_utt[k] != U || first_row[k] != T)
    return fail(-1, "the micro-batches hold %d utterances / %d frames, expected U = %d / T = %d", first_utt[k], first_row[k], U, T);
  HIPCHK(hipSetDevice(e->cfg.device));
  return for_each_run(e, seg_rows.data(), k, [&](int i0, int n, bool stacked) -> int {
    const int last = (i0 + n == k) ? (flags & TFK_LAST_MICROBATCH) : 0;
    const float* rr = raw + (size_t)first_row[i0] * ldraw;
    const int32_t* yr = y + first_row[i0];
    const int32_t* ul = utt_len + first_utt[i0];
    const float* cm = cmvn ? cmvn + (size_t)first_utt[i0] * 2 * D : nullptr;
    const int nu = first_utt[i0 + n] - first_utt[i0], nt = first_row[i0 + n] - first_row[i0];
    if (!stacked) {
      const RawSpec r = {ul, nu, context_width, cm};
      return train_or_eval(e, rr, ldraw, yr, nt, last, 1, &r);
    }
    Stack st;
    stack_layout(e, seg_rows.data() + i0, n, &st);
    CHK(reserve(e, st.T_pad));
    const float* Xd; const int32_t* yd; int ld;
    const int slot_before = e->slot;
    CHK(stage_raw(e, rr, ldraw, yr, nt, ul, nu, context_width, cm, &Xd, &ld, &yd, false, &st, seg_utts + i0));
    return run_stacked(e, Xd, ld, yd, st, last, slot_before);
  });
}

// ---- validation over several micro-batches in one call (reference neuralNetworks/trainer.py:356-441: one
// update_valid_loss run per micro-batch).  In evaluation mode the rows of a micro-batch are independent -- batch norm normalises
// with the MOVING statistics, dropout is the identity, L2Norm is row-wise -- so k runs are one pass of the GEMMs over the
// concatenated rows: no per-segment statistics, no padding between segments, every activation chain.  Passes are cut at
// micro-batch boundaries once they hold TFK_EVAL_PASS_ROWS rows (default 4096: large enough for the GEMMs to fill the chip and
// read the weights once per pass, small enough that the host-fed input of pass i + 1 -- staged through the double-buffered
// pinned slots on the copy stream -- crosses PCIe under the kernels of pass i; one 16384-row pass measured SLOWER than eight
// 2048-row ones at BASELINE cfg4, its 29 MB copy in front of everything.  One longer micro-batch still runs alone).  batch_loss += the k sums and num_frames += T as k tfk_eval_accumulate calls leave them, up to
// the fp32 order of the loss sum.
static int eval_pass_rows() {
  const char* q = getenv("TFK_EVAL_PASS_ROWS");
  const int n = q ? atoi(q) : 4096;
  return n > 0 ? n : 4096;
}
int tfk_eval_accumulate_stacked(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, const int32_t* seg_rows,
                                int32_t k, int flags) {
  if (!e) return fail(-1, "engine is NULL");
  if (!X || !y || !seg_rows) return fail(-1, "X / y / seg_rows is NULL");
  if (k <= 0) return fail(-1, "no micro-batch (k = %d)", k);
  long total = 0;
  for (int i = 0; i < k; ++i) {
    if (seg_rows[i] <= 0) return fail(-1, "micro-batch %d of the stack is empty (%d rows)", i, seg_rows[i]);
    total += seg_rows[i];
  }
  if (total != T) return fail(-1, "the micro-batches hold %ld rows, expected T = %d", total, T);
  const int cap = eval_pass_rows();
  for (int i0 = 0, r0 = 0; i0 < k;) {
    int n = 0, rows = 0;
    while (i0 + n < k && (n == 0 || rows + seg_rows[i0 + n] <= cap)) rows += seg_rows[i0 + n++];
    CHK(train_or_eval(e, X + (size_t)r0 * ldx, ldx, y + r0, rows, flags & ~TFK_LAST_MICROBATCH, 0));
    i0 += n;
    r0 += rows;
  }
  return 0;
}
int tfk_eval_accumulate_stacked_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                                    const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn,
                                    const int32_t* seg_utts, int32_t k, int flags) {
  if (!e) return fail(-1, "engine is NULL");
  if (!raw || !y || !utt_len || !seg_utts) return fail(-1, "raw / y / utt_len / seg_utts is NULL");
  if (flags & (TFK_DEVICE_PTRS | TFK_RAW_DEVICE)) return fail(-1, "tfk_eval_accumulate_stacked_raw takes host pointers");
  if (k <= 0) return fail(-1, "no micro-batch (k = %d)", k);
  const int win = 2 * context_width + 1;
  if (context_width < 0 || e->F % win) return fail(-1, "input_dim %d is not a multiple of 2*context_width+1 = %d", e->F, win);
  const int D = e->F / win;
  std::vector<int> first_utt(k + 1, 0), first_row(k + 1, 0);
  for (int i = 0; i < k; ++i) {
    if (seg_utts[i] <= 0) return fail(-1, "micro-batch %d of the stack holds no utterance", i);
    first_utt[i + 1] = first_utt[i] + seg_utts[i];
    if (first_utt[i + 1] > U) return fail(-1, "the micro-batches hold more than U = %d utterances", U);
    int rows = 0;
    for (int u = first_utt[i]; u < first_utt[i + 1]; ++u) {
      if (utt_len[u] < 0) return fail(-1, "negative utterance length");
      rows += utt_len[u];
    }
    if (rows <= 0) return fail(-1, "micro-batch %d of the stack is empty", i);
    first_row[i + 1] = first_row[i] + rows;
  }
  if (first_utt[k] != U || first_row[k] != T)
    return fail(-1, "the micro-batches hold %d utterances / %d frames, expected U = %d / T = %d", first_utt[k], first_row[k], U, T);
  const int cap = eval_pass_rows();
  for (int i0 = 0; i0 < k;) {
    int n = 1;
    while (i0 + n < k && first_row[i0 + n + 1] - first_row[i0] <= cap) ++n;
    const RawSpec r = {utt_len + first_utt[i0], first_utt[i0 + n] - first_utt[i0], context_width,
                       cmvn ? cmvn + (size_t)first_utt[i0] * 2 * D : nullptr};
    CHK(train_or_eval(e, raw + (size_t)first_row[i0] * ldraw, ldraw, y + first_row[i0], first_row[i0 + n] - first_row[i0],
                      flags & ~TFK_LAST_MICROBATCH, 0, &r));
    i0 += n;
  }
  return 0;
}

int tfk_accumulate_ctc(tfk_engine* e, const float* X, int64_t ldx, int32_t T, const int32_t* utt_len, int32_t U,
                       const int32_t* labels, const int32_t* label_len, int flags) {
  const CtcSpec c = {utt_len, U, labels, label_len};
  return train_or_eval(e, X, ldx, nullptr, T, flags, 1, nullptr, &c);
}
int tfk_eval_accumulate_ctc(tfk_engine* e, const float* X, int64_t ldx, int32_t T, const int32_t* utt_len, int32_t U,
                            const int32_t* labels, const int32_t* label_len, int flags) {
  const CtcSpec c = {utt_len, U, labels, label_len};
  return train_or_eval(e, X, ldx, nullptr, T, flags & ~TFK_LAST_MICROBATCH, 0, nullptr, &c);
}
int tfk_accumulate_ctc_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t T, const int32_t* utt_len, int32_t U,
                           int32_t context_width, const float* cmvn, const int32_t* labels, const int32_t* label_len,
                           int flags) {
  if (!utt_len) return fail(-1, "utt_len is NULL");
  const RawSpec r = {utt_len, U, context_width, cmvn};
  const CtcSpec c = {utt_len, U, labels, label_len};
  return train_or_eval(e, raw, ldraw, nullptr, T, flags, 1, &r, &c);
}
int tfk_eval_accumulate_ctc_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t T, const int32_t* utt_len,
                                int32_t U, int32_t context_width, const float* cmvn, const int32_t* labels,
                                const int32_t* label_len, int flags) {
  if (!utt_len) return fail(-1, "utt_len is NULL");
  const RawSpec r = {utt_len, U, context_width, cmvn};
  const CtcSpec c = {utt_len, U, labels, label_len};
  return train_or_eval(e, raw, ldraw, nullptr, T, flags & ~TFK_LAST_MICROBATCH, 0, &r, &c);
}

// The optimiser step in three parts, so that a data-parallel host can run Adam on each span of parameters as soon
// as ITS gradients are reduced while later collectives are still in flight:
//   begin  needs the reduced scalars / BN increments: learning rate, moving averages, loss hand-over
//   span   mean -> clip -> Adam on parameters [offset, offset + n)
//   end    waits for the loss on the host, re-initialises the accumulators (lazily), global_step += 1
int apply_begin(tfk_engine* e) {
  if (e->apply_open) return fail(-1, "tfk_apply_begin called twice without tfk_apply_end");
  CHK(join_optimizer(e));  // (a second step without a forward pass in between)
  const double lr = current_lr(e);
  e->adam_t += 1;
  // tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); w -= lr_t * m / (sqrt(v) + eps)
  const double t = (double)e->adam_t;
  e->cur_lr_t = (float)(lr * sqrt(1.0 - pow((double)e->b2, t)) / (1.0 - pow((double)e->b1, t)));
  if (e->grads_fresh)  // no micro-batch since the last apply: materialise the zeros Adam is about to read
    HIPCHK(hipMemsetAsync(e->p_grad(), 0, e->P * sizeof(float), e->stream));
  CHK(settle_scalars(e));
  // BN moving averages + re-initialisation of their increments + the loss hand-over, one launch
  {
    ProfScope ps(e, KF_EMA, 0, 16.0 * e->E);
    e->steps_begun += 1;
    e->loss_seq = e->loss_seq + 1 ? e->loss_seq + 1 : 1;  // (never 0: the word starts out as 0)
    step_finish(e->stream, e->p_mov(), e->p_ema(), e->E, e->p_scalars(), e->bn_decay, e->h_scalars_dev, e->d_snap, e->loss_seq);
  }
  // (an event record here sits between step_finish and the optimiser: 5.9 us of idle stream per step, profiles/r06_loss_seq.txt)
  if (e->loss_event) HIPCHK(hipEventRecord(e->ev_loss, e->stream));
  // An arena-mirroring shadow is ALWAYS written with the update: if it is not current (no forward pass since the
  // parameters were last set from outside -- e.g. a data-parallel rank that had no micro-batch in its first step) it is
  // made current first.  The decision must not depend on what this rank happened to run: under the sharded exchange
  // every rank has to gather the same thing (the shadow), or the collectives mismatch.
  if (e->bf16 && e->wb_aligned && e->shadow_dirty) {
    need_params(e, -1);
    CHK(refresh_shadow(e));
  }
  e->apply_direct = e->bf16 && e->wb_aligned;
  e->apply_open = true;
  return 0;
}
int apply_span(tfk_engine* e, size_t off, size_t n, hipStream_t st = nullptr) {
  if (!e->apply_open) return fail(-1, "tfk_apply_span outside tfk_apply_begin / tfk_apply_end");
  if (off >= e->P || n == 0) return 0;
  if (off + n > e->P) n = e->P - off;
  if ((off | n) & 3) return fail(-1, "parameter span [%zu, +%zu) is not a multiple of 4 floats", off, n);
  const size_t w_end = e->lay[0].b_off;  // the weight matrices (and their bf16 shadow) come first
  const size_t n_wb = (e->apply_direct && off < w_end) ? ((off + n < w_end ? off + n : w_end) - off) : 0;
  // on the optimiser stream the launch may run past the next micro-batch's loss_reduce: it reads the step's frame count
  // from the snapshot step_finish took
  ProfScope ps(e, KF_ADAM, 0, 28.0 * n, st);
  adam_apply(st ? st : e->stream, e->p_param() + off, e->p_grad() + off, e->p_m() + off, e->p_v() + off, n,
             st ? e->d_snap : e->p_scalars(), e->cur_lr_t, e->b1, e->b2, e->adam_eps, 0,
             n_wb ? (e->x3 ? e->Wb : e->Wb + off) : nullptr, n_wb, e->x3 ? &e->wb_map : nullptr, off);
  return 0;
}
// the whole optimiser step of tfk_apply, layer by layer on the optimiser stream (vectors first: every layer reads them)
int apply_overlapped(tfk_engine* e) {
  HIPCHK(hipEventRecord(e->ev_opt_begin, e->stream));
  HIPCHK(hipStreamWaitEvent(e->opt_stream, e->ev_opt_begin, 0));
  const size_t vec = e->lay[0].b_off;
  CHK(apply_span(e, vec, e->P - vec, e->opt_stream));
  HIPCHK(hipEventRecord(e->ev_opt_vec, e->opt_stream));
  for (int l = 0; l <= e->L; ++l) {
    CHK(apply_span(e, e->lay[l].w_off, e->lay[l].w_sz, e->opt_stream));
    HIPCHK(hipEventRecord(e->ev_adam[l], e->opt_stream));
  }
  e->opt_pending = true;
  return 0;
}
// The step's (loss, frames, #micro-batches) are in mapped memory once step_finish's sequence word shows this step's number.
// The stream is asked now and then whether it is still alive: a failed launch must not leave the host spinning, and a stream
// that has run dry without the word appearing is an error, not a reason to wait.
// (Rarely -- every 20 ms: hipStreamQuery puts a marker with a completion signal behind the last launch, i.e. behind the optimiser,
// and the next step's first kernel then waits ~6 us for it: asked once per step it gave back everything the missing event record
// had gained, profiles/r06_loss_seq.txt.)
int wait_loss(tfk_engine* e) {
  const unsigned* w = reinterpret_cast<const unsigned*>(e->h_scalars) + kStepSeqWord;
  auto asked = std::chrono::steady_clock::now();
  for (unsigned long spin = 1;; ++spin) {
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == e->loss_seq) return 0;
    if ((spin & 0x3ff) == 0 && std::chrono::steady_clock::now() - asked > std::chrono::milliseconds(20)) {
      asked = std::chrono::steady_clock::now();
      const hipError_t q = hipStreamQuery(e->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == e->loss_seq) return 0;
        return fail(-1, "the optimiser step's loss hand-over did not arrive (sequence word %u, expected %u)", *w, e->loss_seq);
      }
      if (q != hipErrorNotReady) return fail((int)q, "engine stream failed while the step was running: %s", hipGetErrorString(q));
    }
    __builtin_ia32_pause();
  }
}
int apply_end(tfk_engine* e, float* average_loss) {
  if (!e->apply_open) return fail(-1, "tfk_apply_end without tfk_apply_begin");
  e->apply_open = false;
  if (!e->apply_direct) e->shadow_dirty = true;
  HIPCHK(hipGetLastError());
  if (e->loss_event) HIPCHK(hipEventSynchronize(e->ev_loss));
  else CHK(wait_loss(e));
  e->steps_seen = e->steps_begun;  // (step_finish runs behind every micro-batch of the step: their input slots are free)
  e->grads_fresh = true;
  e->scalars_fresh = true;  // init_loss / init_num_frames (trainer.py:350-352) without a memset
  if (check_kernel_errors(e)) {
    if (average_loss) *average_loss = NAN;
    return -1;
  }
  if (!(e->h_scalars[1] > 0.f)) {
    // no frame reached this step (on any rank): G / num_frames is 0/0.  The optimiser kernel left the parameters
    // alone (adam_kernel); undo the step bookkeeping and say so instead of returning a NaN loss.
    e->adam_t -= 1;
    if (average_loss) *average_loss = NAN;
    return fail(-1, "optimiser step without frames: num_frames = %g (no micro-batch was accumulated)",
                (double)e->h_scalars[1]);
  }
  e->global_step += 1;
  if (average_loss) *average_loss = e->h_scalars[0] / e->h_scalars[1];
  return 0;
}

int tfk_apply_begin(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  return apply_begin(e);
}
int tfk_apply_span(tfk_engine* e, size_t offset_floats, size_t num_floats) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  return apply_span(e, offset_floats, num_floats);
}
int tfk_apply_end(tfk_engine* e, float* average_loss) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  return apply_end(e, average_loss);
}

int tfk_apply_enqueue(tfk_engine* e) {
  // tfk_apply without its wait: everything of the optimiser step is on the streams, tfk_apply_end collects the loss
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(apply_begin(e));
  if (e->opt_overlap) return apply_overlapped(e);
  return apply_span(e, 0, e->P);
}

int tfk_apply(tfk_engine* e, float* average_loss) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(apply_begin(e));
  if (e->opt_overlap) CHK(apply_overlapped(e));
  else CHK(apply_span(e, 0, e->P));  // one Adam launch over the whole parameter arena
  return apply_end(e, average_loss);
}

int tfk_eval_finish(tfk_engine* e, float* average_loss) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(read_scalars(e));
  if (average_loss) *average_loss = e->h_scalars[0] / e->h_scalars[1];
  return 0;
}

int tfk_halve_learning_rate(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  e->lr_fact *= 0.5;
  return 0;
}
int tfk_add_layer(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  if (!e->cfg.layerwise_init) return fail(-1, "control op 'add' only exists with layerwise_init");
  e->initialised_layers += 1;
  return 0;
}
int tfk_init_last_layer(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  if (!e->cfg.layerwise_init) return fail(-1, "control op 'init' only exists with layerwise_init");
  // re-run the initialisers of layer L: weights ~ N(0, stddev 0) = 0, biases = 0 (dnn.py:67-68, 114-120)
  const LayerLayout& o = e->lay[e->L];
  HIPCHK(hipSetDevice(e->cfg.device));
  need_params(e, -1);
  CHK(join_optimizer(e));
  HIPCHK(hipMemsetAsync(e->p_param() + o.w_off, 0, o.w_sz * sizeof(float), e->stream));
  HIPCHK(hipMemsetAsync(e->p_param() + o.b_off, 0, o.b_sz * sizeof(float), e->stream));
  if (e->bf16 && e->wb_aligned && !e->shadow_dirty) {
    // a current arena-mirroring shadow stays current: zero its output-layer span too instead of rebuilding it from
    // every fp32 master (under the sharded exchange the masters of other ranks' spans are not valid here)
    HIPCHK(hipMemsetAsync(e->Wb + e->wb_off[e->L], 0,
                          (e->x3 ? x3::elems(o.d_in, e->wb_ld[e->L]) : o.w_sz) * sizeof(bf16_t), e->stream));
  } else {
    e->shadow_dirty = true;
  }
  return 0;
}

int tfk_set_prior(tfk_engine* e, const float* prior, size_t count) {
  if (!e || !prior) return fail(-1, "NULL argument");
  if (count != (size_t)e->O) return fail(-1, "prior has %zu entries, expected %d", count, e->O);
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(e->prior, prior, count * sizeof(float), hipMemcpyHostToDevice));
  e->have_prior = true;
  return 0;
}

static int posteriors_impl(tfk_engine* e, const float* X, int64_t ldx, int32_t N, float* out, int64_t ldo, int flags,
                           const RawSpec* raw) {
  if (!e) return fail(-1, "engine is NULL");
  if (raw && (flags & TFK_DEVICE_PTRS)) return fail(-1, "the raw entry points take host pointers (TFK_RAW_DEVICE: raw alone on the device)");
  if (!raw && (flags & TFK_RAW_DEVICE)) return fail(-1, "TFK_RAW_DEVICE belongs to the *_raw entry points");
  if (N <= 0) return fail(-1, "empty utterance (N = %d)", N);
  if (!X || !out) return fail(-1, "X / out is NULL");
  if (ldo < e->O) return fail(-1, "ldo %lld < output_dim %d", (long long)ldo, e->O);
  if ((flags & TFK_LOG_DIV_PRIOR) && !e->have_prior) return fail(-1, "TFK_LOG_DIV_PRIOR without tfk_set_prior");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(reserve(e, N));
  const float* Xd; const int32_t* yd; int ld;
  const int slot_before = e->slot;
  if (raw) CHK(stage_raw(e, X, ldx, nullptr, N, raw->utt_len, raw->U, raw->context, raw->cmvn, &Xd, &ld, &yd, (flags & TFK_RAW_DEVICE) != 0));
  else CHK(stage_input(e, X, ldx, nullptr, N, flags, &Xd, &ld, &yd));
  if (e->bf16) CHK(twin_input(e, &Xd, &ld, N));
  const int nact = e->nact();
  const uint32_t call = e->call_counter++;
  const float* prior = (flags & TFK_LOG_DIV_PRIOR) ? e->prior : nullptr;
  const bool want_logits = (flags & TFK_RAW_LOGITS) != 0;
  // a caller that hands over PINNED host memory gets the result by DMA, without a staging copy
  bool pinned_out = false;
  if (!(flags & TFK_DEVICE_PTRS)) {
    hipPointerAttribute_t attr;
    pinned_out = hipPointerGetAttributes(&attr, out) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned_out) (void)hipGetLastError();  // (an unregistered pointer reports an error: not one of ours)
  }
  if (pinned_out && !want_logits && !e->bf16 && e->post_chunk > 0 && N >= 2 * e->post_chunk) {
    // Long passes (batched decode): rows are independent in evaluation mode, so the pass runs in chunks and the
    // results of chunk c travel to the host on the copy stream while chunk c + 1 is being computed -- the 8 KB per
    // frame going back over PCIe would otherwise cost as much time as the forward pass itself.
    const int nchunks = (N + e->post_chunk - 1) / e->post_chunk;
    while ((int)e->post_ev.size() < nchunks) {
      hipEvent_t ev;
      HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      e->post_ev.push_back(ev);
    }
    // on an error inside the loop: nothing may still be writing the caller's buffer or reading the input slot
    auto bail = [&](int rc) {
      (void)hipStreamSynchronize(e->copy_stream);
      (void)hipStreamSynchronize(e->stream);
      (void)finish_slot(e, flags, slot_before);
      return rc;
    };
#define CHKB(expr)                         \
  do {                                     \
    int rc_ = (expr);                      \
    if (rc_ != 0) return bail(rc_);        \
  } while (0)
#define HIPB2(expr)                                                                                   \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return bail(fail((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_)));  \
  } while (0)
    for (int c = 0; c < nchunks; ++c) {
      const int r0 = c * e->post_chunk, n = std::min(e->post_chunk, N - r0);
      CHKB(forward(e, Xd + (size_t)r0 * ld, ld, n, 0, nact, nact, call));
      {
        ProfScope ps(e, KF_SOFTMAX, 0, 8.0 * n * e->O);
        softmax_rows(e->stream, e->logits, n, e->O, e->ldO, e->post + (size_t)r0 * e->ldO, e->ldO, prior);
      }
      if (c == nchunks - 1) CHKB(finish_slot(e, flags, slot_before));  // the input slot has been read for the last time
      HIPB2(hipEventRecord(e->post_ev[c], e->stream));
      HIPB2(hipStreamWaitEvent(e->copy_stream, e->post_ev[c], 0));
      HIPB2(hipMemcpy2DAsync(out + (size_t)r0 * ldo, (size_t)ldo * 4, e->post + (size_t)r0 * e->ldO, (size_t)e->ldO * 4,
                             (size_t)e->O * 4, n, hipMemcpyDeviceToHost, e->copy_stream));
    }
#undef CHKB
#undef HIPB2
    HIPCHK(hipStreamSynchronize(e->copy_stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    const int n_last = N - (nchunks - 1) * e->post_chunk;
    e->last_T = n_last; e->last_nfw = nact; e->last_call = call;
    e->last_in = Xd + (size_t)(nchunks - 1) * e->post_chunk * ld;
    return 0;
  }
  CHK(forward(e, Xd, ld, N, 0, nact, nact, call));
  if (flags & TFK_DEVICE_PTRS) {
    if (want_logits) {
      HIPCHK(hipMemcpy2DAsync(out, (size_t)ldo * 4, e->logits, (size_t)e->ldO * 4, (size_t)e->O * 4, N,
                              hipMemcpyDeviceToDevice, e->stream));
    } else {
      ProfScope ps(e, KF_SOFTMAX, 0, 8.0 * N * e->O);
      softmax_rows(e->stream, e->logits, N, e->O, e->ldO, out, ldo, prior);
    }
  } else {
    if (!want_logits) {
      ProfScope ps(e, KF_SOFTMAX, 0, 8.0 * N * e->O);
      softmax_rows(e->stream, e->logits, N, e->O, e->ldO, e->post, e->ldO, prior);
    }
    CHK(finish_slot(e, flags, slot_before));
    const float* src = want_logits ? e->logits : e->post;
    if (pinned_out) {
      HIPCHK(hipMemcpy2DAsync(out, (size_t)ldo * 4, src, (size_t)e->ldO * 4, (size_t)e->O * 4, N, hipMemcpyDeviceToHost,
                              e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      HIPCHK(hipGetLastError());
      e->last_T = N; e->last_nfw = nact; e->last_call = call; e->last_in = Xd;
      return 0;
    }
    const size_t need = (size_t)N * e->O;
    if (need > e->h_post_floats) {
      HIPCHK(hipStreamSynchronize(e->stream));
      if (e->h_post) hipHostFree(e->h_post);
      e->h_post = nullptr;
      HIPCHK(hipHostMalloc((void**)&e->h_post, need * sizeof(float), hipHostMallocDefault));
      e->h_post_floats = need;
    }
    HIPCHK(hipMemcpy2DAsync(e->h_post, (size_t)e->O * 4, src, (size_t)e->ldO * 4,
                            (size_t)e->O * 4, N, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int t = 0; t < N; ++t) memcpy(out + (size_t)t * ldo, e->h_post + (size_t)t * e->O, (size_t)e->O * sizeof(float));
  }
  HIPCHK(hipGetLastError());
  e->last_T = N; e->last_nfw = nact; e->last_call = call; e->last_in = Xd;
  return 0;
}

int tfk_posteriors(tfk_engine* e, const float* X, int64_t ldx, int32_t N, float* out, int64_t ldo, int flags) {
  const int rc = posteriors_impl(e, X, ldx, N, out, ldo, flags, nullptr);
  // (host output: the pass has been waited for, so a kernel-reported failure is visible)
  return (rc == 0 && !(flags & TFK_DEVICE_PTRS)) ? check_kernel_errors(e) : rc;
}
int tfk_posteriors_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t N, const int32_t* utt_len, int32_t U,
                       int32_t context_width, const float* cmvn, float* out, int64_t ldo, int flags) {
  const RawSpec r = {utt_len, U, context_width, cmvn};
  if (!utt_len) return fail(-1, "utt_len is NULL");
  return posteriors_impl(e, raw, ldraw, N, out, ldo, flags, &r);
}

int tfk_reduce_region(tfk_engine* e, void** device_ptr, size_t* num_floats) {
  if (!e || !device_ptr || !num_floats) return fail(-1, "NULL argument");
  *device_ptr = e->p_grad();
  *num_floats = e->reduce_floats;
  return 0;
}
int tfk_moment_regions(tfk_engine* e, void** adam_m, void** adam_v, size_t* num_floats) {
  if (!e || !adam_m || !adam_v || !num_floats) return fail(-1, "NULL argument");
  *adam_m = e->p_m();
  *adam_v = e->p_v();
  *num_floats = e->P;
  return 0;
}
int tfk_param_region(tfk_engine* e, void** device_ptr, size_t* num_floats) {
  if (!e || !device_ptr || !num_floats) return fail(-1, "NULL argument");
  *device_ptr = e->p_param();
  *num_floats = e->P;
  return 0;
}
int tfk_num_buckets(tfk_engine* e, int* n) {
  if (!e || !n) return fail(-1, "NULL argument");
  *n = e->L + 3;
  return 0;
}
int tfk_reduce_bucket(tfk_engine* e, int bucket, size_t* offset_floats, size_t* num_floats) {
  if (!e || !offset_floats || !num_floats) return fail(-1, "NULL argument");
  if (bucket < 0 || bucket > e->L + 2) return fail(-1, "bucket %d out of range", bucket);
  if (bucket == e->L + 1) {  // every bias / beta gradient
    *offset_floats = e->lay[0].b_off;
    *num_floats = e->P - e->lay[0].b_off;
  } else if (bucket == e->L + 2) {  // batch_loss, num_frames, #micro-batches + the BN moving-average increments
    *offset_floats = e->P;
    *num_floats = kScalarFloats + e->E;
  } else {
    const LayerLayout& y = e->lay[e->L - bucket];
    *offset_floats = y.w_off;
    *num_floats = y.w_sz;
  }
  return 0;
}
int tfk_zero_accumulators(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(join_optimizer(e));
  HIPCHK(hipMemsetAsync(e->p_grad(), 0, e->reduce_floats * sizeof(float), e->stream));
  e->grads_fresh = false;  // physically zero now
  e->scalars_fresh = false;
  return 0;
}
int tfk_set_bucket_callback(tfk_engine* e, tfk_bucket_fn fn, void* user) {
  if (!e) return fail(-1, "engine is NULL");
  e->cb = fn;
  e->cb_user = user;
  return 0;
}
int tfk_set_layer_callback(tfk_engine* e, tfk_layer_fn fn, void* user) {
  if (!e) return fail(-1, "engine is NULL");
  e->layer_cb = fn;
  e->layer_user = user;
  return 0;
}
int tfk_params_touched(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  e->shadow_dirty = true;
  if (e->apply_open) e->apply_direct = false;  // tfk_apply_end must not declare the shadow current
  return 0;
}
int tfk_twins_from_params(tfk_engine* e, size_t offset, size_t n, void* stream, int* current) {
  if (!e || !current) return fail(-1, "NULL argument");
  *current = 0;
  if (!e->bf16) { *current = 1; return 0; }  // exact fp32: the contractions read the parameters themselves
  if (!e->x3 || !e->wb_aligned || e->shadow_dirty) return 0;  // (a shadow that is not current is rebuilt whole by the next pass)
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : e->stream;
  for (int l = 0; l <= e->L; ++l) {
    const LayerLayout& y = e->lay[l];
    if (y.w_off + y.w_sz <= offset || y.w_off >= offset + n) continue;
    if (y.w_off < offset || y.w_off + y.w_sz > offset + n)
      return fail(-1, "parameter span [%zu, +%zu) cuts through the weight matrix of layer %d", offset, n, l);
    to_bf16_rows(st, e->p_param() + y.w_off, y.ld_out, e->Wb + e->wb_off[l], e->wb_ld[l], y.d_in, y.d_out, 1);
  }
  HIPCHK(hipGetLastError());
  *current = 1;
  return 0;
}
int tfk_shadow_region(tfk_engine* e, void** device_ptr, size_t* num_elems, int* mirrors_arena) {
  if (!e || !device_ptr || !num_elems || !mirrors_arena) return fail(-1, "NULL argument");
  // (x3: the shadow is three planes outside the arena -- nothing a sharded exchange could gather in place of the parameters)
  const bool mirrors = e->bf16 && !e->x3 && e->wb_aligned;
  *device_ptr = (e->bf16 && !e->x3) ? (void*)e->Wb : nullptr;
  *num_elems = mirrors ? e->lay[0].b_off : 0;
  *mirrors_arena = mirrors ? 1 : 0;
  return 0;
}
int tfk_twin_region(tfk_engine* e, int layer, void** device_ptr, size_t* bytes, int* rows) {
  if (!e || !device_ptr || !bytes || !rows) return fail(-1, "NULL argument");
  *device_ptr = nullptr;
  *bytes = 0;
  *rows = 0;
  if (layer < 0 || layer > e->L) return fail(-1, "layer %d out of range", layer);
  // only where the optimiser writes the twins with the update (x3 with a map of the arena): what a sharded exchange may gather
  // in place of the fp32 parameters
  if (!e->bf16 || !e->x3 || !e->wb_aligned) return 0;
  const LayerLayout& y = e->lay[layer];
  *device_ptr = e->Wb + e->wb_off[layer];
  *bytes = x3::elems(y.d_in, e->wb_ld[layer]) * sizeof(bf16_t);
  *rows = y.d_in;
  return 0;
}
int tfk_apply_writes_shadow(tfk_engine* e, int* direct) {
  if (!e || !direct) return fail(-1, "NULL argument");
  if (!e->apply_open) return fail(-1, "tfk_apply_writes_shadow outside tfk_apply_begin / tfk_apply_end");
  *direct = e->apply_direct ? 1 : 0;
  return 0;
}
int tfk_param_checksum(tfk_engine* e, int which, uint64_t* value) {
  if (!e || !value) return fail(-1, "NULL argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  const uint32_t* p;
  size_t words;
  if (which == 0) {
    p = reinterpret_cast<const uint32_t*>(e->p_param());
    words = e->P;
  } else if (which == 1) {
    if (!e->bf16 || e->x3 || !e->wb_aligned) return fail(-1, "no arena-mirroring bf16 shadow to checksum");
    p = reinterpret_cast<const uint32_t*>(e->Wb);
    words = e->lay[0].b_off / 2;
  } else if (which == 2) {
    p = reinterpret_cast<const uint32_t*>(e->p_param() + e->lay[0].b_off);
    words = e->P - e->lay[0].b_off;
  } else if (which == 3) {
    if (!e->bf16 || !e->x3) return fail(-1, "no three-plane twins to checksum");
    p = reinterpret_cast<const uint32_t*>(e->Wb);
    const LayerLayout& y = e->lay[e->L];
    words = (e->wb_off[e->L] + x3::elems(y.d_in, e->wb_ld[e->L])) / 2;  // (every twin, padding included: zeros everywhere)
  } else {
    return fail(-1, "tfk_param_checksum: which must be 0 (fp32 parameters), 1 (bf16 shadow), 2 (fp32 bias / beta vectors) or 3 "
                    "(three-plane twins of the weights)");
  }
  CHK(join_optimizer(e));
  if (!e->d_checksum) HIPCHK(hipMalloc((void**)&e->d_checksum, sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(e->d_checksum, 0, sizeof(unsigned long long), e->stream));
  checksum_words(e->stream, p, words, e->d_checksum);
  unsigned long long h = 0;
  HIPCHK(hipMemcpyAsync(&h, e->d_checksum, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *value = (uint64_t)h;
  return 0;
}
int tfk_set_later_microbatches(tfk_engine* e, int32_t later) {
  if (!e) return fail(-1, "engine is NULL");
  if (later < 0) return fail(-1, "later micro-batches must be >= 0");
  e->later_mb = later;
  return 0;
}

int tfk_synchronize(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(sync_streams(e));
  return 0;
}
int tfk_stream(tfk_engine* e, void** hip_stream) {
  if (!e || !hip_stream) return fail(-1, "NULL argument");
  *hip_stream = (void*)e->stream;
  return 0;
}

int tfk_profile_begin(tfk_engine* e) {
  if (!e) return fail(-1, "engine is NULL");
  e->prof.clear();
  e->ev_next = 0;
  e->profiling = true;
  return 0;
}
int tfk_profile_end(tfk_engine* e, tfk_kernel_stat* stats, int capacity, int* count) {
  if (!e || !count) return fail(-1, "NULL argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(sync_streams(e));
  e->profiling = false;
  tfk_kernel_stat acc[KF_COUNT];
  memset(acc, 0, sizeof(acc));
  for (int f = 0; f < KF_COUNT; ++f) snprintf(acc[f].name, sizeof(acc[f].name), "%s", kFamilyName[f]);
  for (const ProfRec& r : e->prof) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    acc[r.family].launches += 1;
    acc[r.family].total_ms += ms;
    acc[r.family].flops += r.flops;
    acc[r.family].bytes += r.bytes;
  }
  int n = 0;
  for (int f = 0; f < KF_COUNT; ++f)
    if (acc[f].launches > 0) {
      if (stats && n < capacity) stats[n] = acc[f];
      ++n;
    }
  *count = n;
  e->prof.clear();
  e->ev_next = 0;
  return 0;
}

int tfk_debug_fetch(tfk_engine* e, int what, int layer, float* host, size_t count) {
  if (!e || !host) return fail(-1, "NULL argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  CHK(sync_streams(e));
  const int T = e->last_T;
  if (T <= 0) return fail(-1, "no previous call to fetch from");
  if (what == TFK_DBG_LOGITS) {
    if (count != (size_t)T * e->O) return fail(-1, "count %zu != %d x %d", count, T, e->O);
    HIPCHK(hipMemcpy2D(host, (size_t)e->O * 4, e->logits, (size_t)e->ldO * 4, (size_t)e->O * 4, T, hipMemcpyDeviceToHost));
    return 0;
  }
  if (layer < 0 || layer >= e->L) return fail(-1, "layer %d out of range", layer);
  if (count != (size_t)T * e->H) return fail(-1, "count %zu != %d x %d", count, T, e->H);
  if (what == TFK_DBG_HIDDEN) {
    if (layer >= e->last_nfw) return fail(-1, "layer %d was not evaluated by the last call", layer);
    HIPCHK(hipMemcpy2D(host, (size_t)e->H * 4, e->a[layer], (size_t)e->ldH * 4, (size_t)e->H * 4, T, hipMemcpyDeviceToHost));
    return 0;
  }
  if (what == TFK_DBG_PREACT || what == TFK_DBG_BN_MEAN || what == TFK_DBG_BN_RSTD) {
    // the other tensors batch-norm's backward reads: the affine output z of the layer / its batch mean / rstd (row 0)
    if (what == TFK_DBG_PREACT) {
      HIPCHK(hipMemcpy2D(host, (size_t)e->H * 4, e->z[layer], (size_t)e->ldH * 4, (size_t)e->H * 4, T, hipMemcpyDeviceToHost));
    } else {
      memset(host, 0, count * sizeof(float));
      HIPCHK(hipMemcpy(host, what == TFK_DBG_BN_MEAN ? e->mean[layer] : e->rstd[layer], (size_t)e->H * 4, hipMemcpyDeviceToHost));
    }
    return 0;
  }
  if (what == TFK_DBG_DROPOUT_MASK) {
    const ActDesc d = act_desc(e, layer, 1, e->last_call);
    if (d.keep >= 1.f) {
      for (size_t i = 0; i < count; ++i) host[i] = 1.f;
      return 0;
    }
    dropout_mask(e->stream, d, e->dA[0], T, e->H, e->ldH);  // dA[0] is scratch between calls
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy2D(host, (size_t)e->H * 4, e->dA[0], (size_t)e->ldH * 4, (size_t)e->H * 4, T, hipMemcpyDeviceToHost));
    return 0;
  }
  return fail(-1, "unknown debug tensor %d", what);
}

int tfk_gemm_f32(void* stream, int layout, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                 int N, int K, const float* bias, int epi, int tile_config) {
  if (layout < 0 || layout > 2) return fail(-1, "bad layout %d", layout);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.stats = nullptr;
  g.act_a = g.act_z = g.act_mean = g.act_rstd = nullptr; g.act_nonlin = 0; g.stats_stride = 0;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = epi;
  const int rc = gemm_f32((GemmLayout)layout, g, tile_config, (hipStream_t)stream);
  if (rc != 0) return fail(rc, "gemm_f32 failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

int tfk_gemm_bf16(void* stream, int layout, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc,
                  int M, int N, int K, const float* bias, int epi) {
  if (layout < 0 || layout > 2) return fail(-1, "bad layout %d", layout);
  GemmArgsB g = {};
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = epi;
  const int rc = gemm_bf16((GemmLayout)layout, g, (hipStream_t)stream);
  if (rc != 0) return fail(rc, "gemm_bf16 failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

int tfk_split3(void* stream, const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols) {
  if (!src || !dst) return fail(-1, "NULL argument");
  if ((ldd & 31) || ldd < cols) return fail(-1, "tfk_split3: ldd %d (a multiple of 32, >= cols)", ldd);
  to_bf16_rows((hipStream_t)stream, src, lds, dst, ldd, rows, cols, 1);
  HIPCHK(hipGetLastError());
  return 0;
}

int tfk_debug_poison_splitk(tfk_engine* e) {
  // tests: leave a ticket of the split-K workspace taken, as an interrupted launch would -- the next fp32-emulating contraction
  // that splits its K finds no "first" block for tile 0, both halves wait for the other's partial sums, time out, and the step
  // must FAIL (check_kernel_errors) instead of handing out sums of garbage
  if (!e) return fail(-1, "engine is NULL");
  if (!e->ws_splitk) return fail(-1, "no split-K workspace yet (run a step of a shape that splits first)");
  const unsigned one = 1;
  HIPCHK(hipMemcpyAsync(e->ws_splitk, &one, sizeof(one), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

int tfk_gemm_bf16x3(void* stream, int layout, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc, int M,
                    int N, int K, const float* bias, int epi) {
  if (layout < 0 || layout > 2) return fail(-1, "bad layout %d", layout);
  GemmArgsB g = {};
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = epi;
  {  // (a process-wide split-K workspace for this tool entry: calls are expected on one stream at a time)
    static float* ws = nullptr;
    static size_t ws_floats = 0;
    const size_t need = gemm_bf16x3_splitk_floats((GemmLayout)layout, M, N, K);
    if (need > ws_floats) {
      if (ws) {
        hipDeviceSynchronize();
        hipFree(ws);
      }
      ws = nullptr;
      ws_floats = 0;
      if (hipMalloc((void**)&ws, need * sizeof(float)) != hipSuccess) return fail(-1, "split-K workspace of %zu floats", need);
      hipMemset(ws, 0, need * sizeof(float));
      ws_floats = need;
    }
    g.splitk_ws = ws;
    g.splitk_ws_floats = ws_floats;
  }
  const int rc = gemm_bf16x3((GemmLayout)layout, g, (hipStream_t)stream);
  if (rc != 0) return fail(rc, "gemm_bf16x3 failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

int tfk_gemm_bf16_dual(void* stream, const uint16_t* A_nt, int lda_nt, const uint16_t* B_nt, int ldb_nt, float* C_nt,
                       int ldc_nt, int M_nt, int N_nt, int K_nt, const uint16_t* A_tn, int lda_tn, const uint16_t* B_tn,
                       int ldb_tn, float* C_tn, int ldc_tn, int M_tn, int N_tn, int K_tn, int epi_tn) {
  GemmArgsB a = {}, w = {};
  a.A = A_nt; a.B = B_nt; a.C = C_nt; a.M = M_nt; a.N = N_nt; a.K = K_nt; a.lda = lda_nt; a.ldb = ldb_nt; a.ldc = ldc_nt;
  w.A = A_tn; w.B = B_tn; w.C = C_tn; w.M = M_tn; w.N = N_tn; w.K = K_tn; w.lda = lda_tn; w.ldb = ldb_tn; w.ldc = ldc_tn;
  w.epi = epi_tn;
  const int rc = gemm_bf16_dual(a, w, (hipStream_t)stream);
  if (rc == -1) return fail(-1, "gemm_bf16_dual: this pair of shapes is not eligible for the dual launch");
  if (rc != 0) return fail(rc, "gemm_bf16_dual failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}
int tfk_gemm_bf16_dual_config(int M_nt, int N_nt, int M_tn, int N_tn) { return gemm_bf16_dual_config(M_nt, N_nt, M_tn, N_tn); }
int tfk_gemm_bf16x3_dual(void* stream, const uint16_t* A_nt, int lda_nt, const uint16_t* B_nt, int ldb_nt, float* C_nt,
                         int ldc_nt, int M_nt, int N_nt, int K_nt, const uint16_t* A_tn, int lda_tn, const uint16_t* B_tn,
                         int ldb_tn, float* C_tn, int ldc_tn, int M_tn, int N_tn, int K_tn, int epi_tn) {
  GemmArgsB a = {}, w = {};
  a.A = A_nt; a.B = B_nt; a.C = C_nt; a.M = M_nt; a.N = N_nt; a.K = K_nt; a.lda = lda_nt; a.ldb = ldb_nt; a.ldc = ldc_nt;
  w.A = A_tn; w.B = B_tn; w.C = C_tn; w.M = M_tn; w.N = N_tn; w.K = K_tn; w.lda = lda_tn; w.ldb = ldb_tn; w.ldc = ldc_tn;
  w.epi = epi_tn;
  const int rc = gemm_bf16x3_dual(a, w, (hipStream_t)stream);
  if (rc == -1) return fail(-1, "gemm_bf16x3_dual: this pair of shapes is not eligible for the dual launch");
  if (rc != 0) return fail(rc, "gemm_bf16x3_dual failed: %s", hipGetErrorString((hipError_t)rc));
  return 0;
}

int tfk_gemm_bf16_force_config(int cfg) {
  gemm_bf16_force_config(cfg);
  return 0;
}
int tfk_gemm_bf16_config(int M, int N) { return gemm_bf16_pick_config(M, N); }

}  // extern "C"
