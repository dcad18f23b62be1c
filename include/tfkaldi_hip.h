/*
 * tfkaldi_hip.h -- C ABI of the MI355X (gfx950) DNN acoustic-model training engine.
 *
 * This is the drop-in boundary for the hot path of vrenkens/tfkaldi: everything the reference runs
 * inside `tf.Session.run()` for neuralNetworks/trainer.py (Trainer / CrossEnthropyTrainer),
 * neuralNetworks/decoder.py (Decoder) and neuralNetworks/classifiers/{dnn,layer,activation}.py.
 * The reference has no FFI of its own (its "native layer" is TensorFlow); each entry point below names
 * the TensorFlow graph fetch (reference file:line) it replaces.  A Python host binds it with ctypes
 * (tfkaldi_amd/_lib.py); see INTEGRATION.md.
 *
 * Conventions: plain C, every call returns 0 on success or a non-zero status (tfk_last_error() gives
 * the message, thread-local).  Host buffers are caller-owned and only read/written during the call.
 * One engine per GPU; calls on one engine must be serialised by the caller; all device work is
 * stream-ordered on the engine's HIP stream.  Matrices are dense row-major fp32 unless a leading
 * dimension is given.
 */
#ifndef TFKALDI_HIP_H
#define TFKALDI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFK_ABI_VERSION 8

typedef struct tfk_engine tfk_engine;

/* Arithmetic of the affine / gradient contractions.  TFK_DTYPE_F32 is the reference's (exact-fp32 MFMA).
 * TFK_DTYPE_BF16 is mixed precision (BASELINE cfg3 / cfg4): GEMM operands -- layer inputs, weights, dZ -- are
 * rounded to bfloat16 (round to nearest even), products accumulate in fp32; master parameters, batch-norm
 * statistics, the loss, gradient sums and the Adam update stay fp32. */
/* TFK_DTYPE_F32X3: fp32 arithmetic EMULATED on the bf16 matrix pipe.  Every GEMM operand is split, exactly, into three
 * bfloat16 pieces (8 + 8 + 8 significand bits) and the six piece products of order <= 2^-16 are accumulated in fp32: products of
 * bf16 values are exact in fp32 and the dropped pairs are below 2^-24 of a product, so a contraction differs from the exact
 * fp32 dot product only by fp32 accumulation error -- against float64 it is at least as close as the fp32 MFMA chain of
 * TFK_DTYPE_F32 (tools/bf16x3_probe.py) -- at 6/16 of that mode's matrix-pipe time.  Parameters, statistics, loss, gradient
 * sums and the optimiser are fp32 as in every mode; held to the same parity bounds as TFK_DTYPE_F32. */
enum { TFK_DTYPE_F32 = 0, TFK_DTYPE_BF16 = 1, TFK_DTYPE_F32X3 = 2 };

/* nonlinearity of the hidden layers: neuralNetworks/nnet.py:48-65 */
enum { TFK_NONLIN_RELU = 0, TFK_NONLIN_SIGMOID = 1, TFK_NONLIN_TANH = 2, TFK_NONLIN_LINEAR = 3 };

/*
 * Network + optimiser description: the `[nnet]` hyper-parameters that reach the TF graph
 * (config/config_AURORA4.cfg:102-153 via neuralNetworks/nnet.py:17-78 and trainer.py:13-15).
 * Hidden layer = affine -> Batchnorm -> nonlin -> L2Norm -> Dropout (activation.py:22-42 order).
 */
typedef struct tfk_config {
  int32_t struct_size;         /* = sizeof(tfk_config) */
  int32_t device;              /* HIP device ordinal */
  int32_t input_dim;           /* F: spliced feature dimension (nnet.py:39) */
  int32_t num_layers;          /* L: hidden layers (dnn.py:28) */
  int32_t num_units;           /* H (dnn.py:29) */
  int32_t output_dim;          /* O: pdf-ids (classifier.py:13) */
  int32_t nonlin;              /* TFK_NONLIN_* */
  int32_t batch_norm;          /* activation.py:145-161 */
  int32_t l2_norm;             /* activation.py:87-111 */
  float keep_prob;             /* activation.py:113-143; >= 1 disables dropout */
  int32_t layerwise_init;      /* dnn.py:81-122 */
  float init_learning_rate;    /* trainer.py:110-112 */
  float learning_rate_decay;   /* trainer.py:110-112 */
  int32_t num_steps;           /* trainer.py:88 */
  int32_t max_frames;          /* initial frame capacity of one accumulate call (grows on demand) */
  uint64_t seed;               /* dropout RNG seed */
  /* TF-0.1x defaults that live outside the reference tree; 0 selects the default in brackets */
  float bn_decay;              /* [0.999] tf.contrib.layers.batch_norm decay */
  float bn_epsilon;            /* [1e-3] */
  float adam_beta1;            /* [0.9]   tf.train.AdamOptimizer */
  float adam_beta2;            /* [0.999] */
  float adam_epsilon;          /* [1e-8] */
  int32_t compute_dtype;       /* TFK_DTYPE_*: arithmetic of the three contractions (0 = fp32, the reference's) */
} tfk_config;

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Bytes of persistent state (parameters, gradient sums, Adam moments, BN moving statistics) an
 * engine of this shape keeps in HBM. */
int tfk_state_bytes(const tfk_config* cfg, size_t* bytes);

/* Build the engine: replaces Trainer.__init__ graph construction (trainer.py:37-215) and
 * Decoder.__init__ (decoder.py:11-47).  Parameters start as the reference initialises them EXCEPT the
 * random hidden weights, which the host injects with tfk_tensor_set (layer.py:39-48: hidden W ~
 * N(0, 1/sqrt(d_in)), biases 0; dnn.py:67-68: output W = 0). */
int tfk_create(const tfk_config* cfg, tfk_engine** out);

/* As tfk_create, but the persistent state lives in caller-provided device memory (e.g. a torch
 * tensor, so torch.distributed can all-reduce views of it) and work is enqueued on `stream`
 * (a hipStream_t; NULL = the engine creates its own).  `state` must be 256-byte aligned, hold
 * tfk_state_bytes() bytes, and outlive the engine. */
int tfk_create_ex(const tfk_config* cfg, void* state, size_t state_bytes, void* stream, tfk_engine** out);

int tfk_destroy(tfk_engine* e);

const char* tfk_last_error(void);
int tfk_abi_version(void);
/* Build provenance: the hash of the sources this library was compiled from (tfkaldi_amd/build.py source_id(): every
 * .hip / .h under csrc/, this header, the compiler flags).  The binding refuses a library whose id is not its tree's. */
const char* tfk_build_id(void);

/* ---- state access (checkpointing, weight injection, parity tests) ------------------------------ */

enum { /* tensor kind */
  TFK_WEIGHTS = 0,         /* layer l in [0, L]: [d_in, d_out] row-major (layer.py:42-44) */
  TFK_BIASES = 1,          /* [d_out]                                  (layer.py:46-48) */
  TFK_BN_BETA = 2,         /* layer l in [0, L): [H]   (batch_norm center=True, scale=False) */
  TFK_BN_MOVING_MEAN = 3,  /* [H] init 0 */
  TFK_BN_MOVING_VAR = 4    /* [H] init 1 */
};
enum { /* tensor slot */
  TFK_SLOT_PARAM = 0,
  TFK_SLOT_GRAD = 1,       /* the `gradients/...` accumulators G (trainer.py:118-122) */
  TFK_SLOT_ADAM_M = 2,
  TFK_SLOT_ADAM_V = 3
};
int tfk_tensor_count(tfk_engine* e, int kind, int layer, size_t* count);
int tfk_tensor_get(tfk_engine* e, int kind, int slot, int layer, float* host, size_t count);
int tfk_tensor_set(tfk_engine* e, int kind, int slot, int layer, const float* host, size_t count);

enum { /* scalars */
  TFK_GLOBAL_STEP = 0,          /* trainer.py:98-100 */
  TFK_LEARNING_RATE_FACT = 1,   /* trainer.py:104-106 */
  TFK_INITIALISED_LAYERS = 2,   /* dnn.py:85-89 */
  TFK_ADAM_STEPS = 3,           /* Adam's own step count t (beta powers); NOT checkpointed by the reference */
  TFK_BATCH_LOSS = 4,           /* trainer.py:91-93  (read-only) */
  TFK_NUM_FRAMES = 5,           /* trainer.py:126-128 (read-only) */
  TFK_LEARNING_RATE = 6         /* current decayed rate (read-only) */
};
int tfk_scalar_get(tfk_engine* e, int which, double* value);
int tfk_scalar_set(tfk_engine* e, int which, double value);

/* ---- training: one micro-batch --------------------------------------------------------------- */

enum { /* flags */
  TFK_DEVICE_PTRS = 1,     /* X / y / out are device pointers (already resident in HBM) */
  TFK_LAST_MICROBATCH = 2, /* last accumulate before tfk_apply: fire the bucket callback per layer */
  TFK_LOG_DIV_PRIOR = 4,   /* tfk_posteriors: write log(posterior / prior) (nnet.py:280-286) */
  TFK_RAW_LOGITS = 8,      /* tfk_posteriors: write the logits (Classifier.__call__ output, dnn.py:108) */
  TFK_RAW_DEVICE = 16      /* the *_raw entry points: `raw` is a device pointer -- features that never left HBM, e.g. the output
                            * of tfk_feat_compute; everything else (y, utt_len, cmvn, out) stays a host pointer.  The producer's
                            * work must be complete, or ordered before the engine's stream (tfk_stream), when the call is made. */
};

/* Replaces `update_gradients_op.run(feed_dict)` (trainer.py:160-169, 325-332) for ONE micro-batch,
 * already flattened utterance-major as seq2nonseq would (seq_convertors.py:12-39):
 *   X [T, ldx] fp32 spliced frames, y [T] int32 pdf-ids.
 * Forward (train mode) + softmax cross-entropy (trainer.py:526-531) + backward;
 * G += g, batch_loss += loss, num_frames += T, BN moving-average updates (UPDATE_OPS). */
int tfk_accumulate(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, int flags);

/* The step BEFORE the path, moved onto the device: per-speaker mean/variance normalisation and the
 * +-context splice of processing/feature_reader.py:91-156 (apply_cmvn :109-115, splice :117-156).
 * raw [T, ldraw] = the UNSPLICED frames (raw_dim columns, input_dim = raw_dim * (2*context_width + 1)) of U
 * utterances back to back, utt_len[U] their frame counts (sum = T).
 * cmvn = NULL: raw is already normalised.  cmvn = [U, 2, raw_dim] fp32: row 0 of utterance u is its speaker's
 * mean, row 1 the standard deviation sqrt(E[x^2] - mean^2); the device computes (raw - mean) / std with the same
 * IEEE roundings as numpy's float32 subtract and divide, so the result is bit-identical to the host path.
 * Frames beyond an utterance's edges splice in as zeros, exactly like the host splice.  Only the unspliced frames
 * cross PCIe (11x less at context 5) and the spliced matrix is produced in HBM.  Host pointers (raw, y, utt_len,
 * cmvn) -- TFK_DEVICE_PTRS is not accepted; TFK_RAW_DEVICE makes `raw` alone a device pointer. */
int tfk_accumulate_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                       const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn, int flags);
int tfk_eval_accumulate_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                            const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn, int flags);

/*
 * Several micro-batches of ONE optimiser step in one call -- the same result as calling tfk_accumulate (tfk_accumulate_raw)
 * once per micro-batch in order, `flags` applying to the last one -- so that the engine can STACK them: the reference runs
 * update_gradients_op once per micro-batch (trainer.py:310-332), and at a thousand frames per micro-batch the contractions
 * leave the matrix pipes half idle and re-read every weight matrix k times per step.  A stacked pass multiplies all k
 * micro-batches at once (rows are independent in the affine maps; dW over the stacked rows IS G += g) and keeps per
 * micro-batch whatever couples the rows of one: the batch-norm statistics and their moving-average updates (in order),
 * batch-norm's backward, the dropout stream.  Equal to the sequential calls up to fp32 summation order inside the weight
 * gradients and the loss sum.  Chains the stacked pass does not cover (no batch norm, a nonlinearity other than ReLU,
 * L2Norm, layer-wise growth), micro-batches of more than 2048 frames and env TFK_STACK=0 run one after the other.
 *   tfk_accumulate_stacked      X[T, ldx] (host, or device with TFK_DEVICE_PTRS) = the micro-batches back to back,
 *                               seg_rows[k] frames each
 *   tfk_accumulate_stacked_raw  unspliced frames of U utterances back to back (as tfk_accumulate_raw), seg_utts[k]
 *                               utterances per micro-batch
 */
int tfk_accumulate_stacked(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, const int32_t* seg_rows,
                           int32_t k, int flags);
int tfk_accumulate_stacked_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                               const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn,
                               const int32_t* seg_utts, int32_t k, int flags);
/* Trainer.evaluate (reference neuralNetworks/trainer.py:356-441: one update_valid_loss run per micro-batch) over k micro-batches
 * in ONE call.  In evaluation mode the rows of a micro-batch are independent (batch norm normalises with the moving statistics,
 * dropout is the identity, L2Norm is row-wise), so the k runs are one pass of the GEMMs over the concatenated rows -- every
 * activation chain, no padding between segments; passes are cut at micro-batch boundaries once they hold TFK_EVAL_PASS_ROWS rows
 * (default 4096: the host-fed input of the next pass crosses PCIe under the kernels of this one).  batch_loss / num_frames end
 * up as k tfk_eval_accumulate[_raw] calls leave them, up to the fp32 order of the loss sum.  Arguments as the training entry
 * points above. */
int tfk_eval_accumulate_stacked(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, const int32_t* seg_rows,
                                int32_t k, int flags);
int tfk_eval_accumulate_stacked_raw(tfk_engine* e, const float* raw, int64_t ldraw, const int32_t* y, int32_t T,
                                    const int32_t* utt_len, int32_t U, int32_t context_width, const float* cmvn,
                                    const int32_t* seg_utts, int32_t k, int flags);

/* CTC loss instead of the frame-level cross-entropy (BASELINE configs[4]): what the reference's
 * CTCTrainer.compute_loss means to build with tf.nn.ctc_loss (trainer.py:533-570; its code cannot run, so this is a
 * clean-room implementation of the published forward-backward algorithm with that op's conventions: the blank is
 * the LAST class, repeated labels are merged).  X [T, ldx] = the flat utterance-major frames of U utterances,
 * utt_len[U] their frame counts (sum = T), labels = their label sequences back to back (values in [0, output_dim-1)),
 * label_len[U] the sequence lengths (at most 511 labels per utterance).  batch_loss += sum_u -log p(labels_u | X_u),
 * num_frames += number of labels (the reference counts TARGET lengths, trainer.py:126-133), then backward as
 * tfk_accumulate.  An utterance too short for its labels contributes +inf to the loss and a zero gradient.
 * Host pointers for utt_len / labels / label_len; X as tfk_accumulate (TFK_DEVICE_PTRS allowed). */
int tfk_accumulate_ctc(tfk_engine* e, const float* X, int64_t ldx, int32_t T, const int32_t* utt_len, int32_t U,
                       const int32_t* labels, const int32_t* label_len, int flags);
int tfk_eval_accumulate_ctc(tfk_engine* e, const float* X, int64_t ldx, int32_t T, const int32_t* utt_len, int32_t U,
                            const int32_t* labels, const int32_t* label_len, int flags);
/* The same on UNSPLICED frames (device-side CMVN + splice as tfk_accumulate_raw; the utterance boundaries are
 * shared): 11x less PCIe traffic, which is what bounds a host-fed CTC step on long utterances. */
int tfk_accumulate_ctc_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t T, const int32_t* utt_len, int32_t U,
                           int32_t context_width, const float* cmvn, const int32_t* labels, const int32_t* label_len,
                           int flags);
int tfk_eval_accumulate_ctc_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t T, const int32_t* utt_len,
                                int32_t U, int32_t context_width, const float* cmvn, const int32_t* labels,
                                const int32_t* label_len, int flags);

/* Replaces `[average_loss, apply_gradients_op]` + the three re-initialisations (trainer.py:336-352):
 * g = clip(G / num_frames, -1, 1); Adam; global_step += 1; returns batch_loss / num_frames (the
 * pre-update, train-mode loss); zeroes G, batch_loss, num_frames.  With data parallelism the host
 * all-reduces the reduce region (below) between the last tfk_accumulate and tfk_apply. */
int tfk_apply(tfk_engine* e, float* average_loss);
/* tfk_apply in two halves: tfk_apply_enqueue puts the whole optimiser step on the engine's streams without waiting for it,
 * tfk_apply_end (below) waits for the step's loss.  Between the two the host is free (the dispenser's prefetch). */
int tfk_apply_enqueue(tfk_engine* e);

/* Replaces `update_valid_loss.run(feed_dict)` (trainer.py:188-195, 433): eval-mode forward + loss. */
int tfk_eval_accumulate(tfk_engine* e, const float* X, int64_t ldx, const int32_t* y, int32_t T, int flags);
/* Replaces `average_loss.eval()` + re-initialisation (trainer.py:436-441). */
int tfk_eval_finish(tfk_engine* e, float* average_loss);

int tfk_halve_learning_rate(tfk_engine* e); /* trainer.py:141-142 */
int tfk_add_layer(tfk_engine* e);           /* control_ops['add']  (dnn.py:92) */
int tfk_init_last_layer(tfk_engine* e);     /* control_ops['init'] (dnn.py:114-120) */

/* ---- decoding -------------------------------------------------------------------------------- */

/* Replaces `Decoder.outputs.eval` (decoder.py:41-44, 70-71): eval-mode forward + softmax.
 * X [N, ldx] -> out [N, ldo] posteriors (or log(posterior/prior) with TFK_LOG_DIV_PRIOR). */
int tfk_posteriors(tfk_engine* e, const float* X, int64_t ldx, int32_t N, float* out, int64_t ldo, int flags);
int tfk_set_prior(tfk_engine* e, const float* prior, size_t count); /* prior.npy (nnet.py:241-244) */
/* As tfk_posteriors on unspliced frames of U utterances (device-side CMVN + splice as tfk_accumulate_raw;
 * several utterances per call = the batched decode of SURVEY 8f-2). */
int tfk_posteriors_raw(tfk_engine* e, const float* raw, int64_t ldraw, int32_t N, const int32_t* utt_len, int32_t U,
                       int32_t context_width, const float* cmvn, float* out, int64_t ldo, int flags);

/* ---- data parallelism (one engine per rank; the host owns the collective) --------------------- */

/* The reduce region is one contiguous fp32 span of the state: [ G (all layers) | batch_loss,
 * num_frames, num_microbatches, pad | BN moving-average increments ].  A SUM all-reduce of it across
 * ranks before tfk_apply makes B/U serial micro-batches on one GPU (trainer.py:310-332) and
 * one micro-batch on each of B/U GPUs the same computation.  Buckets: b in [0, L] is the weight-gradient span of
 * layer L - b (the order backward produces them), L + 1 every bias / beta gradient, L + 2 the scalars + the BN
 * increments.  With TFK_LAST_MICROBATCH they are announced in the order L + 2 (right after the loss, before
 * backward), 0 .. L, L + 1 -- also during layer-wise growth, when the layers above the active depth have a zero
 * gradient (they are announced right after bucket 0). */
int tfk_reduce_region(tfk_engine* e, void** device_ptr, size_t* num_floats);
/* `init_grads` / `init_loss` / `init_num_frames` (trainer.py:350-352) done eagerly: writes zeros over the whole
 * reduce region.  tfk_apply re-initialises lazily (the next step's first micro-batch overwrites G), so a rank
 * that contributes NO micro-batch to a step must call this before the all-reduce. */
int tfk_zero_accumulators(tfk_engine* e);
int tfk_reduce_bucket(tfk_engine* e, int bucket, size_t* offset_floats, size_t* num_floats);
int tfk_num_buckets(tfk_engine* e, int* n);

/* Called on the host from tfk_accumulate(TFK_LAST_MICROBATCH) right after the kernels that finish a
 * bucket have been enqueued, so the host can launch that bucket's all-reduce behind them while the
 * remaining backward keeps the GPU busy. */
typedef void (*tfk_bucket_fn)(void* user, int bucket);
int tfk_set_bucket_callback(tfk_engine* e, tfk_bucket_fn fn, void* user);

/* tfk_apply in three parts, for a host that overlaps the optimiser with its collectives: tfk_apply_begin needs
 * bucket L + 2 reduced (learning rate, BN moving averages, loss hand-over); tfk_apply_span runs mean -> clip ->
 * Adam on the parameters [offset, offset + n) of the arena (same offsets as the reduce region), callable as soon as
 * THAT span of G is reduced; tfk_apply_end returns the average loss and re-initialises the accumulators.
 * tfk_apply == begin, span(0, P), end. */
int tfk_apply_begin(tfk_engine* e);
int tfk_apply_span(tfk_engine* e, size_t offset_floats, size_t num_floats);
int tfk_apply_end(tfk_engine* e, float* average_loss);

/* Called on the host right before the kernels that READ the parameters of hidden layer `layer` (0 .. L - 1), of the
 * output layer (L) or of every layer (-1: tensor get / set, the bf16 shadow rebuild) are enqueued.  A host that writes
 * parameters asynchronously -- the all-gather of the sharded exchange step, still in flight when the next step's
 * forward pass starts -- makes the engine stream wait for exactly that write here. */
typedef void (*tfk_layer_fn)(void* user, int layer);
int tfk_set_layer_callback(tfk_engine* e, tfk_layer_fn fn, void* user);

/* Sharded exchange step (reduce-scatter -> tfk_apply_span on this rank's share -> all-gather of the updated
 * parameters written straight into the arena by the collective): tells the engine that parameters changed behind
 * the optimiser's back, so the bf16 weight shadow of the mixed-precision mode is rebuilt before its next use.
 * Call between the last tfk_apply_span and tfk_apply_end (or at any other time). */
int tfk_params_touched(tfk_engine* e);

/* Mixed precision + sharded exchange: the bf16 shadow of the weight matrices.  When every weight matrix has a leading
 * dimension that is a multiple of 8, the shadow mirrors the fp32 parameter arena element for element (`mirrors_arena` = 1,
 * `num_elems` = the length of the weight part of the arena) and lives at the tail of the state arena (tfk_state_bytes
 * counts it), so the host can all-gather each rank's freshly updated shard of the SHADOW (tfk_apply_span writes it with
 * the update) instead of the fp32 parameters: half the bytes, and the next forward pass -- which reads only the shadow --
 * waits for it layer by layer through tfk_set_layer_callback.  The fp32 masters then stay valid only on the rank that
 * owns the span until the host gathers them (checkpoints, tensor get / set).  Otherwise num_elems = 0. */
int tfk_shadow_region(tfk_engine* e, void** device_ptr, size_t* num_elems, int* mirrors_arena);
/* (ABI 8) Emulated fp32 + sharded exchange: the tiled three-plane twin of weight matrix `layer` (0 .. L) -- what the
 * contractions read and what tfk_apply_span writes with the update.  Rows come in pairs (csrc/x3_layout.h), so rank r's rows
 * [r * rows / world, (r + 1) * rows / world) are the r-th of `world` equal, contiguous pieces of the region whenever rows is a
 * multiple of 2 * world: the exchange may then all-gather the owner-written PLANES of the matrix (6 B per weight) in place of
 * its fp32 parameters (4 B) + a local rebuild (tfk_twins_from_params); the fp32 masters of the matrix then stay valid on their
 * owner only, as with the bf16 shadow.  bytes = 0 when the arithmetic has no such twins or the optimiser does not write them.
 * tfk_param_checksum(e, 3, ..) sums every twin. */
int tfk_twin_region(tfk_engine* e, int layer, void** device_ptr, size_t* bytes, int* rows);
/* (ABI 7) The fp32 parameters [offset, offset + n) -- whole weight matrices -- were written behind the optimiser's back (a
 * sharded exchange all-gathered them) and what the contractions READ is derived from them: under TFK_DTYPE_F32X3 the tiled
 * three-plane twins (csrc/x3_layout.h).  Rebuilds the twins of those matrices on `stream` (NULL: the engine's) -- the exchange
 * calls it on the stream the gather ran on, right behind it, so that the rebuild of layer l hides under the forward pass of
 * the layers below instead of the whole set being rebuilt, all gathers awaited, in front of the next pass.  *current = 1: what
 * the contractions read is current for that span (exact fp32: always; emulated fp32: rebuilt); 0: nothing was done (mixed
 * precision without this path, or twins that were stale anyway) and the caller falls back to tfk_params_touched. */
int tfk_twins_from_params(tfk_engine* e, size_t offset, size_t n, void* stream, int* current);
/* Between tfk_apply_begin and tfk_apply_end: does tfk_apply_span write the bf16 shadow along with the parameters
 * (mixed precision, arena-mirroring shadow that was current when the step began)? */
int tfk_apply_writes_shadow(tfk_engine* e, int* direct);
/* Position-weighted 64-bit integer checksum of the fp32 parameter arena (which = 0), of the arena-mirroring bf16
 * shadow (which = 1) or of the fp32 bias / beta vectors alone (which = 2), taken in engine-stream order; synchronises.  Replicas of a data-parallel job compare it after
 * the first optimiser steps: a mis-ordered collective would otherwise train on stale weights silently. */
int tfk_param_checksum(tfk_engine* e, int which, uint64_t* value);

/* Number of micro-batches of this optimiser step that ranks AFTER this one process: weights this
 * rank's BN moving-average increment by bn_decay^later so the all-reduced result equals the
 * reference's sequential per-micro-batch EMA updates. */
int tfk_set_later_microbatches(tfk_engine* e, int32_t later);
/* The fp32 parameter arena [P] (same float offsets as the gradient part of the reduce region): what a sharded exchange
 * step all-gathers into. */
int tfk_param_region(tfk_engine* e, void** device_ptr, size_t* num_floats);
/* (ABI 8) Adam's first and second moments [P] each, same float offsets as the parameter arena.  Under a sharded exchange every
 * rank updates the moments of its OWN shards only -- the optimiser state is sharded with the optimiser -- so whatever changes
 * the assignment of shards to ranks (tfk_comm_set_bucket_bytes, tfk_comm_set_gather) gathers them first. */
int tfk_moment_regions(tfk_engine* e, void** adam_m, void** adam_v, size_t* num_floats);

/* ---- the exchange step inside the library: RCCL over xGMI, no host language in the step ------------------------
 *
 * Replaces, for N > 1 GPUs, what a single reference process does between `update_gradients_op` and
 * `apply_gradients_op` (trainer.py:165-184): the micro-batches of a step run on different GPUs, so G, batch_loss,
 * num_frames and the BN moving-average increments are SUMMED over the ranks before mean -> clip -> Adam.  A tfk_comm
 * attaches to one engine, installs itself behind the engine's bucket / layer hooks (tfk_set_bucket_callback /
 * tfk_set_layer_callback: do not set your own while one is attached) and from then on
 *     tfk_accumulate*(..., TFK_LAST_MICROBATCH)  launches the collectives while backward is still being enqueued,
 *     tfk_comm_apply                             replaces tfk_apply          (Adam per reduced span, parameter gathers),
 *     tfk_comm_eval_finish                       replaces tfk_eval_finish    (scalar tail summed over the ranks),
 *     tfk_comm_idle                              a rank without a micro-batch in this step contributes zeros.
 * Every rank makes the same calls in the same order.  Exchange modes as tfkaldi_amd/dataparallel.py (which drives the
 * same protocol through torch.distributed for backends without RCCL):
 *   TFK_EXCHANGE_SHARDED    in-place reduce-scatter of every coalesced span of weight gradients, Adam on this rank's
 *                           1/world of it, in-place all-gather of the updated parameters -- in mixed precision of the
 *                           bf16 shadow (2 B per parameter; the fp32 masters of a span then live on its owner until
 *                           tfk_comm_gather_masters) -- consumed layer by layer by the next forward pass;
 *   TFK_EXCHANGE_ALLREDUCE  SUM all-reduce of every span, full Adam on every rank.
 * Streams: collectives launched while backward still runs, and the parameter gathers the next forward pass consumes layer
 * by layer, go to a comm stream the tfk_comm owns (ordered against the engine stream by events); the collectives launched
 * behind the LAST backward kernel and the gather the next forward pass reads first go to the engine stream itself
 * (nothing is left to overlap with, and a round trip through another stream costs ~26 us; env TFK_DP_INLINE_TAIL=0:
 * everything on the comm stream).  RCCL orders the operations of one communicator itself.
 * bucket_bytes: adjacent gradient buckets are coalesced until a collective carries at least this much (0: default,
 * 32 MiB -- each collective under backward costs the step a fixed 6-10 us, but the next forward pass cannot start a layer before
 * the whole gather covering it has arrived and the span still coalescing when backward ends is exposed: csrc/exchange.hip).
 * Bootstrap: EVERY rank first calls tfk_comm_available (local, no collective: RCCL loadable, engine and mode acceptable) and
 * the host agrees on the answers (one MIN all-reduce over whatever process group it already has) -- tfk_comm_create is
 * collective (ncclCommInitRank), so a rank that could not even load RCCL must be known BEFORE the others enter it and
 * block.  Then rank 0 calls tfk_comm_unique_id, the host distributes the bytes any way it likes (torch.distributed, MPI, a
 * file), and every rank calls tfk_comm_create with them. */
enum { TFK_EXCHANGE_SHARDED = 0, TFK_EXCHANGE_ALLREDUCE = 1 };
typedef struct tfk_comm tfk_comm;
int tfk_comm_available(tfk_engine* e, int mode);
int tfk_comm_unique_id(void* id, size_t capacity, size_t* size);  /* 128 bytes */
int tfk_comm_create(tfk_engine* e, const void* id, size_t id_size, int rank, int world, int mode, size_t bucket_bytes,
                    tfk_comm** out);
int tfk_comm_destroy(tfk_comm* c);  /* before tfk_destroy of its engine */
int tfk_comm_info(tfk_comm* c, int* rank, int* world, int* mode, int* gathers_shadow);
const char* tfk_comm_backend(tfk_comm* c);  /* "rccl" | "loopback" */
int tfk_comm_apply(tfk_comm* c, float* average_loss);
/* the same in two halves (as tfk_apply_enqueue / tfk_apply_end): _enqueue launches the tail collectives, Adam on this rank's
 * spans and the parameter gathers; _end waits for the loss (and runs the replica check of the first sharded steps). */
int tfk_comm_apply_enqueue(tfk_comm* c);
int tfk_comm_apply_end(tfk_comm* c, float* average_loss);
int tfk_comm_eval_finish(tfk_comm* c, float* average_loss);
/* (tests / diagnostics) launch what is still coalescing and make the engine stream wait for every collective of the step,
 * WITHOUT the optimiser: the reduce region then holds the summed gradients -- of a reduce-scattered span only this rank's
 * 1/world.  tfk_comm_apply may follow. */
int tfk_comm_finish_reduce(tfk_comm* c);
int tfk_comm_idle(tfk_comm* c);
/* make the engine stream wait for parameter gathers still in flight (before the state is read behind the engine's back) */
int tfk_comm_drain(tfk_comm* c);
/* mixed-precision sharded exchange: are the fp32 masters currently valid on their owners only? / COLLECTIVE: bring them
 * home (all-gather of every sharded span) before a checkpoint or a tensor get / set */
int tfk_comm_masters_stale(tfk_comm* c, int* stale);
int tfk_comm_gather_masters(tfk_comm* c);
/* collectives of the last completed step: counts by kind and, in launch order, the gradient spans as triples
 * (offset in floats, floats, 1 = reduce-scattered / 0 = all-reduced); spans holds 3 * capacity values */
int tfk_comm_last_step(tfk_comm* c, int* reduce_scatters, int* all_gathers, int* all_reduces, size_t* spans, int capacity,
                       int* num_spans);
/* (ABI 8) HOW a reduce-scattered span and a parameter gather travel (sharded mode; reference seam trainer.py:165-184 -- the sum
 * `G += g` over the micro-batches of a step, here over ranks):
 *   TFK_ALGO_RCCL    ncclReduceScatter / ncclAllGather: RCCL picks the algorithm (ring / tree over its channels) and with it the
 *                    order the ranks' contributions are added in;
 *   TFK_ALGO_DIRECT  the collective spelled out for a full mesh of point-to-point xGMI links: one grouped ncclSend / ncclRecv
 *                    per peer -- sub-span q of every rank goes straight to rank q over the link between them, all world - 1
 *                    links busy at once -- and the OWNER adds the world contributions in fp32 in RANK ORDER, i.e. in the order a
 *                    single process adds its micro-batches: the reduced shard equals the serial run's G bit for bit.  The
 *                    gather is the same movement backwards (own shard to every peer, in place).
 * Wire format of the reduce-scatter: TFK_WIRE_FP32, or TFK_WIRE_BF16 (direct movement at half the bytes; the owner's own
 * contribution stays exact).  Environment at attach: TFK_DP_ALGO = rccl (default) | direct | auto, TFK_DP_WIRE = fp32 | bf16.
 * `auto` (more than one RCCL rank): tfk_comm_create times both algorithms on scratch memory of a span's size (tfk_comm_tune,
 * collective) and keeps the faster one per operation -- the slowest rank's time decides, identically on every rank.  It is
 * opt-in until a multi-GPU node has run it: bench.py --gpus N runs the same tuning pass and an algorithm x wire A/B BEHIND its
 * timed region and reports both, so the first scaling line shows what `auto` would have chosen and gained.
 * tfk_comm_set_exchange (-1 = keep): between steps only, every rank with the same arguments.
 * tfk_comm_get_exchange: what is in force; chosen_by 0 default / 1 environment / 2 tuned / 3 set; tune_us[4] = microseconds of
 * reduce-scatter rccl, direct, all-gather rccl, direct as tuned (0: never tuned). */
enum { TFK_ALGO_RCCL = 0, TFK_ALGO_DIRECT = 1 };
enum { TFK_WIRE_FP32 = 0, TFK_WIRE_BF16 = 1 };
int tfk_comm_set_exchange(tfk_comm* c, int algo, int wire);
int tfk_comm_get_exchange(tfk_comm* c, int* algo_reduce_scatter, int* algo_all_gather, int* wire, int* chosen_by, double* tune_us);
int tfk_comm_tune(tfk_comm* c, size_t floats, int iters); /* COLLECTIVE */
/* (ABI 8) WHAT is gathered under the emulated fp32 arithmetic (sharded mode).  Default (`params`): the fp32 parameters of every
 * span, and every rank rebuilds the three-plane twins of what it received (tfk_twins_from_params behind each gather).  `planes`
 * (env TFK_DP_GATHER=planes at attach, or this call -- COLLECTIVE, between steps): the sharding unit becomes the weight matrix --
 * rank r owns rows [r R / world, (r + 1) R / world) of every matrix of a span, the span's collectives are launched as one group
 * -- and for every matrix whose rows divide by 2 * world the twin rows the owner's Adam wrote are gathered (tfk_twin_region: 6 B
 * per weight on the wire, nothing rebuilt); the fp32 masters of those matrices stay with their owners until
 * tfk_comm_gather_masters (tfk_comm_masters_stale says so; tfk_comm_info: gathers_shadow = 2).  Switching back gathers them.
 * dataparallel.exchange_model prices both; bench.py --gpus N measures both (`exchange_ab`). */
int tfk_comm_set_gather(tfk_comm* c, int planes);
/* (ABI 8) the coalescing threshold of tfk_comm_create's `bucket_bytes`, changed between steps (every rank alike; 0 = the 32 MiB
 * default): how many weight matrices one collective carries is the first thing to tune on real links -- bench.py sweeps it.
 * COLLECTIVE: another span cut assigns shards to other ranks, so masters left with their owners are gathered first. */
int tfk_comm_set_bucket_bytes(tfk_comm* c, size_t bucket_bytes);
/* (ABI 8) Device time per phase of the exchange step, for diagnosis (bench.py --gpus N: `exchange_phases`): between
 * tfk_comm_timing(c, 1) and tfk_comm_timing_read every phase is bracketed by timing events on the stream it runs on (each
 * record costs that stream a few microseconds: a diagnostic pass, not the one a rate is quoted from).  ms_per_step[k], averaged
 * over the steps completed in between: 0 reduce-scatters (pack / exchange / owner's sum included), 1 all-reduces (vectors, scalar
 * tail), 2 engine-stream time from the last backward kernel to the first optimiser kernel (what the exchange leaves exposed in
 * front of Adam), 3 Adam on this rank's spans, 4 parameter gathers, 5 three-plane twin rebuilds behind them, 6 engine-stream
 * time the next forward pass spends waiting for gathers (the first span's gather + rebuild run on that stream and count in
 * full).  tfk_comm_timing_read synchronises both streams and switches the timing off. */
int tfk_comm_timing(tfk_comm* c, int on);
int tfk_comm_timing_read(tfk_comm* c, double* ms_per_step, int capacity, long* steps);
/* Tests: a group of `world` engines of ONE process on ONE device; each rank is driven by its own host thread and the
 * collectives are a rendezvous + plain kernels.  This is how the protocol above runs at world 2 / 4 / 8 on a single-GPU
 * box (RCCL refuses two ranks on one device). */
typedef struct tfk_loopback tfk_loopback;
int tfk_loopback_create(int world, tfk_loopback** out);
int tfk_loopback_destroy(tfk_loopback* group);
int tfk_comm_create_loopback(tfk_engine* e, tfk_loopback* group, int rank, int mode, size_t bucket_bytes, tfk_comm** out);

/* ---- streams, profiling, debugging ------------------------------------------------------------- */

int tfk_synchronize(tfk_engine* e);
int tfk_stream(tfk_engine* e, void** hip_stream);

typedef struct tfk_kernel_stat {
  char name[48];
  int64_t launches;
  double total_ms;  /* HIP-event time on the engine stream */
  double flops;     /* algorithmic FLOPs of those launches */
  double bytes;     /* algorithmic bytes of those launches */
} tfk_kernel_stat;
/* Bracket every kernel launch with HIP events (on the engine stream) until tfk_profile_end. */
int tfk_profile_begin(tfk_engine* e);
int tfk_profile_end(tfk_engine* e, tfk_kernel_stat* stats, int capacity, int* count);

enum { /* debug tensors of the LAST accumulate / eval / posteriors call */
  TFK_DBG_LOGITS = 0,      /* [T, O] logits; after tfk_accumulate: dLogits = softmax - onehot */
  TFK_DBG_HIDDEN = 1,      /* [T, H] output of hidden layer `layer` */
  TFK_DBG_DROPOUT_MASK = 2,/* [T, H] 0/1 keep mask of hidden layer `layer` (regenerated) */
  TFK_DBG_PREACT = 3,      /* [T, H] affine output z of hidden layer `layer` (what batch norm normalises) */
  TFK_DBG_BN_MEAN = 4,     /* row 0 of [T, H]: batch mean of z's columns as the backward pass reads it (training calls) */
  TFK_DBG_BN_RSTD = 5      /* row 0 of [T, H]: 1 / sqrt(batch variance + eps) */
};
int tfk_debug_fetch(tfk_engine* e, int what, int layer, float* host, size_t count);

/* Stand-alone fp32 GEMM on device pointers (tests / tools): layout 0 NN, 1 NT, 2 TN (gemm_f32.h). */
int tfk_gemm_f32(void* stream, int layout, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 int M, int N, int K, const float* bias, int epi, int tile_config);
/* Stand-alone bf16 GEMM (fp32 accumulate / result) on device pointers: A, B hold bfloat16 bit patterns with
 * leading dimensions (in elements) that are multiples of 8 and zero padding (gemm_bf16.h). */
int tfk_gemm_bf16(void* stream, int layout, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc,
                  int M, int N, int K, const float* bias, int epi);
/* Stand-alone fp32-emulating GEMM (TFK_DTYPE_F32X3's contraction; gemm_bf16.h: gemm_bf16x3) on device pointers: A and B are given
 * as three bf16 planes each in the TILED layout of csrc/x3_layout.h (ld a multiple of 32; 2-row x 32-column units of three 128-byte
 * lines, one per plane: element (r, c) of plane q at ((r / 2) * (ld / 32) + c / 32) * 192 + 64 q + 32 (r % 2) + c % 32;
 * ceil(rows / 2) * (ld / 32) * 192 elements in all); tfk_split3 makes such a twin from an fp32 matrix [rows, lds]:
 * src == plane 0 + plane 1 + plane 2 exactly, padding columns zero.  tfk_gemm_bf16x3_dual: the backward pair of a layer in one
 * launch (as tfk_gemm_bf16_dual).  Tests and tools.  (ABI 7: the separate-plane form of ABI 6 -- `a_plane` / `b_plane`
 * arguments -- is gone.) */
int tfk_split3(void* stream, const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols);
/* tests: mark a ticket of the engine's split-K workspace as taken, as an interrupted launch would leave it.  The next step whose
 * contractions split their K must then FAIL (tfk_apply / tfk_eval_finish / tfk_posteriors return non-zero, tfk_last_error names
 * the timeout) -- a block that waits ~1 s for its partner's partial sums reports it through mapped memory -- and the step after
 * that must work again. */
int tfk_debug_poison_splitk(tfk_engine* e);
int tfk_gemm_bf16x3(void* stream, int layout, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc, int M,
                    int N, int K, const float* bias, int epi);
int tfk_gemm_bf16x3_dual(void* stream, const uint16_t* A_nt, int lda_nt, const uint16_t* B_nt, int ldb_nt, float* C_nt,
                         int ldc_nt, int M_nt, int N_nt, int K_nt, const uint16_t* A_tn, int lda_tn, const uint16_t* B_tn,
                         int ldb_tn, float* C_tn, int ldc_tn, int M_tn, int N_tn, int K_tn, int epi_tn);
/* The backward pair of one layer in ONE launch (gemm_bf16.h: gemm_bf16_dual): C_nt[M_nt, N_nt] = A_nt . B_nt^T and
 * C_tn[M_tn, N_tn] (+)= A_tn^T . B_tn (epi_tn: 0 or 2 = accumulate).  Fails when the pair of shapes is not eligible
 * (tfk_gemm_bf16_dual_config == 0).  Block geometry: env TFK_BF16_DUAL_CFG (3: 128x64, 4: 128x128, 5: 256x128,
 * 8: 256x128 for the NT half + 256x256 for the TN half, 0: off). */
int tfk_gemm_bf16_dual(void* stream, const uint16_t* A_nt, int lda_nt, const uint16_t* B_nt, int ldb_nt, float* C_nt,
                       int ldc_nt, int M_nt, int N_nt, int K_nt, const uint16_t* A_tn, int lda_tn, const uint16_t* B_tn,
                       int ldb_tn, float* C_tn, int ldc_tn, int M_tn, int N_tn, int K_tn, int epi_tn);
int tfk_gemm_bf16_dual_config(int M_nt, int N_nt, int M_tn, int N_tn);
/* Tile configuration of the bf16 GEMM (gemm_bf16.h: 0-2 register-staged, 3-8 LDS-DMA staged): force one for every
 * later call (cfg < 0 restores the heuristic) / ask which one the heuristic gives an [M, N] result.  Tools and
 * tests only. */
int tfk_gemm_bf16_force_config(int cfg);
int tfk_gemm_bf16_config(int M, int N);

/* ---- feature computation: wav samples -> fbank / mfcc / ssc (+ deltas), CMVN statistics ------------ */
/*
 * The step in FRONT of the feature reader: processing/feat.py:7-69 (FeatureComputer),
 * processing/base.py:39-284 (mfcc, fbank, logfbank, ssc, lifter, deriv, delta, ddelta), processing/sigproc.py:33-191
 * (preemphasis, framesig, magspec, powspec) and processing/prepare_data.py:80-118 (compute_cmvn), computed in
 * float64 as the reference does (numpy's default), for a whole BATCH of utterances per call.
 *
 * All pointers of the compute calls are DEVICE pointers; work is ordered on `stream` (a hipStream_t; NULL = the
 * default stream).  A plan may be used from one thread at a time.
 *
 * Signals of the batch are concatenated: utterance u owns samples [sig_off[u], sig_off[u+1]) and frames
 * [frame_off[u], frame_off[u+1]) -- the caller counts frames as sigproc.py:49-55 does (1 if len <= frame_len, else
 * 1 + ceil((len - frame_len) / frame_step)) -- and frame t of an utterance covers its samples
 * [t*frame_step, t*frame_step + frame_len), zero-padded past the end of the utterance (sigproc.py:57-66), after the
 * pre-emphasis y[0] = x[0], y[i] = x[i] - preemph * x[i-1] (sigproc.py:180-191).  Frames longer than nfft are
 * truncated and shorter ones zero-padded by the transform (numpy.fft.rfft(frames, nfft): sigproc.py:138).
 */
typedef struct tfk_feat tfk_feat;

enum { TFK_FEAT_FBANK = 0, TFK_FEAT_MFCC = 1, TFK_FEAT_SSC = 2,        /* feat.py:21-28: log-fbank, mfcc, ssc */
       TFK_FEAT_FBANK_RAW = 3 };  /* base.fbank (base.py:59-98): filterbank energies and frame energy WITHOUT the log */
enum { TFK_DYN_NODELTA = 0, TFK_DYN_DELTA = 1, TFK_DYN_DDELTA = 2 };   /* feat.py:30-37 */
enum { TFK_SAMPLE_I16 = 0, TFK_SAMPLE_F64 = 1,  /* scipy.io.wavfile's int16 / any other integer type promoted to float64 */
       TFK_SAMPLE_F32 = 2 };  /* float32 wav files: numpy keeps `signal[1:] - coeff * signal[:-1]` in float32 for them (the Python
                               * float is cast down), so the pre-emphasis is done in float32 and only then widened */
enum { TFK_STAGE_FRAMES = 1, TFK_STAGE_MAGSPEC = 2, TFK_STAGE_POWSPEC = 3 };

typedef struct tfk_feat_config {
  int32_t struct_size;     /* sizeof(tfk_feat_config) */
  int32_t device;          /* HIP device ordinal */
  int32_t kind;            /* TFK_FEAT_* */
  int32_t dynamic;         /* TFK_DYN_* */
  int32_t frame_len;       /* samples per frame:  round(winlen * rate)  (sigproc.py:50) */
  int32_t frame_step;      /* samples per step:   round(winstep * rate) (sigproc.py:51) */
  int32_t nfft;            /* transform length: a power of two in [32, 4096] */
  int32_t nfilt;           /* mel filters (<= nfft / 2) */
  int32_t numcep;          /* TFK_FEAT_MFCC: cepstra kept (<= nfilt) */
  int32_t include_energy;  /* append log(frame energy) as the last static column (feat.py:61-62) */
  double preemph;          /* pre-emphasis coefficient */
} tfk_feat_config;

/* Tables are HOST pointers, copied: filterbank[nfilt][nfft/2+1] (base.get_filterbanks), bin_weight[nfft/2+1]
 * (TFK_FEAT_SSC: numpy.linspace(1, rate/2, nfft/2+1), base.py:151), dct[nfilt][numcep] and lifter[numcep]
 * (TFK_FEAT_MFCC: the orthonormal DCT-II matrix and the lifter weights, base.py:55-56,226-246). */
int tfk_feat_create(const tfk_feat_config* cfg, const double* filterbank, const double* bin_weight, const double* dct,
                    const double* lifter, tfk_feat** out);
int tfk_feat_destroy(tfk_feat* f);
/* columns of the feature matrix: (nfilt or numcep, + 1 with the energy) * (1, 2 or 3) */
int tfk_feat_dim(const tfk_feat* f, int32_t* dim);
/* FeatureComputer.__call__ (feat.py:42-69) for a batch: out[n_frames][ld_out] as float32 (what ArkWriter stores,
 * ark.py:202) or float64 (what the reference's functions return). */
int tfk_feat_compute(tfk_feat* f, void* stream, const void* signal, int sample_type, const int64_t* sig_off,
                     const int64_t* frame_off, int32_t n_utts, int64_t n_frames, void* out, int64_t ld_out, int out_f64);
/* Intermediate results of the same pipeline as float64: TFK_STAGE_FRAMES [n_frames][frame_len] (sigproc.framesig of
 * the pre-emphasised signal), TFK_STAGE_MAGSPEC / TFK_STAGE_POWSPEC [n_frames][nfft/2+1] (sigproc.py:125-153). */
int tfk_feat_stage(tfk_feat* f, void* stream, int stage, const void* signal, int sample_type, const int64_t* sig_off,
                   const int64_t* frame_off, int32_t n_utts, int64_t n_frames, double* out, int64_t ld_out);
/* base.deriv / delta / ddelta (base.py:248-284) on a float64 matrix x[n_rows][ld_x] of `dim` columns whose rows
 * [row_off[u], row_off[u+1]) are one utterance each (scipy.ndimage.convolve1d's 'reflect' boundary applies per
 * utterance): out[n_rows][ld_out] = [x | d x | d d x] limited to 1 + dynamic blocks.  deriv_only: out = d x. */
int tfk_feat_dynamic(void* stream, const double* x, int64_t ld_x, int32_t dim, const int64_t* row_off, int32_t n_utts,
                     int64_t n_rows, int dynamic, int deriv_only, void* out, int64_t ld_out, int out_f64);
/* sigproc.deframesig (sigproc.py:69-123): overlap-add of frames[n_frames][ld] (frame_len columns used) back into a signal
 * of (n_frames - 1) * frame_step + frame_len samples; every sample is divided by the sum of the window values (+ 1e-15 per
 * frame, as there) of the frames that cover it, both sums taken in frame order.  win[frame_len] = NULL: rectangular. */
int tfk_deframesig(void* stream, const double* frames, int64_t ld, int64_t n_frames, int32_t frame_len, int32_t frame_step,
                   const double* win, double* out);
/* The tail of sigproc.logpowspec (sigproc.py:170-178), in place on n power-spectrum values: 10 log10(max(p, 1e-30)),
 * minus the maximum over all of them when norm != 0.  scratch: device memory for 1024 doubles. */
int tfk_logpow(void* stream, double* p, int64_t n, int norm, double* scratch);
/* compute_cmvn (prepare_data.py:80-118): speaker s owns the utterances [spk_off[s], spk_off[s+1]) of the list
 * (utt_row[q], utt_len[q]) = first row and row count in feats[.][ld] (float32, `dim` columns);
 * stats[s] = [[sum x | count], [sum x^2 | 0]] as [2][dim+1] doubles, the sums accumulated row after row in float32
 * exactly as numpy reduces the reference's float32 matrix. */
int tfk_cmvn_stats(void* stream, const float* feats, int64_t ld, int32_t dim, const int64_t* spk_off,
                   const int64_t* utt_row, const int64_t* utt_len, int32_t n_spk, double* stats);

#ifdef __cplusplus
}
#endif
#endif /* TFKALDI_HIP_H */
