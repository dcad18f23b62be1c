// CTC loss on the device (BASELINE configs[4]): what the reference's CTCTrainer means to build with
// tf.nn.ctc_loss (neuralNetworks/trainer.py:558-570) -- time-major logits, blank = LAST class, repeated labels merged,
// per-utterance loss -log p(labels | logits) -- as a clean-room implementation of the published forward-backward
// recursion (Graves et al. 2006) in log space.  The reference's own method cannot run, so there is no reference
// behaviour to match; the checker is oracle/ctc_oracle.py, itself pinned against torch's CPU ctc_loss.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace tfk {

struct CtcBatch {
  const float* logits;     // [T, ld] pre-softmax outputs of the flat, utterance-major frames
  int ld;
  float* post;             // [T, ld] scratch: softmax(logits)
  float* lse;              // [T]     scratch: log-sum-exp of every row
  const int32_t* seg;      // [U + 1] first row of every utterance
  const int32_t* labels;   // concatenated label sequences, values in [0, O - 1)
  const int32_t* lab_off;  // [U + 1] first label of every utterance
  int U, T, O;
  int sext;                // row stride of lp / ab: >= 2 * (longest label sequence) + 1, a multiple of 64 * R
  float* lp;               // [T, sext] scratch: log p_t(state s), states = blank, l_0, blank, l_1, ..., blank
  float* ab;               // [T, sext] scratch: forward variables alpha~ (relative to off)
  float* bb;               // [T, sext] scratch: backward variables beta~ (relative to offb); gradient only
  double* off;             // [T]     scratch: per-frame offset of the re-centred forward variables
  double* offb;            // [T]     scratch: the same for the backward variables
  double* logz;            // [U]     scratch: log p(labels) in double
  float* utt_loss;         // [U] out: -log p, +inf for an utterance too short for its labels
};

// Longest label sequence the wave-per-utterance recursion handles (64 lanes x 16 states).
constexpr int kCtcMaxLabels = (64 * 16 - 1) / 2;
// sext for a batch whose longest label sequence has max_labels entries
int ctc_state_stride(int max_labels);

// with_grad: dlogits [T, ld] <- softmax - state posteriors folded onto the classes (zero rows for utterances with an
// infinite loss); tw: bf16 twin of dlogits (mixed-precision mode).  Without with_grad only utt_loss is produced.
void ctc_loss_grad(hipStream_t s, const CtcBatch& b, float* dlogits, int with_grad, Twin tw);

// scalars[0] (+)= sum of the utterance losses, scalars[1] (+)= number of labels (trainer.py:126-133 counts TARGET
// lengths), scalars[2] (+)= 1
void ctc_loss_reduce(hipStream_t s, const float* utt_loss, const int32_t* lab_off, int U, float* scalars, bool overwrite);

}  // namespace tfk
