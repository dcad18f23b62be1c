"""CPU oracle of the CTC loss (test infrastructure: imported by tests/ only, never by the product).

What it restates: the loss the reference's CTCTrainer means to build -- `tf.nn.ctc_loss(tf.pack(logits),
sparse_targets, logit_seq_length)` (neuralNetworks/trainer.py:558-570) -- with TensorFlow's conventions for that op:
time-major logits, the BLANK is the LAST class (num_classes - 1), repeated labels are merged
(ctc_merge_repeated=True), labels are not pre-collapsed, and the per-sequence loss is -log p(labels | logits).
The reference's own method cannot run (it iterates `range(len(batch_size))` over an int, fills the sparse values from
the logits and returns nothing), so there is no reference behaviour to pin: **parity unpinned**.  The
arithmetic below is the published algorithm (Graves et al. 2006, "Connectionist Temporal Classification", eqs. 6-16,
in log space) and is pinned against an independent implementation, torch.nn.functional.ctc_loss on CPU
(tests/test_ctc_oracle.py).

Trainer conventions that DO come from the reference (neuralNetworks/trainer.py:126-133, 165-169, 174-175, 198):
batch_loss is the SUM of the per-utterance losses and num_frames counts TARGET labels, so the reported average loss
and the mean gradient divide by the number of labels of the step.
"""
import numpy as np


def _logsumexp(*xs):
    m = np.max(xs)
    if not np.isfinite(m):
        return m
    return m + np.log(sum(np.exp(x - m) for x in xs))


def log_softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def ctc_loss_and_grad(logits, labels, blank=None):
    """logits [T, O] float64, labels [S] ints in [0, O-1) -> (loss, d loss / d logits [T, O]).
    An utterance too short for its labels has loss +inf and a zero gradient."""
    logits = np.asarray(logits, dtype=np.float64)
    T, O = logits.shape
    blank = O - 1 if blank is None else blank
    labels = [int(x) for x in labels]
    assert all(0 <= x < O and x != blank for x in labels)
    ext = [blank]
    for x in labels:
        ext += [x, blank]
    n = len(ext)
    logp = log_softmax(logits)
    ninf = -np.inf

    def skip_ok(s):  # transition s-2 -> s allowed
        return s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]

    alpha = np.full((T, n), ninf)
    alpha[0, 0] = logp[0, ext[0]]
    if n > 1:
        alpha[0, 1] = logp[0, ext[1]]
    for t in range(1, T):
        for s in range(n):
            terms = [alpha[t - 1, s]]
            if s >= 1:
                terms.append(alpha[t - 1, s - 1])
            if skip_ok(s):
                terms.append(alpha[t - 1, s - 2])
            alpha[t, s] = _logsumexp(*terms) + logp[t, ext[s]]
    log_z = _logsumexp(alpha[T - 1, n - 1], alpha[T - 1, n - 2]) if n > 1 else alpha[T - 1, 0]
    if not np.isfinite(log_z):
        return np.inf, np.zeros_like(logits)
    beta = np.full((T, n), ninf)
    beta[T - 1, n - 1] = logp[T - 1, ext[n - 1]]
    if n > 1:
        beta[T - 1, n - 2] = logp[T - 1, ext[n - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(n):
            terms = [beta[t + 1, s]]
            if s + 1 < n:
                terms.append(beta[t + 1, s + 1])
            if s + 2 < n and skip_ok(s + 2):
                terms.append(beta[t + 1, s + 2])
            beta[t, s] = _logsumexp(*terms) + logp[t, ext[s]]
    grad = np.exp(logp)  # softmax
    for t in range(T):
        for s in range(n):
            v = alpha[t, s] + beta[t, s]
            if np.isfinite(v):
                grad[t, ext[s]] -= np.exp(v - logp[t, ext[s]] - log_z)
    return -log_z, grad


def ctc_batch(logits, utt_len, labels, label_len):
    """flat utterance-major logits [sum(utt_len), O], concatenated labels -> (sum of losses, gradient, #labels)"""
    logits = np.asarray(logits, dtype=np.float64)
    grad = np.zeros_like(logits)
    total, r, q = 0.0, 0, 0
    for n_t, n_s in zip(utt_len, label_len):
        loss, g = ctc_loss_and_grad(logits[r:r + n_t], labels[q:q + n_s])
        total += loss
        grad[r:r + n_t] = g
        r += n_t
        q += n_s
    return total, grad, int(np.sum(label_len))
