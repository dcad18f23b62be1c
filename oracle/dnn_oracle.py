"""CPU ORACLE (test infrastructure, NOT product code) for the DNN training hot path of vrenkens/tfkaldi.

A float64 numpy restatement of what the reference's TensorFlow graph computes for
neuralNetworks/trainer.py (Trainer / CrossEnthropyTrainer), neuralNetworks/decoder.py (Decoder) and
neuralNetworks/classifiers/{dnn,layer,activation,seq_convertors}.py.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product (tfkaldi_amd/) never does.

PARITY PINNING.  The arithmetic of this path lives in TensorFlow (import at trainer.py:5, nnet.py:8,
decoder.py:3, classifiers/*.py), a third-party dependency that is absent from /root/reference, not
pinned by it (no requirements file; README.md:12 points at TF 0.6, the API used bounds it to ~0.10-0.12)
and not installable here; the reference holds no tests, golden vectors or fixtures for the path
(SURVEY.md section 4, 8c).  The TensorFlow half of this oracle is therefore **parity unpinned**: it
restates the published TF-0.1x semantics of the ops at the reference's call sites, flagged ASSUMPTION
below where the behaviour is not visible in the reference tree, and is cross-checked against PyTorch
autograd (tests/test_oracle.py) and the known-answer properties of SURVEY.md section 8c.  The numpy
half of the reference (processing/*.py) IS importable; its golden vectors are in tests/golden/ (made by
oracle/make_golden_io.py).

ASSUMPTIONS (TF-0.1x behaviour outside the reference tree):
  A1 tf.contrib.layers.batch_norm defaults: decay 0.999, epsilon 1e-3, center (beta) only, no gamma,
     batch variance biased (tf.nn.moments), moving_mean init 0 / moving_variance init 1, the moving
     variance tracks the biased batch variance, one EMA update per session.run of the training graph.
  A2 tf.train.AdamOptimizer defaults beta1 .9, beta2 .999, epsilon 1e-8;
     lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); w -= lr_t * m / (sqrt(v) + epsilon); t starts at 1.
  A3 tf.train.exponential_decay non-staircase: lr = init * decay ** (global_step / num_steps).
  A4 tf.nn.dropout(x, keep) = x / keep * Bernoulli(keep) mask.
  A5 tf.nn.softmax_cross_entropy_with_logits = logsumexp(z) - z[y] for one-hot labels.
"""
import numpy as np

NONLINS = ("relu", "sigmoid", "tanh", "linear")


def _nonlin(u, kind):
    """activation.py:84 applied with the function chosen at nnet.py:48-62."""
    if kind == "relu":
        return np.maximum(u, 0.0)
    if kind == "sigmoid":
        return 1.0 / (1.0 + np.exp(-u))
    if kind == "tanh":
        return np.tanh(u)
    if kind == "linear":
        return u
    raise Exception('unkown nonlinearity')  # nnet.py:65


def _nonlin_grad(v, kind):
    """derivative of the nonlinearity expressed through its output v."""
    if kind == "relu":
        return (v > 0).astype(np.float64)
    if kind == "sigmoid":
        return v * (1.0 - v)
    if kind == "tanh":
        return 1.0 - v * v
    return np.ones_like(v)


def round_bf16(x):
    """float -> nearest bfloat16 (ties to even), returned as float64; as the engine rounds its fp32 values"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32).astype(np.float64)


class OracleDNN(object):
    """State + ops of Trainer (trainer.py:13-218) around DNN (dnn.py:10-131), in float64."""

    def __init__(self, input_dim, num_layers, num_units, output_dim, nonlin="relu", batch_norm=False,
                 l2_norm=False, keep_prob=1.0, layerwise_init=False, init_learning_rate=1e-3,
                 learning_rate_decay=1.0, num_steps=1, bn_decay=0.999, bn_epsilon=1e-3, beta1=0.9, beta2=0.999,
                 adam_epsilon=1e-8, gemm_dtype="float32"):
        assert nonlin in NONLINS
        # "bfloat16": restates the engine's mixed-precision mode (TFK_DTYPE_BF16, not a reference behaviour): the
        # operands of the three contractions are rounded to bfloat16, everything else keeps full precision
        assert gemm_dtype in ("float32", "bfloat16")
        self.gemm_dtype = gemm_dtype
        assert 0 < keep_prob  # activation.py:126 (values > 1 never reach Dropout: nnet.py:70-72)
        self.F, self.L, self.H, self.O = input_dim, num_layers, num_units, output_dim
        self.nonlin, self.bn, self.l2 = nonlin, bool(batch_norm), bool(l2_norm)
        self.keep = float(keep_prob)
        self.dropout = self.keep < 1  # nnet.py:70
        self.layerwise = bool(layerwise_init)
        self.init_lr, self.decay, self.num_steps = init_learning_rate, learning_rate_decay, num_steps
        self.bn_decay, self.bn_eps = bn_decay, bn_epsilon
        self.b1, self.b2, self.eps = beta1, beta2, adam_epsilon
        dims = [input_dim] + [num_units] * num_layers + [output_dim]
        # layer.py:42-48: weights [d_in, d_out], biases zeros.  Hidden weights are injected by the caller
        # (random init cannot match TF); dnn.py:67-68 + layer.py:39-44: output weights stddev 0 => zeros.
        self.W = [np.zeros((dims[l], dims[l + 1])) for l in range(num_layers + 1)]
        self.b = [np.zeros(dims[l + 1]) for l in range(num_layers + 1)]
        self.beta = [np.zeros(num_units) for _ in range(num_layers)]       # A1: center=True
        self.mov_mean = [np.zeros(num_units) for _ in range(num_layers)]   # A1
        self.mov_var = [np.ones(num_units) for _ in range(num_layers)]     # A1
        self.global_step = 0        # trainer.py:98-100
        self.lr_fact = 1.0          # trainer.py:104-106
        self.initialisedlayers = 0  # dnn.py:85-89
        self.adam_t = 0             # Adam's beta-power step count (A2)
        self._zero_accumulators()
        self.m = self._like_params()
        self.v = self._like_params()

    def _mm(self, a, b):
        """a . b in the arithmetic of the contractions"""
        if self.gemm_dtype == "bfloat16":
            return round_bf16(a).dot(round_bf16(b))
        return a.dot(b)

    # ---- parameter bookkeeping ----
    def params(self):
        """tf.trainable_variables() of the classifier (trainer.py:82): W, b of every layer, BN beta."""
        p = {}
        for l in range(self.L + 1):
            p["W%d" % l] = self.W[l]
            p["b%d" % l] = self.b[l]
        if self.bn:
            for l in range(self.L):
                p["beta%d" % l] = self.beta[l]
        return p

    def _like_params(self):
        return {k: np.zeros_like(v) for k, v in self.params().items()}

    def _zero_accumulators(self):
        self.G = self._like_params()   # trainer.py:118-122, init_grads :350
        self.batch_loss = 0.0          # trainer.py:91-93,  init_loss :351
        self.num_frames = 0            # trainer.py:126-128, init_num_frames :352

    def init_hidden_weights(self, rng):
        """layer.py:39-44: N(0, 1/sqrt(d_in)) for the hidden layers; returns the float32 values used."""
        out = []
        for l in range(self.L):
            d_in = self.W[l].shape[0]
            w = (rng.standard_normal(self.W[l].shape) / np.sqrt(d_in)).astype(np.float32)
            self.W[l] = w.astype(np.float64)
            out.append(w)
        return out

    def num_active(self):
        """dnn.py:97-102: tf.case selects activations[initialisedlayers], default activations[-1]."""
        if not self.layerwise:
            return self.L
        n = self.initialisedlayers + 1
        return self.L if (n > self.L or self.initialisedlayers < 0) else n

    def learning_rate(self):
        """trainer.py:110-112 (A3)."""
        return self.init_lr * self.decay ** (float(self.global_step) / self.num_steps) * self.lr_fact

    # ---- forward ----
    def _forward(self, X, train, masks=None, nfw=None, relu_active=None):
        """dnn.py:73-108 on flat frames X[T, F] (seq2nonseq, seq_convertors.py:12-39, is the caller's
        utterance-major concatenation).  Returns logits and the per-layer cache for backward.

        relu_active (test hook, ReLU nets only): per hidden layer a boolean [T, H] on/off pattern that REPLACES the
        oracle's own `u > 0` in the forward pass and in the derivative.  A unit whose pre-activation lies within
        fp32 round-off of the kink can come out on the other side in an fp32 implementation; pinning the pattern to
        the one the implementation chose lets every gradient be compared element-wise, while the caller asserts
        how rare (and how close to the kink) the disagreements are."""
        nact = self.num_active()
        if nfw is None:
            nfw = nact
        cache = []
        inp = np.asarray(X, dtype=np.float64)
        T = inp.shape[0]
        for l in range(nfw):
            c = {"in": inp}
            z = self._mm(inp, self.W[l]) + self.b[l]                 # layer.py:52
            u = z
            if self.bn:                                               # activation.py:159-161 (A1)
                if train:
                    mu = z.mean(axis=0)
                    var = ((z - mu) ** 2).mean(axis=0)                # biased
                    c["batch_mean"], c["batch_var"] = mu, var
                else:
                    mu, var = self.mov_mean[l], self.mov_var[l]
                rstd = 1.0 / np.sqrt(var + self.bn_eps)
                xhat = (z - mu) * rstd
                u = xhat + self.beta[l]
                c["xhat"], c["rstd"] = xhat, rstd
            v = _nonlin(u, self.nonlin)                               # activation.py:84
            if relu_active is not None:
                assert self.nonlin == "relu"
                c["own_active"], c["u"] = u > 0, u
                c["dact"] = np.asarray(relu_active[l], dtype=np.float64)
                v = u * c["dact"]
            c["v"] = v
            w = v
            if self.l2:                                               # activation.py:101-111
                s = (v ** 2).mean(axis=1, keepdims=True)              # the mean SQUARE, not its root
                w = np.where(s > 1, v / s, v)
                c["s"] = s
            a = w
            if self.dropout and train:                                # activation.py:140-141 (A4)
                mask = masks[l]
                a = w * mask / self.keep
                c["mask"] = mask
            c["a"] = a
            cache.append(c)
            inp = a
        logits = self._mm(cache[nact - 1]["a"], self.W[self.L]) + self.b[self.L]  # dnn.py:108, identity activation
        return logits, cache, nact

    @staticmethod
    def _xent(logits, y):
        """trainer.py:526-531 (A5): summed softmax cross-entropy and softmax probabilities."""
        mx = logits.max(axis=1, keepdims=True)
        ex = np.exp(logits - mx)
        se = ex.sum(axis=1, keepdims=True)
        lse = (mx + np.log(se))[:, 0]
        loss = float((lse - logits[np.arange(len(y)), y]).sum())
        return loss, ex / se

    # ---- update_gradients_op: one micro-batch (trainer.py:160-169) ----
    def accumulate(self, X, y, masks=None, relu_active=None):
        """Forward (train) + loss + backward; G += g, batch_loss += loss, num_frames += T, BN EMA.
        masks: per hidden layer 0/1 keep masks [T, H] when dropout is on.  relu_active: see _forward.  Returns the
        dict of this micro-batch's gradients (tf.gradients, trainer.py:155) for inspection."""
        y = np.asarray(y).astype(np.int64)
        T = len(y)
        logits, cache, nact = self._train_forward(X, masks, relu_active)
        loss, prob = self._xent(logits, y)
        dz = prob.copy()
        dz[np.arange(T), y] -= 1.0                                    # d(sum CE)/dlogits
        return self._backward_and_accumulate(dz, loss, T, logits, cache, nact)

    def _train_forward(self, X, masks=None, relu_active=None):
        # All BN layers' UPDATE_OPS are fetched (trainer.py:164-169), so with layer-wise growth the
        # hidden layers above the active depth are still evaluated in training mode.
        nfw = self.L if (self.layerwise and self.bn) else None
        return self._forward(X, True, masks, nfw, relu_active)

    # Loss-agnostic entry points (used with the CTC oracle, oracle/ctc_oracle.py): the training-mode logits of a
    # micro-batch, then the rest of update_gradients_op from a given d loss / d logits.
    def forward_logits(self, X, train=True, masks=None):
        if not train:
            return self._forward(X, False)[0]
        self._pending = self._train_forward(X, masks)
        return self._pending[0]

    def backward_from_dlogits(self, dlogits, loss, num_frames):
        logits, cache, nact = self._pending
        return self._backward_and_accumulate(np.asarray(dlogits, dtype=np.float64), loss, num_frames, logits, cache, nact)

    def _backward_and_accumulate(self, dz, loss, num_frames, logits, cache, nact):
        g = self._like_params()
        g["W%d" % self.L] = self._mm(cache[nact - 1]["a"].T, dz)
        g["b%d" % self.L] = dz.sum(axis=0)
        da = self._mm(dz, self.W[self.L].T)
        for l in range(nact - 1, -1, -1):
            c = cache[l]
            dw = da
            if self.dropout:
                dw = da * c["mask"] / self.keep
            dv = dw
            if self.l2:
                s, v = c["s"], c["v"]
                dot = (dw * v).sum(axis=1, keepdims=True)
                dv = np.where(s > 1, dw / s - v * (2.0 * dot / (self.H * s * s)), dw)
            du = dv * (c["dact"] if "dact" in c else _nonlin_grad(c["v"], self.nonlin))
            if self.bn:
                g["beta%d" % l] = du.sum(axis=0)
                xhat, rstd = c["xhat"], c["rstd"]
                dzl = rstd * (du - du.mean(axis=0) - xhat * (du * xhat).mean(axis=0))
            else:
                dzl = du
            g["W%d" % l] = self._mm(c["in"].T, dzl)
            g["b%d" % l] = dzl.sum(axis=0)
            if l > 0:
                da = self._mm(dzl, self.W[l].T)
        for k in g:                                                   # trainer.py:165-169
            self.G[k] = self.G[k] + g[k]
        self.batch_loss += loss
        self.num_frames += num_frames
        if self.bn:                                                   # UPDATE_OPS (A1)
            for l in range(len(cache)):
                d = self.bn_decay
                self.mov_mean[l] = d * self.mov_mean[l] + (1 - d) * cache[l]["batch_mean"]
                self.mov_var[l] = d * self.mov_var[l] + (1 - d) * cache[l]["batch_var"]
        self.last_logits, self.last_cache = logits, cache
        return g

    # ---- [average_loss, apply_gradients_op] + re-initialisation (trainer.py:336-352) ----
    def apply(self):
        """mean (trainer.py:174-175) -> clip (:178-179) -> Adam (:182-184, A2); returns the average loss
        evaluated in the same run (pre-update, train mode)."""
        avg_loss = self.batch_loss / float(self.num_frames)          # trainer.py:198
        lr = self.learning_rate()
        self.adam_t += 1
        t = self.adam_t
        lr_t = lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        p = self.params()
        for k in p:
            gk = np.clip(self.G[k] / float(self.num_frames), -1.0, 1.0)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * gk
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * gk * gk
            p[k] -= lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)
        self.global_step += 1
        self._zero_accumulators()
        return avg_loss

    # ---- update_valid_loss / average_loss (trainer.py:188-198, 433-441) ----
    def eval_accumulate(self, X, y):
        y = np.asarray(y).astype(np.int64)
        logits, _, _ = self._forward(X, False)
        loss, _ = self._xent(logits, y)
        self.batch_loss += loss
        self.num_frames += len(y)
        self.last_logits = logits

    def eval_finish(self):
        avg = self.batch_loss / float(self.num_frames)
        self.batch_loss, self.num_frames = 0.0, 0
        return avg

    # ---- Decoder (decoder.py:36-44) ----
    def posteriors(self, X):
        logits, _, _ = self._forward(X, False)
        mx = logits.max(axis=1, keepdims=True)
        ex = np.exp(logits - mx)
        return ex / ex.sum(axis=1, keepdims=True)

    # ---- control ops ----
    def halve_learning_rate(self):
        self.lr_fact /= 2.0                                           # trainer.py:141-142

    def add_layer(self):
        self.initialisedlayers += 1                                   # dnn.py:92

    def init_last_layer(self):
        self.W[self.L][...] = 0.0                                     # dnn.py:114-120 (stddev 0, zeros)
        self.b[self.L][...] = 0.0


def reference_microbatches(num_utt, per_minibatch):
    """Index lists of the micro-batches Trainer.update feeds (trainer.py:280-332), including its padding
    quirk: it appends `num_utt % U` zero-length dummies (not U - num_utt % U) and then runs
    floor(len / U) micro-batches.  Returns a list of lists of real utterance indices."""
    U = per_minibatch
    total = num_utt + (num_utt % U)
    out = []
    for k in range(total // U):
        idx = [i for i in range(k * U, (k + 1) * U) if i < num_utt]
        out.append(idx)
    return out
