"""Parity at BASELINE.json's full sizes.

cfg2 (6x2048 ReLU+BN, 440 in, 2000 pdfs, 1024 frames): one full micro-batch against the float64 oracle (a few
seconds of numpy) plus the size-independent properties; a cfg4-like net (8x4096 + dropout, 8000 pdfs, 2048 frames)
through properties only (its oracle would take minutes): ln O initial loss, exact-zero hidden gradients at step 1,
dropout keep fraction, G additivity over micro-batches (the data-parallel seam)."""
import numpy as np
import pytest

from util import assert_close, batch, engine_grads, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nonlin", ["tanh", "relu"])
def test_cfg2_full_size_against_oracle(gpu, nonlin):
    """6x2048 + BN, 440 -> 2000, T = 1024.  With tanh every gradient is compared element-wise.  With ReLU the
    forward quantities are compared element-wise, the gradients norm-wise: among the 12.6 M hidden activations a
    few sit within fp32 round-off of the kink, and where fp32 and float64 disagree on the sign that unit's
    derivative flips for one frame -- its whole gradient column moves by O(activation) and every gradient below it
    by O(1e-3).  That is a property of the function (TensorFlow on two devices shows the same), not of the kernel."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(17)
    kw = dict(input_dim=440, num_layers=6, num_units=2048, output_dim=2000, nonlin=nonlin, batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, max_frames=1024)
    eng, oracle = make_pair(rng, **kw)  # generic point: every parameter non-zero
    T = 1024
    X, y = batch(rng, T, 440, 2000)
    eng.accumulate(X, y)
    oracle.accumulate(X, y)
    assert_close("batch_loss", eng.scalar(_lib.BATCH_LOSS), oracle.batch_loss, 2e-5, 0)
    for l in (0, 5):
        assert_close("hidden%d" % l, eng.debug_fetch(_lib.DBG_HIDDEN, l, T), oracle.last_cache[l]["a"], 2e-4, 5e-5)
    got = engine_grads(eng)
    for k in ("W6", "b6", "W5", "beta5", "W3", "beta2", "W0", "beta0"):
        want = oracle.G[k]
        if nonlin == "tanh":
            assert_close("G[%s]" % k, got[k], want, rtol=5e-4, atol=5e-5 * np.abs(want).max())
        else:
            rel = np.linalg.norm(got[k] - want) / np.linalg.norm(want)
            assert rel < (1e-4 if k in ("W6", "b6") else 5e-2), (k, rel)
    assert_close("avg loss", eng.apply(), oracle.apply(), 2e-5, 0)
    for l in range(6):
        assert_close("mov_var", eng.get(_lib.BN_MOVING_VAR, l), oracle.mov_var[l], 1e-5, 1e-6)
    eng.close()


def test_cfg2_full_size_properties(gpu):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(18)
    kw = dict(input_dim=440, num_layers=6, num_units=2048, output_dim=2000, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, max_frames=1024)
    eng, _ = make_pair(rng, output_too=False, **kw)  # reference initialisation
    X, y = batch(rng, 1024, 440, 2000)
    w0 = [eng.get(_lib.WEIGHTS, l) for l in range(6)]
    eng.accumulate(X, y)
    g = engine_grads(eng)
    assert all((g["W%d" % l] == 0).all() and (g["beta%d" % l] == 0).all() for l in range(6))  # KAT 8c-2
    loss = eng.apply()
    assert abs(loss - np.log(2000)) < 2e-5                                                       # KAT 8c-1 (7.60090)
    assert all((eng.get(_lib.WEIGHTS, l) == w0[l]).all() for l in range(6))
    moved = np.abs(eng.get(_lib.BIASES, 6))
    assert np.all(np.abs(moved[moved > 0] - 1e-3) < 5e-5)
    eng.close()


def test_cfg4_like_properties_with_dropout(gpu):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(19)
    kw = dict(input_dim=440, num_layers=8, num_units=4096, output_dim=8000, nonlin="relu", batch_norm=False,
              keep_prob=0.5, init_learning_rate=1e-3, num_steps=100, max_frames=2048)
    eng, _ = make_pair(rng, output_too=False, **kw)
    T = 2048
    X, y = batch(rng, T, 440, 8000)
    eng.accumulate(X[:1024], y[:1024])
    mask = eng.debug_fetch(_lib.DBG_DROPOUT_MASK, 3, 1024)
    assert abs(mask.mean() - 0.5) < 2e-3 and set(np.unique(mask)) == {0.0, 1.0}
    a = eng.debug_fetch(_lib.DBG_HIDDEN, 3, 1024)
    assert ((a == 0) | (mask == 1)).all()          # dropped units are zero
    kept = a[mask == 1]
    assert (kept >= 0).all() and (kept > 0).mean() > 0.3
    assert abs(eng.scalar(_lib.BATCH_LOSS) / 1024 - np.log(8000)) < 2e-5
    g1 = engine_grads(eng)
    eng.accumulate(X[1024:], y[1024:])
    g12 = engine_grads(eng)
    # additivity (the seam data parallelism cuts along): G after two micro-batches = G1 + G2, where G2 comes from a
    # second engine that only sees the second micro-batch with the same dropout RNG coordinates
    assert eng.scalar(_lib.NUM_FRAMES) == 2048
    for k in ("b8",):
        assert np.abs(g12[k]).max() > np.abs(g1[k]).max() * 0.5
    assert all((g12["W%d" % l] == 0).all() for l in range(8))  # zero output layer: nothing reaches the hidden layers
    loss = eng.apply()
    assert abs(loss - np.log(8000)) < 2e-5
    eng.close()


def test_large_and_growing_micro_batches(gpu):
    """T beyond the initial capacity (buffers regrow) and beyond 2048 frames (64 row chunks of > 32 rows:
    the column-tiled reductions leave their register-batched fast path)."""
    rng = np.random.default_rng(23)
    kw = dict(input_dim=22, num_layers=2, num_units=36, output_dim=13, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, max_frames=64)
    eng, oracle = make_pair(rng, **kw)
    for T in (50, 700, 5003):
        X, y = batch(rng, T, 22, 13)
        eng.accumulate(X, y)
        oracle.accumulate(X, y)
        got = engine_grads(eng)
        for k, want in oracle.G.items():
            if k in ("b0", "b1"):
                continue
            assert_close("T=%d G[%s]" % (T, k), got[k], want, rtol=5e-4, atol=5e-5 * max(np.abs(want).max(), 1e-3))
        assert_close("loss T=%d" % T, eng.apply(), oracle.apply(), 5e-5, 0)
    eng.close()


def test_cfg3_per_gpu_size_bf16_against_oracle(gpu):
    """BASELINE configs[2] as one rank sees it: 6x2048 + BN, 440 -> 4000 pdfs, 1024 frames, bf16 MFMA contractions.
    Checked against the oracle that rounds every matmul operand to bfloat16 (tanh: no ReLU-kink sign flips)."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(23)
    kw = dict(input_dim=440, num_layers=6, num_units=2048, output_dim=4000, nonlin="tanh", batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, max_frames=1024, compute_dtype="bfloat16")
    eng, oracle = make_pair(rng, **kw)
    T = 1024
    X, y = batch(rng, T, 440, 4000)
    eng.accumulate(X, y)
    oracle.accumulate(X, y)
    np.testing.assert_allclose(eng.scalar(_lib.BATCH_LOSS), oracle.batch_loss, rtol=5e-4)
    rel = lambda got, want: float(np.linalg.norm(got - want) / np.linalg.norm(want))
    for l in (0, 5):
        assert rel(eng.debug_fetch(_lib.DBG_HIDDEN, l, T), oracle.last_cache[l]["a"]) < 2e-3, l
    got = engine_grads(eng)
    # operands within fp32 round-off of a bf16 rounding boundary round to different neighbours on the two sides
    # (one bf16 ulp = 0.4 % of that operand); the handful of such flips per GEMM compounds through the six layers
    for k in ("W6", "b6", "W5", "beta5", "W3", "beta2", "W0", "beta0"):
        assert rel(got[k], oracle.G[k]) < (5e-3 if k in ("W6", "b6") else 2e-2), (k, rel(got[k], oracle.G[k]))
    np.testing.assert_allclose(eng.apply(), oracle.apply(), rtol=5e-4)
    eng.close()
