"""Evaluation-mode hidden layers in one launch (EPI_EVAL_ACT: affine + moving-statistics batch norm + nonlinearity in
the GEMM epilogue; reference decoder.py:36-44, trainer.py:77-79 with is_training=False) against the three-launch
path it replaces (GEMM + bias, bn_stats_eval, act_forward; TFK_FUSE_EVAL=0) and against the float64 oracle.
The epilogue performs the same fp32 operations in the same order, so fp32 results are expected bit for bit; the
assertion allows 2 ulp in case the compiler contracts the two code sites differently.  Also the register-resident
softmax_rows kernel (posteriors and log(post / prior)) against the oracle."""
import numpy as np
import pytest

from util import assert_close, batch, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("nonlin,bn,dims", [("relu", True, (22, 3, 36, 13)), ("sigmoid", True, (40, 2, 64, 24)),
                                            ("tanh", False, (17, 2, 50, 9)), ("linear", True, (24, 2, 128, 300)),
                                            ("relu", True, (440, 2, 256, 100))])
def test_eval_fused_matches_three_launch_path(gpu, monkeypatch, nonlin, bn, dims, dtype):
    from tfkaldi_amd import _lib
    F, L, H, O = dims
    kw = dict(input_dim=F, num_layers=L, num_units=H, output_dim=O, nonlin=nonlin, batch_norm=bn,
              init_learning_rate=1e-3, num_steps=10, max_frames=512, compute_dtype=dtype)
    engines = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("TFK_FUSE_EVAL", flag)
        engines[flag], oracle = make_pair(np.random.default_rng(3), **kw)  # same parameters, non-trivial moving stats
    rng = np.random.default_rng(4)
    for T in (37, 300):
        X, y = batch(rng, T, F, O)
        outs = {}
        for flag, eng in engines.items():
            post = eng.posteriors(X)
            hidden = [eng.debug_fetch(_lib.DBG_HIDDEN, l, T) for l in range(L)]
            eng.eval_accumulate(X, y)
            outs[flag] = (post, hidden, eng.eval_finish())
        (p0, h0, l0), (p1, h1, l1) = outs["0"], outs["1"]
        for l in range(L):
            np.testing.assert_array_max_ulp(h1[l], h0[l], maxulp=2)
        if dtype == "float32":
            np.testing.assert_array_max_ulp(p1, p0, maxulp=4)
            assert abs(l1 - l0) <= 1e-6 * abs(l0)
            assert_close("posteriors vs oracle", p1, oracle.posteriors(X), 2e-4, 1e-7)
        else:  # the bf16 twin of a layer output is rounded from the same fp32 value on both paths
            np.testing.assert_allclose(p1, p0, rtol=1e-5, atol=1e-8)
    for eng in engines.values():
        eng.close()


@pytest.mark.parametrize("O", [9, 100, 1000, 2000, 4001, 9000])  # every register-resident width + the generic kernel
def test_softmax_rows_and_log_prior(gpu, O):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(O)
    kw = dict(input_dim=20, num_layers=1, num_units=32, output_dim=O, nonlin="tanh", batch_norm=False,
              init_learning_rate=1e-3, num_steps=10, max_frames=64)
    eng, oracle = make_pair(rng, **kw)
    oracle.W[1] *= 6.0  # spread the logits: a peaked softmax exercises the max / log-sum-exp path
    eng.set(_lib.WEIGHTS, 1, oracle.W[1])
    X, _ = batch(rng, 23, 20, O)
    want = oracle.posteriors(X)
    got = eng.posteriors(X)
    assert_close("posteriors", got, want, 2e-5, 1e-9)
    assert np.abs(got.sum(axis=1) - 1).max() < 1e-5
    prior = rng.random(O) + 0.05
    prior = (prior / prior.sum()).astype(np.float32)
    eng.set_prior(prior)
    ll = eng.posteriors(X, log_div_prior=True)
    logits = oracle._forward(X, False)[0]
    lse = np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)) + logits.max(1, keepdims=True)
    assert_close("log(post / prior)", ll, logits - lse - np.log(prior.astype(np.float64)), 2e-5, 2e-5)
    eng.close()
