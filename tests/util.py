"""Shared helpers for the parity tests (tests may import oracle/; the product never does)."""
import os

import numpy as np

from oracle.dnn_oracle import OracleDNN


def oracle_kwargs(kw):
    keys = ("input_dim", "num_layers", "num_units", "output_dim", "nonlin", "batch_norm", "l2_norm", "keep_prob",
            "layerwise_init", "init_learning_rate", "learning_rate_decay", "num_steps")
    return {k: kw[k] for k in keys if k in kw}


def randomize(oracle, rng, output_too=True, scale=1.0):
    """Reference init (hidden N(0, 1/sqrt(d_in)), rest 0) or, with output_too, a generic point where every
    parameter is non-zero so that every gradient path is exercised.  Values are fp32-representable."""
    f = lambda a: a.astype(np.float32).astype(np.float64)
    oracle.init_hidden_weights(rng)
    if output_too:
        L = oracle.L
        oracle.W[L] = f(rng.standard_normal(oracle.W[L].shape) * scale / np.sqrt(oracle.H))
        for l in range(L + 1):
            oracle.b[l] = f(rng.standard_normal(oracle.b[l].shape) * 0.1)
        if oracle.bn:
            for l in range(L):
                oracle.beta[l] = f(rng.standard_normal(oracle.H) * 0.1)
                oracle.mov_mean[l] = f(rng.standard_normal(oracle.H) * 0.1)
                oracle.mov_var[l] = f(1.0 + 0.2 * rng.random(oracle.H))


def copy_oracle_to_engine(oracle, eng):
    from tfkaldi_amd import _lib
    for l in range(oracle.L + 1):
        eng.set(_lib.WEIGHTS, l, oracle.W[l])
        eng.set(_lib.BIASES, l, oracle.b[l])
    if oracle.bn:
        for l in range(oracle.L):
            eng.set(_lib.BN_BETA, l, oracle.beta[l])
            eng.set(_lib.BN_MOVING_MEAN, l, oracle.mov_mean[l])
            eng.set(_lib.BN_MOVING_VAR, l, oracle.mov_var[l])


def make_pair(rng, output_too=True, **kw):
    """(Engine, OracleDNN) of the same shape holding the same parameters."""
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    # "bfloat16": engine in mixed precision, oracle rounding GEMM operands.  "float32" (the default) = fp32 emulated on the bf16
    # pipe; TFK_TEST_DTYPE=float32_mfma (tests/test_gpu_f32x3.py) re-runs the fp32 suites on the exact fp32 matrix instructions:
    # same oracle, same bounds -- both claim to be fp32
    dtype = kw.get("compute_dtype", os.environ.get("TFK_TEST_DTYPE", "float32"))
    oracle = OracleDNN(gemm_dtype="float32" if dtype.startswith("float32") else dtype, **oracle_kwargs(kw))
    randomize(oracle, rng, output_too)
    cfg = _lib.make_config(max_frames=kw.get("max_frames", 256), seed=kw.get("seed", 1234), compute_dtype=dtype,
                           device=kw.get("device", 0), **oracle_kwargs(kw))
    eng = Engine(cfg, torch_state=kw.get("torch_state", False))
    copy_oracle_to_engine(oracle, eng)
    return eng, oracle


def batch(rng, T, F, O):
    X = (rng.standard_normal((T, F)) * 1.5).astype(np.float32)
    y = rng.integers(0, O, size=T).astype(np.int32)
    return X, y


def engine_grads(eng):
    from tfkaldi_amd import _lib
    g = {}
    for l in range(eng.L + 1):
        g["W%d" % l] = eng.get(_lib.WEIGHTS, l, _lib.SLOT_GRAD)
        g["b%d" % l] = eng.get(_lib.BIASES, l, _lib.SLOT_GRAD)
    if eng.batch_norm:
        for l in range(eng.L):
            g["beta%d" % l] = eng.get(_lib.BN_BETA, l, _lib.SLOT_GRAD)
    return g


def engine_params(eng):
    from tfkaldi_amd import _lib
    p = {}
    for l in range(eng.L + 1):
        p["W%d" % l] = eng.get(_lib.WEIGHTS, l)
        p["b%d" % l] = eng.get(_lib.BIASES, l)
    if eng.batch_norm:
        for l in range(eng.L):
            p["beta%d" % l] = eng.get(_lib.BN_BETA, l)
    return p


def assert_close(name, got, want, rtol, atol):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: |%.9g - %.9g| = %.3g > %.3g at %s (worst of %d / %d violations)" % (
            name, got[i], want[i], err[i], tol[i], i, int((err > tol).sum()), err.size))
