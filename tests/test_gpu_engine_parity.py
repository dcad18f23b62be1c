"""HIP engine (through the C ABI) against the float64 oracle on identical inputs.

Tolerances (fp32 arithmetic on the GPU vs float64 oracle; BASELINE.json asks for "stated fp32 tolerance"):
  loss per step                rtol 2e-5
  logits / posteriors          rtol 1e-4, atol 2e-5
  gradient sums G              rtol 2e-4 + atol 2e-5 * max|G| of that tensor (fp32 sums over T frames)
  parameters after Adam        see test_multi_step_training (Adam divides by sqrt(v): it amplifies round-off of
                               near-zero gradients, in TF as much as here)
"""
import os

import numpy as np
import pytest

from util import assert_close, batch, engine_grads, engine_params, make_pair

pytestmark = pytest.mark.gpu

SMALL = dict(input_dim=22, num_layers=2, num_units=36, output_dim=13, init_learning_rate=1e-3, num_steps=100)
CHAINS = [
    dict(nonlin="relu", batch_norm=True),                                  # the AURORA4 recipe
    dict(nonlin="relu"),
    dict(nonlin="sigmoid", batch_norm=True),
    dict(nonlin="tanh", l2_norm=True),
    dict(nonlin="relu", batch_norm=True, l2_norm=True),
    dict(nonlin="linear"),
    dict(nonlin="relu", batch_norm=True, keep_prob=0.7),
    dict(nonlin="sigmoid", l2_norm=True, keep_prob=0.6),
    dict(nonlin="relu", l2_norm=True, keep_prob=0.9, big=True),           # rows with mean square > 1
]


def _masks(eng, T):
    from tfkaldi_amd import _lib
    if eng.cfg.keep_prob >= 1:
        return None
    return [eng.debug_fetch(_lib.DBG_DROPOUT_MASK, l, T).astype(np.float64) for l in range(eng.L)]


def _check_grads(eng, oracle):
    got = engine_grads(eng)
    for k, want in oracle.G.items():
        if oracle.bn and k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
            # bias under batch norm: the true gradient is 0, both sides hold round-off only
            assert np.abs(got[k]).max() <= 1e-4 * max(1.0, np.abs(oracle.G["W" + k[1:]]).max()), k
            continue
        assert_close("G[%s]" % k, got[k], want, rtol=2e-4, atol=2e-5 * max(np.abs(want).max(), 1e-3))


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "-".join("%s=%s" % kv for kv in sorted(c.items())))
def test_accumulate_matches_oracle(gpu, chain):
    """one micro-batch: logits, loss, every gradient, BN moving averages"""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(5)
    kw = dict(SMALL, **{k: v for k, v in chain.items() if k != "big"})
    eng, oracle = make_pair(rng, **kw)
    if chain.get("big"):
        from util import copy_oracle_to_engine
        for l in range(oracle.L):
            oracle.W[l] = oracle.W[l] * 3
        copy_oracle_to_engine(oracle, eng)
    T = 75
    X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
    if chain.get("big"):
        X[: T // 2] *= 0.02  # quiet frames stay below the L2Norm threshold, loud ones exceed it
    eng.accumulate(X, y)
    oracle.accumulate(X, y, _masks(eng, T))
    # the buffer holds dLogits = softmax - onehot after the backward pass
    dlog = eng.debug_fetch(_lib.DBG_LOGITS, 0, T)
    prob = np.exp(oracle.last_logits - oracle.last_logits.max(1, keepdims=True))
    prob /= prob.sum(1, keepdims=True)
    prob[np.arange(T), y] -= 1
    assert_close("dlogits", dlog, prob, rtol=1e-4, atol=2e-6)
    for l in range(eng.L):
        assert_close("hidden%d" % l, eng.debug_fetch(_lib.DBG_HIDDEN, l, T), oracle.last_cache[l]["a"], 1e-4, 2e-5)
    assert_close("batch_loss", eng.scalar(_lib.BATCH_LOSS), oracle.batch_loss, 2e-5, 0)
    assert eng.scalar(_lib.NUM_FRAMES) == T
    _check_grads(eng, oracle)
    if chain.get("big"):  # both L2 branches taken
        s = oracle.last_cache[0]["s"]
        assert (s > 1).any() and (s <= 1).any()
    # second micro-batch accumulates on top (G += g)
    X2, y2 = batch(rng, 41, kw["input_dim"], kw["output_dim"])
    eng.accumulate(X2, y2)
    oracle.accumulate(X2, y2, _masks(eng, 41))
    _check_grads(eng, oracle)
    loss = eng.apply()
    assert_close("avg loss", loss, oracle.apply(), 2e-5, 0)
    if oracle.bn:
        for l in range(eng.L):
            assert_close("mov_mean", eng.get(_lib.BN_MOVING_MEAN, l), oracle.mov_mean[l], 1e-5, 1e-6)
            assert_close("mov_var", eng.get(_lib.BN_MOVING_VAR, l), oracle.mov_var[l], 1e-5, 1e-6)
    eng.close()


@pytest.mark.parametrize("dims", [dict(input_dim=7, num_units=10, output_dim=5, num_layers=3),
                                  dict(input_dim=440, num_units=256, output_dim=100, num_layers=2)],
                         ids=["odd-dims", "baseline-cfg1"])
def test_multi_step_training(gpu, dims):
    """per-step loss trace + parameters over several optimiser steps from the reference initialisation
    (output layer zero).  dims[0] has no dimension divisible by 4 (padded leading dimensions);
    dims[1] is BASELINE configs[0] (2x256, 440 in, 100 pdfs, batch 256)."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(11)
    kw = dict(SMALL, nonlin="relu", batch_norm=True, **dims)
    eng, oracle = make_pair(rng, output_too=False, **kw)
    F, O = kw["input_dim"], kw["output_dim"]
    lr = kw["init_learning_rate"]
    data = [batch(rng, 256 if F == 440 else 37, F, O) for _ in range(3)]
    w0 = engine_params(eng)
    for step in range(6):
        for X, y in data[: 1 + step % 2]:
            eng.accumulate(X, y)
            oracle.accumulate(X, y)
        got, want = eng.apply(), oracle.apply()
        if step == 0:
            # KAT: zero output layer => initial loss is ln O per frame
            assert abs(got - np.log(O)) < 1e-5
            # KAT (8c-2): dA = dZ . W_out^T = 0 => hidden W/beta get g = 0 and Adam leaves them bit-identical
            p1 = engine_params(eng)
            for l in range(eng.L):
                assert (p1["W%d" % l] == w0["W%d" % l]).all()
                assert (p1["beta%d" % l] == w0["beta%d" % l]).all()
            moved = np.abs(p1["b%d" % eng.L] - w0["b%d" % eng.L])
            assert np.all(np.abs(moved - lr) < 0.05 * lr)  # first Adam step ~ lr * sign(g)
        assert_close("loss step %d" % step, got, want, 5e-5 * (step + 1), 0)
    assert eng.global_step == 6
    got = engine_params(eng)
    for k, want in oracle.params().items():
        if k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
            continue  # dead parameter under batch norm (gradient is round-off noise amplified by Adam)
        err = np.abs(got[k] - want)
        # Adam normalises each gradient by its own magnitude: elements whose gradient is near the fp32
        # round-off floor can differ by a fraction of lr per step; all others track the oracle closely.
        assert np.mean(err > 0.02 * lr * 6) < 0.01, (k, np.mean(err > 0.02 * lr * 6))
        assert err.max() <= 2.0 * lr * 6, k
    eng.close()


def test_adam_known_answer(gpu):
    """mean -> clip -> Adam on injected gradient sums spanning the clip range (KAT 8c-4)."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(3)
    kw = dict(SMALL, nonlin="relu")
    eng, oracle = make_pair(rng, **kw)
    T = 20
    X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
    eng.accumulate(X, y)
    oracle.accumulate(X, y)
    for step in range(3):
        for l in range(eng.L + 1):
            gw = (rng.uniform(-3, 3, oracle.W[l].shape) * T).astype(np.float32)
            gb = (rng.uniform(-3, 3, oracle.b[l].shape) * T).astype(np.float32)
            eng.set(_lib.WEIGHTS, l, gw, _lib.SLOT_GRAD)
            eng.set(_lib.BIASES, l, gb, _lib.SLOT_GRAD)
            oracle.G["W%d" % l], oracle.G["b%d" % l] = gw.astype(np.float64), gb.astype(np.float64)
        eng.apply()
        oracle.apply()
        got = engine_params(eng)
        for k, want in oracle.params().items():
            assert_close("%s step %d" % (k, step), got[k], want, 1e-5, 2e-6)
        # gradient sums are zeroed by apply (trainer.py:350)
        assert all((g == 0).all() for g in engine_grads(eng).values())
        if step < 2:
            eng.accumulate(X, y)
            oracle.accumulate(X, y)
    eng.close()


def test_eval_and_posteriors(gpu):
    """Trainer.evaluate (eval-mode BN / no dropout) and Decoder posteriors, incl. log(post / prior)."""
    rng = np.random.default_rng(9)
    kw = dict(SMALL, nonlin="relu", batch_norm=True, keep_prob=0.8)
    eng, oracle = make_pair(rng, **kw)
    for T in (64, 9):
        X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
        eng.eval_accumulate(X, y)
        oracle.eval_accumulate(X, y)
    assert_close("valid loss", eng.eval_finish(), oracle.eval_finish(), 2e-5, 0)
    X, _ = batch(rng, 50, kw["input_dim"], kw["output_dim"])
    post = eng.posteriors(X)
    want = oracle.posteriors(X)
    assert_close("posteriors", post, want, 1e-4, 1e-7)
    assert np.allclose(post.sum(1), 1, atol=1e-5)
    prior = rng.random(kw["output_dim"]) + 0.1
    prior = (prior / prior.sum()).astype(np.float32)
    eng.set_prior(prior)
    assert_close("log(post/prior)", eng.posteriors(X, log_div_prior=True), np.log(want / prior), 1e-4, 2e-5)
    # KAT 8c-9: eval-mode BN at initialisation is z / sqrt(1 + 1e-3)
    eng.close()


def test_layerwise_growth(gpu):
    """dnn.py:81-122: logits taken after `initialisedlayers + 1` hidden layers; 'add' / 'init' control ops."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(21)
    kw = dict(SMALL, num_layers=3, nonlin="relu", batch_norm=True, layerwise_init=True)
    eng, oracle = make_pair(rng, **kw)
    X, y = batch(rng, 48, kw["input_dim"], kw["output_dim"])
    for phase in range(4):
        eng.accumulate(X, y)
        oracle.accumulate(X, y)
        _check_grads(eng, oracle)
        assert_close("loss", eng.apply(), oracle.apply(), 5e-5, 0)
        for l in range(eng.L):  # every BN layer's moving stats move, active or not
            assert_close("mov_mean", eng.get(_lib.BN_MOVING_MEAN, l), oracle.mov_mean[l], 1e-5, 1e-6)
        eng.add_layer(); oracle.add_layer()
        eng.init_last_layer(); oracle.init_last_layer()
        assert (eng.get(_lib.WEIGHTS, eng.L) == 0).all()
    eng.close()


def test_data_parallel_equivalence(gpu):
    """KAT 8c-3: k serial micro-batches on one engine == one micro-batch on each of k engines followed by a
    SUM all-reduce of the reduce region (emulated here on one GPU by adding the regions), including the
    sequential BN moving-average composition."""
    import torch
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(31)
    kw = dict(SMALL, nonlin="relu", batch_norm=True)
    k = 4
    serial, oracle = make_pair(np.random.default_rng(1), **kw)
    ranks = [make_pair(np.random.default_rng(1), torch_state=True, **kw)[0] for _ in range(k)]
    mbs = [batch(rng, 30 + 7 * r, kw["input_dim"], kw["output_dim"]) for r in range(k)]
    for step in range(2):
        for X, y in mbs:
            serial.accumulate(X, y)
            oracle.accumulate(X, y)
        fired = []
        for r, eng in enumerate(ranks):
            eng.set_later_microbatches(k - 1 - r)
            eng.set_bucket_callback(lambda b, r=r: fired.append((r, b)))
            eng.accumulate(mbs[r][0], mbs[r][1], last=True)
        # announcement order: scalars + BN increments first (final after the loss), weights in reverse-layer order,
        # then the bias / beta gradients
        L = ranks[0].L
        assert [b for (r, b) in fired if r == 0] == [L + 2] + list(range(L + 1)) + [L + 1]
        for eng in ranks:
            eng.synchronize()
        total = sum(eng.reduce_view().clone() for eng in ranks)
        for eng in ranks:
            eng.reduce_view().copy_(total)
        torch.cuda.synchronize()
        losses = [eng.apply() for eng in ranks]
        want = serial.apply()
        assert_close("oracle loss", want, oracle.apply(), 2e-5, 0)
        for l in losses:
            assert_close("dp loss", l, want, 1e-6, 0)
        ps = engine_params(serial)
        lr = kw["init_learning_rate"]
        for eng in ranks:
            pr = engine_params(eng)
            for name in ps:
                if name.startswith("b") and not name.startswith("beta") and name != "b%d" % eng.L:
                    continue
                err = np.abs(pr[name] - ps[name])
                assert np.mean(err > 0.02 * lr) < 0.01 and err.max() <= 2 * lr * (step + 1), name
            for l in range(eng.L):
                assert_close("mov_mean", eng.get(_lib.BN_MOVING_MEAN, l), serial.get(_lib.BN_MOVING_MEAN, l), 1e-5, 1e-6)
                assert_close("mov_var", eng.get(_lib.BN_MOVING_VAR, l), serial.get(_lib.BN_MOVING_VAR, l), 1e-5, 1e-6)
    buckets = ranks[0].buckets()
    _, n = ranks[0].reduce_region()
    assert sum(c for _, c in buckets) == n  # the buckets tile the reduce region exactly
    for e in ranks + [serial]:
        e.close()


def test_narrow_net_many_frames_split_k(gpu):
    """3000 frames through a 48-unit net: the weight-gradient GEMMs have ONE output tile and a long contraction, so
    they run split-K (gemm_f32.h): partial results per chunk, summed in chunk order.  First micro-batch overwrites
    G, the second accumulates on top; two engines give identical bits."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(29)
    kw = dict(input_dim=24, num_layers=2, num_units=48, output_dim=20, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, max_frames=3000)
    eng, oracle = make_pair(rng, **kw)
    twin, _ = make_pair(np.random.default_rng(29), **kw)
    for T in (3000, 2500):
        X, y = batch(rng, T, 24, 20)
        eng.accumulate(X, y)
        twin.accumulate(X, y)
        oracle.accumulate(X, y)
        _check_grads(eng, oracle)
    a, b = engine_grads(eng), engine_grads(twin)
    assert all((a[k] == b[k]).all() for k in a)
    assert_close("avg loss", eng.apply(), oracle.apply(), 2e-5, 0)
    eng.close(); twin.close()


@pytest.mark.parametrize("keep_prob", [1.0, 0.8])
def test_tall_microbatch_merges_statistics_once(gpu, keep_prob):
    """More than 32 GEMM row tiles per micro-batch (here 2200 frames = 35 tiles): the per-tile batch-norm statistics
    and the EPI_DACT partial sums are merged by one small kernel instead of inside every block of the column-tiled
    kernels (kMergeOnceChunks, csrc/kernels.h).  Same numbers as ever: hidden outputs, loss, every gradient, moving
    averages -- with and without dropout (the row-wise kernel applies the same masks)."""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(41)
    kw = dict(input_dim=33, num_layers=3, num_units=200, output_dim=45, nonlin="tanh", batch_norm=True,
              keep_prob=keep_prob, init_learning_rate=1e-3, num_steps=100, max_frames=2200)
    eng, oracle = make_pair(rng, **kw)
    for T in (2200, 2113):
        X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
        eng.accumulate(X, y)
        oracle.accumulate(X, y, _masks(eng, T))
        for l in range(eng.L):
            assert_close("hidden%d" % l, eng.debug_fetch(_lib.DBG_HIDDEN, l, T), oracle.last_cache[l]["a"], 1e-4, 2e-5)
        _check_grads(eng, oracle)
    assert_close("avg loss", eng.apply(), oracle.apply(), 2e-5, 0)
    for l in range(eng.L):
        assert_close("mov_mean", eng.get(_lib.BN_MOVING_MEAN, l), oracle.mov_mean[l], 1e-5, 1e-6)
        assert_close("mov_var", eng.get(_lib.BN_MOVING_VAR, l), oracle.mov_var[l], 1e-5, 1e-6)
    eng.close()


def test_long_posteriors_pass_is_pipelined_and_equal(gpu, monkeypatch):
    """tfk_posteriors on a long pass (>= 2 chunks of TFK_POST_CHUNK rows, pinned output) computes chunk c + 1 while
    chunk c travels back over PCIe.  Rows are independent in evaluation mode: same numbers as the one-shot pass and as
    the oracle, for spliced and for device-spliced input."""
    rng = np.random.default_rng(53)
    kw = dict(input_dim=44, num_layers=2, num_units=96, output_dim=70, nonlin="relu", batch_norm=True, max_frames=5000)
    T = 4600
    monkeypatch.setenv("TFK_POST_CHUNK", "2048")
    piped, oracle = make_pair(rng, **kw)
    monkeypatch.setenv("TFK_POST_CHUNK", "0")
    whole, _ = make_pair(np.random.default_rng(53), **kw)
    X, _ = batch(rng, T, kw["input_dim"], kw["output_dim"])
    prior = rng.random(kw["output_dim"]).astype(np.float32) + 0.1
    prior /= prior.sum()
    for e in (piped, whole):
        e.set_prior(prior)
    want = oracle.posteriors(X)
    for log_div in (False, True):
        a, b = piped.posteriors(X, log_div_prior=log_div), whole.posteriors(X, log_div_prior=log_div)
        assert a.shape == b.shape == (T, kw["output_dim"])
        assert_close("piped vs whole", a, b, 1e-5, 1e-6)
        ref = np.log(want / prior) if log_div else want
        assert_close("piped vs oracle", a, ref, 2e-4, 2e-6)
    # unspliced frames, splice on the device (4 utterances)
    raw = rng.standard_normal((T, 4)).astype(np.float32)
    lens = [1000, 1500, 1100, 1000]
    a = piped.posteriors_raw(raw, lens, 5)
    b = whole.posteriors_raw(raw, lens, 5)
    assert_close("raw piped vs whole", a, b, 1e-5, 1e-6)
    piped.close(); whole.close()


def test_apply_without_frames_fails_loudly(gpu):
    """an optimiser step with num_frames = 0 is G / 0 (trainer.py:174-175): the reference would write NaN into every
    parameter; the engine leaves the parameters alone and tfk_apply reports the error (round-1 advisor finding)"""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(77)
    kw = dict(SMALL, nonlin="relu", batch_norm=True)
    eng, oracle = make_pair(rng, **kw)
    before = engine_params(eng)
    with pytest.raises(_lib.EngineError, match="without frames"):
        eng.apply()
    after = engine_params(eng)
    assert all((after[k] == before[k]).all() for k in before)
    assert eng.global_step == 0 and eng.scalar(_lib.ADAM_STEPS) == 0
    X, y = batch(rng, 40, kw["input_dim"], kw["output_dim"])  # the engine is still usable, and still in step with the oracle
    eng.accumulate(X, y)
    oracle.accumulate(X, y)
    assert_close("loss", eng.apply(), oracle.apply(), 2e-5, 0)
    assert eng.global_step == 1
    eng.close()


def test_parity_with_the_optimiser_on_its_own_stream():
    """TFK_ADAM_OVERLAP=1 (off by default: measured no faster, profiles/r03_fusion_experiments.txt) moves Adam to a second
    stream behind per-layer events; every reader of parameters, gradients or moments must join it.  The training, Adam
    known-answer, evaluation / tensor-get and layer-wise growth tests of this file, run again with the switch on."""
    import subprocess
    import sys
    env = dict(os.environ, TFK_ADAM_OVERLAP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "multi_step_training or adam_known_answer or eval_and_posteriors or layerwise_growth or "
                        "data_parallel_equivalence"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
