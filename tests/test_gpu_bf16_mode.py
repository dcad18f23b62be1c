"""Mixed-precision mode (TFK_DTYPE_BF16, BASELINE cfg3 / cfg4): the three contractions take bf16 operands
(v_mfma_f32_32x32x16_bf16, fp32 accumulate), everything else is the fp32 path.

The oracle restates exactly that (`gemm_dtype="bfloat16"`: operands of every matmul rounded to bfloat16, ties to
even), so the comparison is as tight as fp32-vs-float64 allows -- except where a value sits within fp32 round-off of
a bf16 rounding boundary and the two sides round it to different neighbours (one bf16 ulp = 2^-8 of that operand).
Tensors are therefore compared by relative Frobenius error:
    hidden outputs / dlogits / gradients    <= 2e-3        (a handful of flipped operands among thousands)
    loss                                    rtol 5e-4
and against the pure-fp32 oracle the bf16 loss trace must stay within 2 % over 8 optimiser steps.
"""
import numpy as np
import pytest

from util import batch, copy_oracle_to_engine, engine_grads, engine_params, make_pair

pytestmark = pytest.mark.gpu

SMALL = dict(input_dim=24, num_layers=2, num_units=40, output_dim=16, init_learning_rate=1e-3, num_steps=100)
CHAINS = [
    dict(nonlin="relu", batch_norm=True),
    dict(nonlin="tanh", batch_norm=True),
    dict(nonlin="relu"),
    dict(nonlin="sigmoid", l2_norm=True),
    dict(nonlin="relu", batch_norm=True, keep_prob=0.7),
    dict(nonlin="relu", batch_norm=True, input_dim=22, num_units=37, output_dim=13),  # no dimension % 8 == 0
]


def _rel(got, want):
    return float(np.linalg.norm(np.asarray(got, dtype=np.float64) - want) / max(np.linalg.norm(want), 1e-30))


def _masks(eng, T):
    from tfkaldi_amd import _lib
    if eng.cfg.keep_prob >= 1:
        return None
    return [eng.debug_fetch(_lib.DBG_DROPOUT_MASK, l, T).astype(np.float64) for l in range(eng.L)]


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "-".join("%s=%s" % kv for kv in sorted(c.items())))
def test_bf16_accumulate_matches_bf16_oracle(gpu, chain):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(5)
    kw = dict(SMALL, compute_dtype="bfloat16", **chain)
    eng, oracle = make_pair(rng, **kw)
    T = 200
    X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
    eng.accumulate(X, y)
    oracle.accumulate(X, y, _masks(eng, T))
    for l in range(eng.L):
        assert _rel(eng.debug_fetch(_lib.DBG_HIDDEN, l, T), oracle.last_cache[l]["a"]) <= 2e-3, l
    np.testing.assert_allclose(eng.scalar(_lib.BATCH_LOSS), oracle.batch_loss, rtol=5e-4)
    got = engine_grads(eng)
    for k, want in oracle.G.items():
        if oracle.bn and k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
            continue  # bias under batch norm: true gradient 0, both sides hold round-off
        assert _rel(got[k], want) <= 2e-3, (k, _rel(got[k], want))
    # second micro-batch accumulates, then the optimiser step
    X2, y2 = batch(rng, 77, kw["input_dim"], kw["output_dim"])
    eng.accumulate(X2, y2)
    oracle.accumulate(X2, y2, _masks(eng, 77))
    np.testing.assert_allclose(eng.apply(), oracle.apply(), rtol=5e-4)
    eng.close()


@pytest.mark.parametrize("dims", [dict(input_dim=440, num_units=256, output_dim=100, num_layers=2),
                                  dict(input_dim=22, num_units=37, output_dim=13, num_layers=3)],
                         ids=["baseline-cfg1", "odd-dims"])
def test_bf16_training_trace(gpu, dims):
    """several optimiser steps from the reference initialisation: the loss trace follows the bf16 oracle closely
    and the full-precision oracle within 2 %; the bf16 weight shadow maintained by the optimiser (or rebuilt after
    it when the layout does not allow that) equals a shadow rebuilt from the fp32 master weights"""
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    rng = np.random.default_rng(11)
    kw = dict(SMALL, nonlin="relu", batch_norm=True, **dims)
    eng, oracle = make_pair(rng, output_too=False, compute_dtype="bfloat16", **kw)
    from oracle.dnn_oracle import OracleDNN
    from util import oracle_kwargs
    full = OracleDNN(**oracle_kwargs(kw))
    for l in range(oracle.L + 1):
        full.W[l] = oracle.W[l].copy()
    F, O = kw["input_dim"], kw["output_dim"]
    data = [batch(rng, 256, F, O) for _ in range(2)]
    for step in range(8):
        for X, y in data[: 1 + step % 2]:
            eng.accumulate(X, y)
            oracle.accumulate(X, y)
            full.accumulate(X, y)
        got, want, ref = eng.apply(), oracle.apply(), full.apply()
        if step == 0:
            assert abs(got - np.log(O)) < 1e-4  # KAT: zero output layer
        np.testing.assert_allclose(got, want, rtol=2e-3, err_msg="step %d vs bf16 oracle" % step)
        np.testing.assert_allclose(got, ref, rtol=2e-2, err_msg="step %d vs full-precision oracle" % step)
    # shadow check: a fresh engine fed the trained fp32 parameters must produce the same logits bit for bit
    params = engine_params(eng)
    cfg = _lib.make_config(max_frames=256, seed=1234, compute_dtype="bfloat16", **oracle_kwargs(kw))
    other = Engine(cfg)
    for l in range(eng.L + 1):
        other.set(_lib.WEIGHTS, l, params["W%d" % l])
        other.set(_lib.BIASES, l, params["b%d" % l])
    for l in range(eng.L):
        other.set(_lib.BN_BETA, l, params["beta%d" % l])
        other.set(_lib.BN_MOVING_MEAN, l, eng.get(_lib.BN_MOVING_MEAN, l))
        other.set(_lib.BN_MOVING_VAR, l, eng.get(_lib.BN_MOVING_VAR, l))
    X = data[0][0]
    assert (eng.posteriors(X, raw_logits=True) == other.posteriors(X, raw_logits=True)).all()
    eng.close(); other.close()


def test_bf16_cfg2_size_step(gpu):
    """BASELINE cfg2 shape in mixed precision: initial loss = ln O exactly (zero output layer), the loss falls, and
    the step is deterministic (two engines, same inputs, identical bits)"""
    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    T, F, L, H, O = 1024, 440, 6, 2048, 2000
    losses = []
    for rep in range(2):
        cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, max_frames=T, num_steps=100,
                               compute_dtype="bfloat16")
        eng = Engine(cfg)
        eng.init_hidden_weights(np.random.default_rng(7))
        g = torch.Generator(device="cuda").manual_seed(3)
        X = torch.randn(T, F, device="cuda", generator=g)
        y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32, generator=g)
        trace = []
        for _ in range(6):
            eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
            trace.append(eng.apply())
        losses.append(trace)
        eng.close()
    assert abs(losses[0][0] - np.log(O)) < 1e-3
    assert losses[0][-1] < 0.5 * losses[0][0]
    assert losses[0] == losses[1]


def test_bf16_trainer_end_to_end(gpu, tmp_path):
    """DNN(compute_dtype='bfloat16') through Trainer / Decoder on synthetic ark files"""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    from tfkaldi_amd.neuralNetworks.decoder import Decoder
    from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    D, C, O = 8, 2, 11
    F = D * (2 * C + 1)
    paths = synthetic.write_corpus(str(tmp_path / "data"), 12, O, feat_dim=D, utt_len=30, num_speakers=2)
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], C, 30,
                                          cmvn_on_device=True)
    disp = batchdispenser.AlignmentBatchDispenser(reader, target_coder.AlignmentCoder(lambda x, y: x, O), 4,
                                                  paths["alignments"])
    dnn = DNN(O, 2, 32, act.TfActivation(act.Batchnorm(None), "relu"), False, compute_dtype="bfloat16")
    tr = CrossEnthropyTrainer(dnn, F, 30, disp.max_target_length, 1e-2, 1.0, 20, 2, seed=7)
    tr.initialize()
    batch_ = disp.get_batch()
    losses = [tr.update(*batch_) for _ in range(12)]
    assert abs(losses[0] - np.log(O)) < 1e-3 and losses[-1] < 0.7 * losses[0]
    val = tr.evaluate(*batch_)
    assert np.isfinite(val)
    tr.save_model(str(tmp_path / "model"))
    tr.close()
    dec = Decoder(dnn, F, 30)
    dec.restore(str(tmp_path / "model"))
    post = dec(batch_[0][0])
    assert post.shape == (30, O) and np.allclose(post.sum(1), 1, atol=1e-5)
    dec.close()


def test_bf16_input_paths_and_layerwise_growth(gpu):
    """mixed precision on the other entry points: device-resident X, unspliced frames (+ CMVN table), layer-wise
    growth (the weight shadow is rebuilt after control_ops['init'] zeroes the output layer)"""
    import torch
    from tfkaldi_amd import _lib
    D, C = 8, 2
    F = D * (2 * C + 1)
    kw = dict(input_dim=F, num_layers=3, num_units=40, output_dim=16, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-3, num_steps=100, compute_dtype="bfloat16")
    rng = np.random.default_rng(8)
    lens = [20, 9, 31]
    utts = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    spliced = []
    for u in utts:
        pad = np.zeros((u.shape[0] + 2 * C, D), dtype=np.float32)
        pad[C:C + u.shape[0]] = u
        spliced.append(np.concatenate([pad[j:j + u.shape[0]] for j in range(2 * C + 1)], axis=1))
    X = np.concatenate(spliced)
    T = X.shape[0]
    y = rng.integers(0, 16, size=T).astype(np.int32)
    engines = [make_pair(np.random.default_rng(2), **kw)[0] for _ in range(3)]
    a, b, c = engines
    dX = torch.from_numpy(X).cuda(); dy = torch.from_numpy(y).cuda()
    for _ in range(2):
        a.accumulate(X, y, last=True)
        b.accumulate_device(dX.data_ptr(), F, dy.data_ptr(), T, last=True)
        c.accumulate_raw(np.concatenate(utts), y, lens, C, last=True)
        la, lb, lc = a.apply(), b.apply(), c.apply()
        assert la == lb == lc
    assert (a.posteriors(X) == c.posteriors_raw(np.concatenate(utts), lens, C)).all()
    for e in engines:
        e.close()
    # layer-wise growth against the bf16 oracle
    eng, oracle = make_pair(np.random.default_rng(3), output_too=False, layerwise_init=True, **kw)
    for step in range(4):
        eng.accumulate(X, y, last=True)
        oracle.accumulate(X, y)
        np.testing.assert_allclose(eng.apply(), oracle.apply(), rtol=2e-3)
        if step in (0, 2):
            eng.add_layer(); oracle.add_layer()
            eng.init_last_layer(); oracle.init_last_layer()
            eng.eval_accumulate(X, y)
            assert abs(eng.eval_finish() - np.log(16)) < 1e-5  # zero output layer again: ln O exactly
    eng.close()
