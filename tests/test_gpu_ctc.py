"""CTC loss on the device (tfkaldi_amd/csrc/ctc.hip through tfk_accumulate_ctc) against the float64 oracle
(oracle/ctc_oracle.py, itself pinned against torch's ctc_loss in tests/test_ctc_oracle.py).

The kernel works in fp32 log space; tolerances: loss rtol 2e-5, dLogits atol 2e-5 (values are O(1) probabilities),
parameter gradients as in the cross-entropy parity test (rtol 2e-4 + 2e-5 * max)."""
import numpy as np
import pytest

from oracle.ctc_oracle import ctc_batch
from util import assert_close, engine_grads, make_pair

pytestmark = pytest.mark.gpu

KW = dict(input_dim=20, num_layers=2, num_units=32, output_dim=9, nonlin="tanh", batch_norm=True,
          init_learning_rate=1e-3, num_steps=50)


def _batch(rng, utt_lens, label_lens, F, O, repeat_heavy=False):
    X = (rng.standard_normal((int(np.sum(utt_lens)), F)) * 1.5).astype(np.float32)
    hi = 2 if repeat_heavy else O - 1   # few distinct labels -> many repeats
    labels = np.concatenate([rng.integers(0, hi, size=n) for n in label_lens] + [np.zeros(0, dtype=np.int64)])
    return X, labels.astype(np.int32)


def _oracle_step(oracle, X, utt_lens, labels, label_lens):
    """oracle forward, CTC loss + dLogits from the CTC oracle, oracle backward from those dLogits"""
    logits = oracle.forward_logits(X)
    loss, dlog, n_labels = ctc_batch(logits, utt_lens, labels, label_lens)
    oracle.backward_from_dlogits(dlog, loss, n_labels)
    return loss, dlog


@pytest.mark.parametrize("case", [
    dict(utt=[30, 17, 44, 9], lab=[5, 3, 11, 0]),                       # incl. an empty label sequence
    dict(utt=[25, 40], lab=[12, 19], repeat=True),                      # repeated labels: blanks are mandatory
    dict(utt=[7, 60], lab=[7, 2], repeat=False),                        # T == S: every frame emits a label
    dict(utt=[300], lab=[140]),                                         # 281 states: 8 states per lane
    dict(utt=[210, 150], lab=[70, 66]),                                 # 141 states: 4 states per lane
], ids=["mixed", "repeats", "tight", "long", "medium"])
def test_ctc_accumulate_matches_oracle(gpu, case):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(41)
    eng, oracle = make_pair(rng, max_frames=512, **KW)
    X, labels = _batch(rng, case["utt"], case["lab"], KW["input_dim"], KW["output_dim"], case.get("repeat", False))
    if case["utt"][0] == 7:  # the tight case must be feasible: no repeats in the 7-label utterance
        labels[:7] = np.arange(7) % (KW["output_dim"] - 1)
    T = X.shape[0]
    eng.accumulate_ctc(X, case["utt"], labels, case["lab"])
    loss, dlog = _oracle_step(oracle, X, case["utt"], labels, case["lab"])
    assert np.isfinite(loss)
    assert_close("batch_loss", eng.scalar(_lib.BATCH_LOSS), loss, 2e-5, 0)
    assert eng.scalar(_lib.NUM_FRAMES) == sum(case["lab"])
    # the posteriors come out of a T-step fp32 log-space recursion: round-off grows with the utterance length
    assert_close("dlogits", eng.debug_fetch(_lib.DBG_LOGITS, 0, T), dlog, rtol=1e-4, atol=2e-5 if T < 200 else 2e-4)
    got = engine_grads(eng)
    for k, want in oracle.G.items():
        if k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
            continue  # bias under batch norm: true gradient 0
        assert_close("G[%s]" % k, got[k], want, rtol=2e-4 if T < 200 else 1e-3,
                     atol=(2e-5 if T < 200 else 2e-4) * max(np.abs(want).max(), 1e-3))
    assert_close("avg loss", eng.apply(), oracle.apply(), 2e-5, 0)
    # evaluation mode (moving-average batch norm), loss only; the parameters now carry Adam's round-off amplification
    # (see test_gpu_engine_parity.test_multi_step_training), hence the wider tolerance
    eng.eval_accumulate_ctc(X, case["utt"], labels, case["lab"])
    want = ctc_batch(oracle.forward_logits(X, train=False), case["utt"], labels, case["lab"])[0] / max(sum(case["lab"]), 1)
    assert_close("eval loss", eng.eval_finish(), want, 5e-4, 0)
    eng.close()


def test_ctc_infeasible_utterance_and_errors(gpu):
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(42)
    eng, oracle = make_pair(rng, max_frames=128, **KW)
    X = rng.standard_normal((3 + 20, KW["input_dim"])).astype(np.float32)
    labels = np.array([1, 1, 2, 0, 3, 4], dtype=np.int32)  # first utterance: 3 frames for "1 1 2" needs 4
    eng.accumulate_ctc(X, [3, 20], labels, [3, 3])
    assert eng.scalar(_lib.BATCH_LOSS) == np.inf
    dlog = eng.debug_fetch(_lib.DBG_LOGITS, 0, 23)
    assert not dlog[:3].any() and dlog[3:].any()           # zero gradient for the impossible utterance only
    _, want, _ = ctc_batch(oracle.forward_logits(X), [3, 20], labels, [3, 3])
    assert_close("dlogits", dlog, want, rtol=1e-4, atol=2e-5)
    with pytest.raises(_lib.EngineError, match="outside"):
        eng.accumulate_ctc(X, [3, 20], np.array([1, 1, 2, 0, 3, 8], dtype=np.int32), [3, 3])  # 8 = the blank
    with pytest.raises(ValueError, match="do not match"):
        eng.accumulate_ctc(X, [3, 19], labels, [3, 3])
    eng.close()


def test_ctc_training_reduces_loss_and_is_deterministic(gpu):
    rng = np.random.default_rng(43)
    traces = []
    for rep in range(2):
        eng, _ = make_pair(np.random.default_rng(5), output_too=False, max_frames=256, **dict(KW, init_learning_rate=3e-3))
        r = np.random.default_rng(44)
        utt, lab = [40, 35, 50, 28], [6, 4, 9, 3]
        X, labels = _batch(r, utt, lab, KW["input_dim"], KW["output_dim"])
        trace = []
        for _ in range(80):
            eng.accumulate_ctc(X, utt, labels, lab, last=True)
            trace.append(eng.apply())
        traces.append(trace)
        eng.close()
    assert traces[0] == traces[1]
    assert traces[0][-1] < 0.7 * traces[0][0]


def test_ctc_trainer_end_to_end(gpu, tmp_path):
    """TextBatchDispenser (character targets through TextCoder + aurora4_normalizer) -> CTCTrainer -> engine:
    the first update's loss equals the CTC oracle's on the oracle DNN's logits, training reduces it, evaluate runs"""
    from oracle.dnn_oracle import OracleDNN
    from tfkaldi_amd import _lib, synthetic
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    from tfkaldi_amd.neuralNetworks.trainer import CTCTrainer
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder, target_normalizers
    D, C, U = 8, 2, 4
    F = D * (2 * C + 1)
    lengths = [90, 70, 120, 80, 100, 60, 75, 110]
    paths = synthetic.write_corpus(str(tmp_path), len(lengths), 10, feat_dim=D, lengths=lengths, num_speakers=2)
    text = synthetic.write_text_targets(str(tmp_path), len(lengths))
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], C, max(lengths))
    deferred = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], C, max(lengths),
                                            cmvn_on_device=True)
    coder = target_coder.TextCoder(target_normalizers.aurora4_normalizer)
    disp = batchdispenser.TextBatchDispenser(reader, coder, U, text)
    O = coder.num_labels + 1  # + the blank, last class
    dnn = DNN(O, 2, 48, act.TfActivation(act.Batchnorm(None), "relu"), False)
    tr = CTCTrainer(dnn, F, max(lengths), disp.max_target_length, 3e-3, 1.0, 100, 2, seed=11)
    tr.initialize()
    xs, ys = disp.get_batch()
    assert all(y.dtype == np.uint32 for y in ys) and all(len(y) < len(x) for x, y in zip(xs, ys))
    # oracle: same parameters, the two micro-batches of this batch
    oracle = OracleDNN(F, 2, 48, O, nonlin="relu", batch_norm=True, init_learning_rate=3e-3, num_steps=100)
    for l in range(3):
        oracle.W[l] = tr.engine.get(_lib.WEIGHTS, l).astype(np.float64)
    for k in range(0, U, 2):
        X = np.concatenate(xs[k:k + 2])
        labels = np.concatenate(ys[k:k + 2]).astype(np.int64)
        logits = oracle.forward_logits(X)
        loss, dlog, n = ctc_batch(logits, [len(x) for x in xs[k:k + 2]], labels, [len(y) for y in ys[k:k + 2]])
        oracle.backward_from_dlogits(dlog, loss, n)
    want = oracle.apply()
    first = tr.update(xs, ys)
    assert_close("first loss", first, want, 2e-5, 0)
    losses = [first] + [tr.update(xs, ys) for _ in range(60)]
    assert losses[-1] < 0.6 * losses[0]
    assert np.isfinite(tr.evaluate(xs, ys))
    tr.close()
    # the same batch with CMVN + splice deferred to the device (tfk_accumulate_ctc_raw): bit-identical losses
    disp2 = batchdispenser.TextBatchDispenser(deferred, coder, U, text)
    xs2, ys2 = disp2.get_batch()
    assert type(xs2[0]).__name__ == "Unspliced" and xs2[0].cmvn is not None
    tr2 = CTCTrainer(dnn, F, max(lengths), disp2.max_target_length, 3e-3, 1.0, 100, 2, seed=11)
    tr2.initialize()
    assert [tr2.update(xs2, ys2) for _ in range(3)] == losses[:3]
    tr2.evaluate(xs2, ys2)
    tr2.close()


@pytest.mark.parametrize("T,S", [(800, 100), (1600, 100), (800, 30)])
def test_ctc_long_utterances_stay_accurate(gpu, T, S):
    """log p runs into the thousands here; the recursion keeps its state relative to a double-precision offset
    (re-centred on the maximum every 8 frames), so loss and posteriors stay at fp32 round-off of the INPUTS instead of
    at the 1e-4 resolution of an fp32 number of that magnitude (which gave 3e-3 errors at T = 1600)"""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(T + S)
    kw = dict(KW, output_dim=36)
    eng, oracle = make_pair(rng, max_frames=T, **kw)
    X = (rng.standard_normal((T, kw["input_dim"])) * 1.5).astype(np.float32)
    labels = rng.integers(0, 35, size=S).astype(np.int32)
    eng.accumulate_ctc(X, [T], labels, [S])
    loss, dlog, _ = ctc_batch(oracle.forward_logits(X), [T], labels, [S])
    assert_close("loss", eng.scalar(_lib.BATCH_LOSS), loss, 1e-6, 0)
    assert np.abs(eng.debug_fetch(_lib.DBG_LOGITS, 0, T) - dlog).max() < 5e-4
    eng.close()


def test_ctc_cfg5_size_against_oracle(gpu):
    """BASELINE configs[4] at its stated size: 4x512 stacked DNN over 16 utterances of ~800 frames (12.8 k frames per
    micro-batch), ~100 character labels each, 35 characters + blank -- loss, dLogits and every gradient against the
    float64 CTC + DNN oracles.  (tanh: keeps the ReLU-kink sign question out of a test about the CTC recursion;
    800-frame utterances have log p ~ -2500, which the kernel's re-centred fp32 state vector has to carry.)"""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(47)
    kw = dict(input_dim=440, num_layers=4, num_units=512, output_dim=36, nonlin="tanh", batch_norm=True,
              init_learning_rate=1e-3, num_steps=50)
    utt = [800, 760, 800, 640, 800, 800, 712, 800, 800, 555, 800, 800, 800, 790, 800, 800]
    lab = [100, 96, 100, 80, 100, 110, 90, 100, 100, 70, 100, 104, 100, 99, 100, 100]
    T = int(np.sum(utt))
    eng, oracle = make_pair(rng, max_frames=T, **kw)
    X, labels = _batch(rng, utt, lab, kw["input_dim"], kw["output_dim"])
    eng.accumulate_ctc(X, utt, labels, lab)
    loss, dlog = _oracle_step(oracle, X, utt, labels, lab)
    assert np.isfinite(loss) and loss / sum(lab) > 1.0
    assert_close("batch_loss", eng.scalar(_lib.BATCH_LOSS), loss, 5e-5, 0)
    assert eng.scalar(_lib.NUM_FRAMES) == sum(lab)
    assert_close("dlogits", eng.debug_fetch(_lib.DBG_LOGITS, 0, T), dlog, rtol=1e-4, atol=5e-4)
    got = engine_grads(eng)
    for k, want in oracle.G.items():
        if k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
            continue  # bias under batch norm: true gradient 0
        assert_close("G[%s]" % k, got[k], want, rtol=1e-3, atol=5e-4 * max(np.abs(want).max(), 1e-3))
    assert_close("avg loss", eng.apply(), oracle.apply(), 5e-5, 0)
    eng.close()
