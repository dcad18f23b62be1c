"""the +-context splice on the device (tfk_*_raw) and batched decoding.  The device splice
must reproduce the reference's host splice (processing/feature_reader.py:117-156, golden-tested on CPU) so
exactly that everything downstream is BIT-identical to feeding the host-spliced matrix."""
import numpy as np
import pytest

from util import make_pair

pytestmark = pytest.mark.gpu

D, C = 8, 2
F = D * (2 * C + 1)
KW = dict(input_dim=F, num_layers=2, num_units=32, output_dim=11, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-3, num_steps=10)


def host_splice(utt, c):
    """zero-padded splice that (unlike the reference, which returns None) also accepts very short utterances"""
    n, d = utt.shape
    pad = np.zeros((n + 2 * c, d), dtype=np.float32)
    pad[c:c + n] = utt
    return np.concatenate([pad[j:j + n] for j in range(2 * c + 1)], axis=1)


def _utts(rng, lens):
    return [rng.standard_normal((n, D)).astype(np.float32) for n in lens]


def test_raw_accumulate_is_bit_identical(gpu):
    from tfkaldi_amd import _lib
    from tfkaldi_amd.processing.feature_reader import splice
    rng = np.random.default_rng(0)
    lens = [9, 5, 1, 2, 17, 30]  # includes utterances shorter than the context
    utts = _utts(rng, lens)
    for u in utts:
        if u.shape[0] >= 2 * C + 1:
            assert (host_splice(u, C) == splice(u, C)).all()  # same thing as the golden-tested product splice
    X = np.concatenate([host_splice(u, C) for u in utts])
    raw = np.concatenate(utts)
    T = raw.shape[0]
    y = rng.integers(0, KW["output_dim"], size=T).astype(np.int32)
    a, _ = make_pair(np.random.default_rng(1), **KW)
    b, _ = make_pair(np.random.default_rng(1), **KW)
    for step in range(2):
        a.accumulate(X, y)
        b.accumulate_raw(raw, y, lens, C)
        for l in range(a.L):
            assert (a.debug_fetch(_lib.DBG_HIDDEN, l, T) == b.debug_fetch(_lib.DBG_HIDDEN, l, T)).all()
        assert (a.debug_fetch(_lib.DBG_LOGITS, 0, T) == b.debug_fetch(_lib.DBG_LOGITS, 0, T)).all()
        assert a.apply() == b.apply()
    a.eval_accumulate(X, y)
    b.eval_accumulate_raw(raw, y, lens, C)
    assert a.eval_finish() == b.eval_finish()
    # batched decode: all utterances in one pass == one pass per utterance
    post = b.posteriors_raw(raw, lens, C)
    start = 0
    for u in utts:
        n = u.shape[0]
        assert (post[start:start + n] == a.posteriors(host_splice(u, C))).all()
        start += n
    with pytest.raises(_lib.EngineError, match="sum to"):
        b.lib  # noqa: B018
        _lib.check(b.lib.tfk_accumulate_raw(b._h, raw.ctypes.data, D, y.ctypes.data, T,
                                            np.array([3, 4], dtype=np.int32).ctypes.data, 2, C, None, 0))
    with pytest.raises(_lib.EngineError, match="multiple"):
        b.accumulate_raw(raw, y, lens, 3)  # F = 40 is not a multiple of 7
    a.close(); b.close()


def test_device_cmvn_is_bit_identical(gpu):
    """raw frames + per-utterance (mean, std) table: the device's (x - mean) / std then splice == numpy's"""
    from tfkaldi_amd import _lib
    from tfkaldi_amd.processing.feature_reader import apply_cmvn, cmvn_params
    rng = np.random.default_rng(5)
    lens = [12, 3, 25, 7]
    utts = [(rng.standard_normal((n, D)) * rng.uniform(0.5, 30.0, size=D) + rng.uniform(-50, 50, size=D))
            .astype(np.float32) for n in lens]
    stats = []
    for u in utts:  # Kaldi-style accumulated statistics, float32 as ArkWriter stores them
        st = np.zeros((2, D + 1), dtype=np.float32)
        pool = np.concatenate([u, rng.standard_normal((40, D)).astype(np.float32) * 5 + u.mean(0)])
        st[0, :-1], st[0, -1], st[1, :-1] = pool.sum(0), pool.shape[0], np.square(pool).sum(0)
        stats.append(st)
    normalised = [apply_cmvn(u, st) for u, st in zip(utts, stats)]
    assert all(n.dtype == np.float32 for n in normalised)
    X = np.concatenate([host_splice(n, C) for n in normalised])
    raw = np.concatenate(utts)
    table = np.stack([np.stack(cmvn_params(st)) for st in stats])
    T = raw.shape[0]
    y = rng.integers(0, KW["output_dim"], size=T).astype(np.int32)
    a, _ = make_pair(np.random.default_rng(1), **KW)
    b, _ = make_pair(np.random.default_rng(1), **KW)
    for step in range(2):
        a.accumulate(X, y)
        b.accumulate_raw(raw, y, lens, C, cmvn=table)
        for l in range(a.L):
            assert (a.debug_fetch(_lib.DBG_HIDDEN, l, T) == b.debug_fetch(_lib.DBG_HIDDEN, l, T)).all()
        assert a.apply() == b.apply()
    a.eval_accumulate(X, y)
    b.eval_accumulate_raw(raw, y, lens, C, cmvn=table)
    assert a.eval_finish() == b.eval_finish()
    assert (a.posteriors(X) == b.posteriors_raw(raw, lens, C, cmvn=table)).all()
    # a larger table than the first one (buffers grow) and then none at all
    lens2 = [4] * 9
    raw2 = rng.standard_normal((36, D)).astype(np.float32)
    table2 = np.stack([np.stack([rng.standard_normal(D), rng.uniform(0.5, 2, D)]) for _ in lens2]).astype(np.float32)
    want = np.concatenate([host_splice((raw2[4 * i:4 * i + 4] - table2[i, 0]) / table2[i, 1], C) for i in range(9)])
    assert (a.posteriors(want) == b.posteriors_raw(raw2, lens2, C, cmvn=table2)).all()
    assert (a.posteriors(np.concatenate([host_splice(raw2[4 * i:4 * i + 4], C) for i in range(9)]))
            == b.posteriors_raw(raw2, lens2, C)).all()
    with pytest.raises(ValueError, match="cmvn table"):
        b.posteriors_raw(raw2, lens2, C, cmvn=table2[:3])
    a.close(); b.close()


def test_feature_reader_to_trainer_and_decoder(gpu, tmp_path):
    """FeatureReader(splice_on_device=True) -> dispenser -> Trainer.update / evaluate and Decoder give exactly
    the results of the host-splice pipeline."""
    from tfkaldi_amd import synthetic
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    from tfkaldi_amd.neuralNetworks.decoder import Decoder
    from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    O = KW["output_dim"]
    lengths = np.random.default_rng(3).integers(5, 40, size=14)
    paths = synthetic.write_corpus(str(tmp_path / "data"), 14, O, feat_dim=D, lengths=lengths, num_speakers=2)

    def run(on_device):
        reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], C, 40,
                                              splice_on_device=on_device == 1, cmvn_on_device=on_device == 2)
        disp = batchdispenser.AlignmentBatchDispenser(reader, target_coder.AlignmentCoder(lambda x, y: x, O), 4,
                                                      paths["alignments"])
        dnn = DNN(O, 2, 32, act.TfActivation(act.Batchnorm(None), "relu"), False)
        tr = CrossEnthropyTrainer(dnn, F, 40, disp.max_target_length, 1e-2, 1.0, 10, 3, seed=7)
        tr.initialize()
        val = disp.get_batch()
        losses = [tr.update(*disp.get_batch()) for _ in range(2)]
        losses.append(tr.evaluate(*val))
        tr.save_model(str(tmp_path / ("model%d" % on_device)))
        tr.close()
        dec = Decoder(dnn, F, 40)
        dec.restore(str(tmp_path / ("model%d" % on_device)))
        dec.set_prior(np.full(O, 1.0 / O, dtype=np.float32))
        single = [dec.log_likelihoods(u) for u in val[0]]
        batched = dec.decode_batch(val[0])
        post = dec(val[0][0])
        dec.close()
        return losses, single, batched, post, type(val[0][0]).__name__

    l0, s0, b0, p0, t0 = run(0)
    for mode in (1, 2):  # 1: splice on the device, 2: CMVN + splice on the device
        l1, s1, b1, p1, t1 = run(mode)
        assert (t0, t1) == ("ndarray", "Unspliced")
        assert l0 == l1
        assert (p0 == p1).all()
        for x0, x1, y0, y1 in zip(s0, s1, b0, b1):
            assert (x0 == x1).all() and (y0 == y1).all() and (x0 == y0).all()
