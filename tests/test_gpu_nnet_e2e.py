"""The reference-facing Python API end to end on a GPU: Trainer / Decoder / Nnet over synthetic ark data,
checked against the float64 oracle fed with the same utterances, weights and micro-batch split."""
import configparser
import os

import numpy as np
import pytest

from oracle.dnn_oracle import OracleDNN, reference_microbatches
from util import assert_close

pytestmark = pytest.mark.gpu

F_RAW, CONTEXT, O = 8, 2, 12


def _corpus(tmp_path, num_utt=26):
    from tfkaldi_amd import synthetic
    lengths = np.random.default_rng(0).integers(6, 30, size=num_utt)
    lengths[3] = 3  # too short to splice with context 2 -> skipped with a WARNING
    return synthetic.write_corpus(str(tmp_path / "data"), num_utt, O, feat_dim=F_RAW, lengths=lengths, num_speakers=3), lengths


def _dispenser(paths, size):
    from tfkaldi_amd.processing import batchdispenser, feature_reader, target_coder
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, 30)
    coder = target_coder.AlignmentCoder(lambda x, y: x, O)
    return batchdispenser.AlignmentBatchDispenser(reader, coder, size, paths["alignments"])


def _oracle_like(trainer, dnn, seed, **kw):
    """an oracle holding what Trainer.initialize() put on the GPU (same rng stream)"""
    from tfkaldi_amd import _lib
    o = OracleDNN(**kw)
    for l in range(o.L + 1):
        o.W[l] = trainer.engine.get(_lib.WEIGHTS, l).astype(np.float64)
    return o


def test_trainer_update_matches_oracle(gpu, tmp_path):
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer, CTCTrainer
    paths, _ = _corpus(tmp_path)
    disp = _dispenser(paths, 5)
    F = F_RAW * (2 * CONTEXT + 1)
    dnn = DNN(O, 2, 24, act.TfActivation(act.Batchnorm(None), "relu"), False)
    trainer = CrossEnthropyTrainer(dnn, F, disp.max_input_length, disp.max_target_length, 1e-2, 0.5, 20, 2, seed=5)
    trainer.initialize()
    kw = dict(input_dim=F, num_layers=2, num_units=24, output_dim=O, nonlin="relu", batch_norm=True,
              init_learning_rate=1e-2, learning_rate_decay=0.5, num_steps=20)
    oracle = _oracle_like(trainer, dnn, 5, **kw)
    assert trainer.control_ops is None and trainer.global_step.eval() == 0
    val_x, val_y = disp.get_batch()
    for step in range(4):
        xs, ys = disp.get_batch()
        got = trainer.update(xs, ys)
        # 5 utterances in micro-batches of 2 -> the reference quirk: [0,1], [2,3], [4]
        for idx in reference_microbatches(len(xs), 2):
            oracle.accumulate(np.concatenate([xs[i] for i in idx]), np.concatenate([ys[i] for i in idx]))
        assert_close("loss %d" % step, got, oracle.apply(), 1e-4 * (step + 1), 0)
        if step == 1:
            trainer.halve_learning_rate(); oracle.halve_learning_rate()
    for idx in reference_microbatches(len(val_x), 2):
        oracle.eval_accumulate(np.concatenate([val_x[i] for i in idx]), np.concatenate([val_y[i] for i in idx]))
    assert_close("valid", trainer.evaluate(val_x, val_y), oracle.eval_finish(), 2e-4, 0)
    assert trainer.evaluate(None, None) is None
    # checkpoint round trip: model + train variables, Adam state untouched (reference trainer.py:204-205)
    prefix = str(tmp_path / "ckpt")
    trainer.save_trainer(prefix)
    assert os.path.isfile(prefix) and os.path.isfile(prefix + "_trainvars")
    xs, ys = disp.get_batch()
    before = trainer.evaluate(val_x, val_y)
    trainer.update(xs, ys)
    assert trainer.global_step.eval() == 5
    trainer.restore_trainer(prefix)
    assert trainer.global_step.eval() == 4
    assert trainer.evaluate(val_x, val_y) == before  # bit-identical parameters back
    with pytest.raises(ValueError):
        trainer.update([xs[0]], [ys[0][:-1]])
    trainer.close()
    ctc = CTCTrainer(dnn, F, 30, 30, 1e-3, 1.0, 10, 2)  # the CTC loss has its own tests (test_gpu_ctc.py)
    assert ctc.loss_kind == "ctc"
    ctc.close()


def test_nnet_train_and_decode(gpu, tmp_path, capsys):
    from tfkaldi_amd import compat
    from tfkaldi_amd.processing import ark, feature_reader
    compat.install()
    from neuralNetworks import nnet  # the reference's import line
    paths, lengths = _corpus(tmp_path)
    conf = configparser.ConfigParser()
    conf.add_section("directories"); conf.set("directories", "expdir", str(tmp_path))
    conf.add_section("nnet")
    for k, v in dict(name="dnn", context_width=str(CONTEXT), num_hidden_units="32", num_hidden_layers="2",
                     add_layer_period="3", starting_step="0", nonlin="relu", l2_norm="False", dropout="1",
                     batch_norm="True", num_epochs="2", initial_learning_rate="0.01", learning_rate_decay="1",
                     batch_size="4", numutterances_per_minibatch="2", valid_batches="1", valid_frequency="2",
                     valid_adapt="False", valid_retries="3", check_freq="4", visualise="True").items():
        conf.set("nnet", k, v)
    net = nnet.Nnet(conf, F_RAW, O)
    disp = _dispenser(paths, 4)
    net.train(disp)
    out = capsys.readouterr().out
    assert "validation loss at step 0:" in out and "step 0/" in out and "adding layer" in out
    assert "WARNING utt000003 is too short to splice" in out
    savedir = str(tmp_path / "dnn")
    for f in ("final", "prior.npy", "training/validated", "training/validated_trainvars", "logdir/summaries.jsonl"):
        assert os.path.isfile(os.path.join(savedir, f)), f
    prior = np.load(os.path.join(savedir, "prior.npy"))
    assert prior.dtype == np.float32 and abs(prior.sum() - 1) < 1e-6
    # first validation loss of an untrained net (zero output layer) is ln O
    first = float(out.split("validation loss at step 0: ")[1].split("\n")[0])
    assert abs(first - np.log(O)) < 1e-4

    # decode the corpus and compare with the oracle evaluated on the saved model
    reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, 30)
    # drop the too-short utterance from the decode list (the reference would crash on it too)
    keep = [i for i, u in enumerate(reader.reader.utt_ids) if lengths[i] >= 2 * CONTEXT + 1]
    reader.reader.utt_ids = [reader.reader.utt_ids[i] for i in keep]
    reader.reader.scp_data = [reader.reader.scp_data[i] for i in keep]
    decodedir = tmp_path / "decode"; decodedir.mkdir()
    writer = ark.ArkWriter(str(decodedir / "feats.scp"), str(decodedir / "likelihoods.ark"))
    net.decode(reader, writer)
    model = np.load(os.path.join(savedir, "final"))
    F = F_RAW * (2 * CONTEXT + 1)
    o = OracleDNN(input_dim=F, num_layers=2, num_units=32, output_dim=O, nonlin="relu", batch_norm=True,
                  layerwise_init=True)
    for l in range(3):
        o.W[l] = model["layer%d/weights" % l].astype(np.float64)
        o.b[l] = model["layer%d/biases" % l].astype(np.float64)
    for l in range(2):
        o.beta[l] = model["layer%d/batch_norm/beta" % l].astype(np.float64)
        o.mov_mean[l] = model["layer%d/batch_norm/moving_mean" % l].astype(np.float64)
        o.mov_var[l] = model["layer%d/batch_norm/moving_variance" % l].astype(np.float64)
    o.initialisedlayers = int(model["initialisedlayers"])
    assert o.initialisedlayers >= 1
    like = ark.ArkReader(str(decodedir / "feats.scp"))
    ref_reader = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, 30)
    n = 0
    while True:
        uid, mat, looped = like.read_next_utt()
        if looped:
            break
        x = feature_reader.splice(feature_reader.apply_cmvn(ref_reader.reader.read_utt(uid),
                                                           ref_reader.reader_cmvn.read_utt(ref_reader.utt2spk[uid])), CONTEXT)
        want = np.log(o.posteriors(x) / prior)
        assert mat.dtype == np.float32 and mat.shape == want.shape
        assert_close("loglik " + uid, mat, want, 2e-4, 2e-4)
        n += 1
    assert n == len(keep)
    # the same decode with the splice on the device and small decode batches: identical archive contents
    reader2 = feature_reader.FeatureReader(paths["feats_scp"], paths["cmvn_scp"], paths["utt2spk"], CONTEXT, 30,
                                           splice_on_device=True)
    reader2.reader.utt_ids = [reader2.reader.utt_ids[i] for i in keep]
    reader2.reader.scp_data = [reader2.reader.scp_data[i] for i in keep]
    net.conf["decode_batch_frames"] = "40"
    writer2 = ark.ArkWriter(str(decodedir / "feats2.scp"), str(decodedir / "likelihoods2.ark"))
    net.decode(reader2, writer2)
    assert open(str(decodedir / "likelihoods2.ark"), "rb").read() == open(str(decodedir / "likelihoods.ark"), "rb").read()


def test_dnn_call_signature(gpu):
    """Classifier.__call__(inputs, seq_length, is_training, reuse, scope) -> (logits, seq_length, saver, control_ops)"""
    from tfkaldi_amd.neuralNetworks.classifiers import activation as act
    from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN
    rng = np.random.default_rng(0)
    dnn = DNN(7, 2, 16, act.TfActivation(None, "tanh"), True)
    lens = [4, 6]
    inputs = [rng.standard_normal((2, 10)).astype(np.float32) for _ in range(6)]
    logits, seq_length, saver, control_ops = dnn(inputs, lens, scope="Classifier")
    assert len(logits) == 6 and logits[0].shape == (2, 7) and set(control_ops) == {"add", "init"}
    assert (np.stack(logits) == 0).all()  # zero output layer => zero logits (and zero padding)
    with pytest.raises(ValueError):
        dnn(inputs, lens, scope="Classifier")  # variables exist already
    again, _, _, _ = dnn(inputs, lens, reuse=True, scope="Classifier")
    assert (np.stack(again) == np.stack(logits)).all()
    with pytest.raises(ValueError):
        dnn(inputs, lens, reuse=True, scope="Other")


def test_minimal_classifier_subclass_trains(gpu, tmp_path):
    """The extension contract of neuralNetworks/classifiers/classifier.py: a user classifier that defines ONLY
    engine_config() (a 3 x 20 sigmoid net without batch norm -- not something DNN's constructor arguments of this
    test would give) is trainable through CrossEnthropyTrainer, evaluable through the reference's call signature and
    decodable through Decoder, and matches the float64 oracle of that network."""
    from tfkaldi_amd import _lib
    from tfkaldi_amd.neuralNetworks.classifiers.classifier import Classifier
    from tfkaldi_amd.neuralNetworks.decoder import Decoder
    from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer

    class Sigmoid3(Classifier):
        def engine_config(self, input_dim, **trainer_options):
            return _lib.make_config(input_dim, 3, 20, self.output_dim, nonlin="sigmoid", **trainer_options)

    paths, _ = _corpus(tmp_path)
    disp = _dispenser(paths, 4)
    F = F_RAW * (2 * CONTEXT + 1)
    net = Sigmoid3(O)
    trainer = CrossEnthropyTrainer(net, F, disp.max_input_length, disp.max_target_length, 1e-2, 1.0, 20, 2, seed=11)
    trainer.initialize()
    assert trainer.control_ops is None
    oracle = _oracle_like(trainer, net, 11, input_dim=F, num_layers=3, num_units=20, output_dim=O, nonlin="sigmoid",
                          batch_norm=False, init_learning_rate=1e-2, learning_rate_decay=1.0, num_steps=20)
    for step in range(3):
        xs, ys = disp.get_batch()
        got = trainer.update(xs, ys)
        for idx in reference_microbatches(len(xs), 2):
            oracle.accumulate(np.concatenate([xs[i] for i in idx]), np.concatenate([ys[i] for i in idx]))
        assert_close("loss %d" % step, got, oracle.apply(), 1e-4 * (step + 1), 0)
    prefix = str(tmp_path / "sigmoid3")
    trainer.save_model(prefix)
    xs, _ = disp.get_batch()
    dec = Decoder(net, F, 64)
    dec.restore(prefix)
    assert_close("posteriors", dec(xs[0]), oracle.posteriors(xs[0]), 5e-3, 1e-6)
    dec.close()
    trainer.close()
    # the reference's call signature (classifier.py:16-37), inference mode, fresh variables in their own scope
    seq = [np.random.default_rng(1).standard_normal((2, F)).astype(np.float32) for _ in range(5)]
    logits, lens, saver, ops = net(seq, [5, 3], scope="probe")
    assert len(logits) == 5 and logits[0].shape == (2, O) and ops is None and list(lens) == [5, 3]
    assert not np.any(logits[0])  # zero output layer: dnn.py:67-68
    with pytest.raises(NotImplementedError):
        net(seq, [5, 3], is_training=True, reuse=True, scope="probe")
