"""The reference's main.py, minus Kaldi (GMM alignments come from a file here): the statements of main.py:33-64 and
:129-186 run unchanged in meaning through the reference's own import lines -- wav files -> prepare_data / compute_cmvn /
shuffle_examples -> FeatureReader + AlignmentBatchDispenser -> Nnet.train -> Nnet.decode -> pseudo-log-likelihood ark --
every stage on the GPU.  A learnable toy corpus (the alignment of a frame is a function of its dominant tone), so that the
acceptance criterion is the one a user has: the validation loss falls well below ln(num_labels)."""
import configparser
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATE, NUM_LABELS = 16000, 6
FEAT = dict(name="40fbank", type="fbank", dynamic="nodelta", winlen="0.025", winstep="0.01", nfilt="40", nfft="512",
            lowfreq="0", highfreq="-1", preemph="0.97", include_energy="False", snip_edges="True")


def _toy_corpus(root, n_utts, seed):
    """utterances made of 0.12 s segments, each a tone whose frequency band IS the label of its frames"""
    import scipy.io.wavfile as wav
    rng = np.random.default_rng(seed)
    os.makedirs(root)
    seg = int(0.12 * RATE)
    utts, labels = [], {}
    for i in range(n_utts):
        uid = "spk%d_utt%03d" % (i % 3, i)
        segs = rng.integers(0, NUM_LABELS, size=int(rng.integers(4, 9)))
        t = np.arange(seg) / RATE
        x = np.concatenate([3000 * np.sin(2 * np.pi * (300 + 900 * s) * t + rng.uniform(0, 6)) for s in segs])
        x = x + 200 * rng.standard_normal(x.size)
        wav.write(os.path.join(root, uid + ".wav"), RATE, np.round(x).astype(np.int16))
        utts.append(uid)
        labels[uid] = segs
    utts.sort()
    open(os.path.join(root, "wav.scp"), "w").write("".join("%s %s\n" % (u, os.path.join(root, u + ".wav")) for u in utts))
    open(os.path.join(root, "utt2spk"), "w").write("".join("%s %s\n" % (u, u[:4]) for u in utts))
    spk = {}
    for u in utts:
        spk.setdefault(u[:4], []).append(u)
    open(os.path.join(root, "spk2utt"), "w").write("".join("%s %s\n" % (s, " ".join(us)) for s, us in sorted(spk.items())))
    open(os.path.join(root, "text"), "w").write("".join("%s X\n" % u for u in utts))
    return utts, labels


def _alignments(featdir, labels, path):
    """one pdf-id per frame: the label of the segment the frame's centre falls into (what a GMM alignment would say)"""
    from processing import ark
    reader = ark.ArkReader(featdir + "/feats.scp")
    with gzip.open(path, "wt") as f:
        for uid in reader.utt_ids:
            n = reader.read_utt(uid).shape[0]
            centre = (np.arange(n) * 160 + 200) / (0.12 * RATE)
            ids = labels[uid][np.minimum(centre.astype(int), len(labels[uid]) - 1)]
            f.write("%s %s\n" % (uid, " ".join(str(int(k)) for k in ids)))


def test_main_flow_from_wav_files_to_likelihoods(gpu, tmp_path, capsys):
    import tfkaldi_amd.compat
    tfkaldi_amd.compat.install()
    from neuralNetworks import nnet                                                            # main.py:6
    from processing import ark, prepare_data, feature_reader, batchdispenser, target_coder     # main.py:7
    root = str(tmp_path)
    train_utts, train_labels = _toy_corpus(root + "/data_train", 60, seed=1)
    test_utts, test_labels = _toy_corpus(root + "/data_test", 8, seed=2)
    config = configparser.ConfigParser()
    config.read_dict({
        "directories": dict(train_data=root + "/data_train", test_data=root + "/data_test", train_features=root + "/feat_train",
                            test_features=root + "/feat_test", expdir=root + "/exp"),
        "dnn-features": FEAT,
        "nnet": dict(name="dnn", gmm_name="gmm", context_width="3", num_hidden_units="64", num_hidden_layers="2",
                     add_layer_period="0", starting_step="0", nonlin="relu", l2_norm="False", dropout="1", batch_norm="True",
                     num_epochs="6", initial_learning_rate="0.005", learning_rate_decay="1", batch_size="8",
                     numutterances_per_minibatch="4", valid_batches="1", valid_frequency="5", valid_adapt="False",
                     valid_retries="3", check_freq="10", visualise="False"),
    })
    os.makedirs(root + "/exp/gmm/ali")
    feat_cfg = dict(config.items("dnn-features"))
    # main.py:44-52, 66-74: features + CMVN statistics of the training and the test set
    for which in ("train", "test"):
        featdir = config.get("directories", which + "_features") + "/" + feat_cfg["name"]
        prepare_data.prepare_data(config.get("directories", which + "_data"), featdir, feat_cfg, feat_cfg["type"], feat_cfg["dynamic"])
        prepare_data.compute_cmvn(featdir)
    featdir = config.get("directories", "train_features") + "/" + config.get("dnn-features", "name")
    alifile = root + "/exp/gmm/ali/pdf.all"
    _alignments(featdir, train_labels, alifile)
    # main.py:117-123: input dimension from the first utterance
    reader = ark.ArkReader(featdir + "/feats.scp")
    _, features, _ = reader.read_next_utt()
    input_dim = features.shape[1]
    assert input_dim == 40
    net = nnet.Nnet(config, input_dim, NUM_LABELS)                                             # main.py:129
    # main.py:133-157: shuffle, reader, coder, dispenser, train
    prepare_data.shuffle_examples(featdir)
    with open(featdir + "/maxlength", "r") as fid:
        max_input_length = int(fid.read())
    featreader = feature_reader.FeatureReader(featdir + "/feats_shuffled.scp", featdir + "/cmvn.scp", featdir + "/utt2spk",
                                              int(config.get("nnet", "context_width")), max_input_length)
    coder = target_coder.AlignmentCoder(lambda x, y: x, NUM_LABELS)
    dispenser = batchdispenser.AlignmentBatchDispenser(featreader, coder, int(config.get("nnet", "batch_size")), alifile)
    net.train(dispenser)
    out = capsys.readouterr().out
    losses = [float(line.split(": ")[1]) for line in out.split("\n") if line.startswith("validation loss at step")]
    assert abs(losses[0] - np.log(NUM_LABELS)) < 1e-3          # zero output layer at the start (dnn.py:67-68)
    assert losses[-1] < 0.5 * losses[0], losses                 # it learns the tone -> label map
    # main.py:160-182: decode the test set into a likelihood ark
    savedir = config.get("directories", "expdir") + "/" + config.get("nnet", "name")
    decodedir = savedir + "/decode"
    os.mkdir(decodedir)
    featdir = config.get("directories", "test_features") + "/" + config.get("dnn-features", "name")
    with open(featdir + "/maxlength", "r") as fid:
        max_length = int(fid.read())
    featreader = feature_reader.FeatureReader(featdir + "/feats.scp", featdir + "/cmvn.scp", featdir + "/utt2spk",
                                              int(config.get("nnet", "context_width")), max_length)
    writer = ark.ArkWriter(decodedir + "/feats.scp", decodedir + "/likelihoods.ark")
    net.decode(featreader, writer)
    like = ark.ArkReader(decodedir + "/feats.scp")
    assert like.utt_ids == test_utts
    prior = np.load(savedir + "/prior.npy")
    hits = total = 0
    for uid in test_utts:
        mat = like.read_utt(uid)                                 # log(posterior / prior), [frames, num_labels]
        assert mat.dtype == np.float32 and mat.shape[1] == NUM_LABELS and np.isfinite(mat).all()
        n = mat.shape[0]
        centre = (np.arange(n) * 160 + 200) / (0.12 * RATE)
        want = test_labels[uid][np.minimum(centre.astype(int), len(test_labels[uid]) - 1)]
        inner = (np.abs(centre - np.round(centre)) > 0.25)       # frames that do not straddle a segment boundary
        hits += int(((mat + np.log(prior)).argmax(1) == want)[inner].sum())
        total += int(inner.sum())
    assert hits / total > 0.8, (hits, total)                     # unseen utterances, frame accuracy away from the boundaries
