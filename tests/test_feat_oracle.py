"""oracle/feat_oracle.py (the float64 restatement of processing/{feat,base,sigproc,prepare_data}.py) against the golden
vectors the REFERENCE's own code produced (oracle/make_golden_feat.py), plus the host-side pieces of the product's
feature modules (tables, frame counting, kaldi text readers).  CPU only."""
import gzip
import json
import os
import random
import shutil

import numpy as np
import pytest

from oracle import feat_oracle as fo
from tfkaldi_amd import features as dev_features
from tfkaldi_amd.processing import ark, base, feat, prepare_data, readfiles

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "feat_golden.npz"))


def case(gold, name):
    meta = json.loads(str(gold["case_%s_meta" % name]))
    return gold["case_%s_signal" % name], meta, gold["case_%s_features" % name]


def test_oracle_feature_computer_matches_reference(gold):
    for name in gold["case_names"]:
        sig, meta, ref = case(gold, name)
        with np.errstate(all="ignore"):
            got = fo.compute_features(sig, meta["rate"], meta["type"], meta["dynamic"], meta["conf"])
        assert got.shape == ref.shape, name
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        if meta["type"] == "mfcc":
            # the reference applies scipy's FFT-based DCT, the oracle the same transform as a matrix product
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-11, equal_nan=True), name
            assert np.array_equal(got.astype(np.float32), ref.astype(np.float32)), name
        else:
            assert np.array_equal(got, ref, equal_nan=True), name  # bit-exact float64


def test_oracle_building_blocks_match_reference(gold):
    assert np.array_equal(fo.preemphasis(gold["blk_signal"], 0.97), gold["blk_preemph"])
    assert np.array_equal(fo.framesig(gold["blk_preemph"], 400.0, 160.0), gold["blk_frames"])
    assert np.array_equal(fo.framesig(gold["blk_signal"][:333].astype(np.float64), 100.4, 33.6), gold["blk_frames_odd"])
    assert np.array_equal(fo.magspec(gold["blk_frames"], 512), gold["blk_magspec"])
    assert np.array_equal(fo.powspec(gold["blk_frames"], 512), gold["blk_powspec"])
    assert np.array_equal(fo.powspec(gold["blk_frames"], 256), gold["blk_powspec_trunc"])
    for key, args in (("blk_fb_40_512_16k", (40, 512, 16000, 0, 8000)), ("blk_fb_23_512_16k", (23, 512, 16000, 0, 8000)),
                      ("blk_fb_15_256_8k", (15, 256, 8000, 100, 3800)), ("blk_fb_default", ())):
        assert np.array_equal(fo.get_filterbanks(*args), gold[key])
        assert np.array_equal(base.get_filterbanks(*args), gold[key])  # the product's host-side table builder
    assert np.array_equal(fo.hz2mel(gold["blk_hz"]), gold["blk_hz2mel"])
    assert np.array_equal(fo.mel2hz(gold["blk_hz2mel"]), gold["blk_mel2hz"])
    assert np.array_equal(base.hz2mel(gold["blk_hz"]), gold["blk_hz2mel"])
    assert np.array_equal(base.mel2hz(gold["blk_hz2mel"]), gold["blk_mel2hz"])
    assert np.array_equal(fo.lifter_weights(13, 22.0) * gold["blk_cepstra"], gold["blk_lifter22"])
    assert np.array_equal(base.lifter(gold["blk_cepstra"], 22.0), gold["blk_lifter22"])
    assert np.array_equal(base.lifter(gold["blk_cepstra"], 0.0), gold["blk_lifter0"])
    assert np.array_equal(fo.deriv(gold["blk_matrix"]), gold["blk_deriv"])
    assert np.array_equal(fo.delta(gold["blk_matrix"]), gold["blk_delta"])
    assert np.array_equal(fo.ddelta(gold["blk_matrix"]), gold["blk_ddelta"])
    for n in (1, 2, 3, 4, 5):
        assert np.array_equal(fo.deriv(gold["blk_matrix"][:n]), gold["blk_deriv_n%d" % n])
    lens = (300, 400, 559, 560, 561, 16000)
    assert [len(fo.snip(np.zeros(n), 16000, 0.025, 0.01)) for n in lens] == list(gold["blk_snip"])
    assert [len(feat.snip(np.zeros(n), 16000, 0.025, 0.01)) for n in lens] == list(gold["blk_snip"])


def test_oracle_deframesig_and_logpowspec_match_reference(gold):
    fr = gold["blk2_frames"]
    assert np.array_equal(fo.deframesig(fr, 1500, 400.0, 160.0), gold["blk2_deframe"])
    assert np.array_equal(fo.deframesig(fr, 0, 400.0, 160.0), gold["blk2_deframe_full"])
    assert np.array_equal(fo.deframesig(fr * np.hamming(400), 1500, 400.0, 160.0, winfunc=np.hamming), gold["blk2_deframe_hamming"])
    assert np.array_equal(fo.logpowspec(fr, 512), gold["blk2_logpowspec"])
    assert np.array_equal(fo.logpowspec(fr, 512, norm=0), gold["blk2_logpowspec_raw"])
    assert np.array_equal(fo.logpowspec(np.zeros((3, 400)), 512, norm=0), gold["blk2_logpowspec_silence"])
    assert (gold["blk2_logpowspec_silence"] == -300.0).all() and gold["blk2_logpowspec"].max() == 0.0


def test_dct_matrix_is_scipys_orthonormal_dct():
    from scipy.fftpack import dct
    x = np.random.default_rng(3).standard_normal((7, 23))
    ref = dct(x, type=2, axis=1, norm="ortho")[:, :13]
    assert np.allclose(x @ fo.dct_matrix(23, 13), ref, rtol=1e-12, atol=1e-13)
    assert np.allclose(x @ base.dct_matrix(23, 13), ref, rtol=1e-12, atol=1e-13)


def materialise(tmp_path, names_bin, names_txt):
    d = str(tmp_path)
    for n in names_bin:
        shutil.copy(os.path.join(GOLD, "feat_prep_%s.bin" % n), os.path.join(d, n))
    for n in names_txt:
        open(os.path.join(d, n), "w").write(open(os.path.join(GOLD, "feat_prep_%s.txt" % n)).read().replace("@DIR@", d))
    return d


def test_oracle_cmvn_sums_are_the_reference_float32_sums(tmp_path, gold):
    d = materialise(tmp_path, ("feats.ark", "cmvn.ark"), ("feats.scp", "cmvn.scp"))
    feats, cmvn = ark.ArkReader(os.path.join(d, "feats.scp")), ark.ArkReader(os.path.join(d, "cmvn.scp"))
    for spk, utts in (("spkA", ("spkA_u1", "spkA_u2")), ("spkB", ("spkB_u1", "spkB_u2", "spkB_u3"))):
        rows = np.concatenate([feats.read_utt(u) for u in utts])
        stats = fo.cmvn_stats(rows)
        assert stats.dtype == np.float64 and stats.shape == (2, 41)
        assert np.array_equal(stats.astype(np.float32), cmvn.read_utt(spk))  # bit-exact
        assert stats[0, 40] == rows.shape[0] and stats[1, 40] == 0


def test_frame_counting_and_python2_rounding():
    assert dev_features.py2_round(1102.5) == 1103 and dev_features.py2_round(2.5) == 3 and dev_features.py2_round(400.0) == 400
    assert fo.py2_round(1102.5) == 1103
    for slen, want in ((0, 1), (300, 1), (400, 1), (401, 2), (560, 2), (561, 3), (16000, 99)):
        assert dev_features.count_frames(slen, 400, 160) == want == fo.num_frames(slen, 400, 160)


def test_kaldi_text_readers(tmp_path):
    d = str(tmp_path)
    open(os.path.join(d, "segments"), "w").write("s1 recA 0.0 1.5\ns2 recB 0.25 0.75\ns3 recA 1.5 2.0\n")
    seg = readfiles.read_segments(os.path.join(d, "segments"))
    assert list(seg) == ["recA", "recB"]
    assert seg["recA"] == [("s1", 0.0, 1.5), ("s3", 1.5, 2.0)] and seg["recB"] == [("s2", 0.25, 0.75)]
    open(os.path.join(d, "wav.scp"), "w").write("u1 /data/u1.wav\nu2 sph2pipe -f wav /data/u2.sph |\n")
    wavs = readfiles.read_wavfiles(os.path.join(d, "wav.scp"))
    assert list(wavs.items()) == [("u1", ("/data/u1.wav", False)), ("u2", ("sph2pipe -f wav /data/u2.sph |", True))]
    with gzip.open(os.path.join(d, "ali.gz"), "wt") as f:
        f.write("u1 3 3 17 \nu2 1 2\n")
    ali = readfiles.read_alignments(os.path.join(d, "ali.gz"))
    assert list(ali["u1"]) == [3, 3, 17] and list(ali["u2"]) == [1, 2]


def test_shuffle_examples_keeps_every_line(tmp_path, gold):
    d = materialise(tmp_path, (), ("feats.scp",))
    random.seed(5)
    prepare_data.shuffle_examples(d)
    lines = open(os.path.join(d, "feats_shuffled.scp")).read().replace(d, "@DIR@").split("\n")
    assert sorted(lines) == list(gold["prep_shuffled_sorted"])


def test_read_wav_plain_and_piped(tmp_path):
    path = os.path.join(GOLD, "feat_data_spkA_u1.wav.bin")
    rate, samples = prepare_data.read_wav((path, False))
    assert rate == 16000 and samples.dtype == np.int16 and samples.shape == (5000,)
    rate2, piped = prepare_data.read_wav(("cat %s |" % path, True))
    assert rate2 == 16000 and np.array_equal(piped, samples)


def test_empty_signal_raises_like_the_reference():
    conf = dict(winlen='0.025', winstep='0.01', snip_edges='True', include_energy='False')
    with pytest.raises(IndexError):
        feat.FeatureComputer("fbank", "nodelta", conf)(np.zeros(0, dtype=np.int16), 16000)
    with pytest.raises(IndexError):
        fo.compute_features(np.zeros(0, dtype=np.int16), 16000, "fbank", "nodelta",
                            dict(conf, nfilt='40', nfft='512', lowfreq='0', highfreq='-1', preemph='0.97'))


def test_feature_computer_rejects_unknown_types():
    with pytest.raises(Exception, match="unknown feature type"):
        feat.FeatureComputer("plp", "nodelta", {})
    with pytest.raises(Exception, match="unknown dynamic type"):
        feat.FeatureComputer("fbank", "dddelta", {})


def test_unsupported_transform_is_refused_before_any_file_is_touched(tmp_path):
    """numpy's rfft takes any nfft, the device transform a power of two in [32, 4096]: the configuration is checked on the
    host when the FeatureComputer is built -- prepare_data fails BEFORE it removes an existing feats.ark (round-2 advisor)"""
    from tfkaldi_amd.processing import prepare_data
    conf = dict(winlen='0.025', winstep='0.01', nfilt='40', nfft='400', lowfreq='0', highfreq='-1', preemph='0.97',
                include_energy='False', snip_edges='True')
    with pytest.raises(ValueError, match="power of two"):
        feat.FeatureComputer("fbank", "nodelta", conf)
    with pytest.raises(ValueError, match="nfilt"):
        feat.FeatureComputer("fbank", "nodelta", dict(conf, nfft='64'))
    datadir, featdir = tmp_path / "data", tmp_path / "feat"
    datadir.mkdir(); featdir.mkdir()
    (datadir / "wav.scp").write_text("u1 /nonexistent.wav\n")
    (featdir / "feats.ark").write_bytes(b"precious")
    with pytest.raises(ValueError, match="power of two"):
        prepare_data.prepare_data(str(datadir), str(featdir), conf, "fbank", "nodelta")
    assert (featdir / "feats.ark").read_bytes() == b"precious"


def test_reference_import_line_resolves_to_the_gpu_backed_modules():
    """main.py:7 `from processing import ark, prepare_data, feature_reader, batchdispenser, target_coder`"""
    import sys
    from tfkaldi_amd import compat
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "processing" or k.startswith("processing.")
             or k == "neuralNetworks" or k.startswith("neuralNetworks.")}
    try:
        compat.install()
        from processing import ark as a, prepare_data as p, feature_reader as fr, batchdispenser as bd, target_coder as tc  # noqa: F401
        import processing.feat
        import processing.base
        import processing.sigproc
        assert p is prepare_data and processing.feat is feat and processing.base is base
        assert p.__name__ == "tfkaldi_amd.processing.prepare_data"
    finally:
        for k in [k for k in sys.modules if k == "processing" or k.startswith("processing.") or k == "neuralNetworks"
                  or k.startswith("neuralNetworks.")]:
            if k not in saved:
                del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
