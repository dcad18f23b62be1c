"""GPU parity of the feature computation (csrc/features.hip behind processing/{sigproc,base,feat,prepare_data}.py)
against (a) the golden vectors the REFERENCE's own numpy code produced (tests/golden/feat_*, made by
oracle/make_golden_feat.py) and (b) the float64 oracle at larger, randomised sizes.

Tolerances.  The reference computes in float64 and the ark files store float32.  The device computes in float64 as
well, with its own transform (radix-2 instead of pocketfft) and summation orders, so features agree to ~1e-12 relative
and are compared with rtol 1e-9 / atol 1e-9 in float64; after the cast to float32 they must be IDENTICAL except for
values that sit on a rounding boundary (<= 1 float32 ulp, fewer than 1 in 1000).  Pre-emphasis, framing, the delta
filters and the CMVN sums have a defined operation order and are bit-exact."""
import io
import json
import os
import shutil
import zlib
from contextlib import redirect_stdout

import numpy as np
import pytest

from oracle import feat_oracle as fo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

AURORA_DNN = dict(winlen='0.025', winstep='0.01', nfilt='40', nfft='512', lowfreq='0', highfreq='-1', preemph='0.97',
                  include_energy='False', snip_edges='True')
AURORA_GMM = dict(AURORA_DNN, nfilt='23', numcep='13', ceplifter='22')


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "feat_golden.npz"))


def close64(got, ref, what=""):
    assert got.shape == ref.shape and got.dtype == np.float64, what
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-9, equal_nan=True), \
        "%s: max abs diff %.3e" % (what, np.nanmax(np.abs(got - ref)))


def same32(got, ref, what=""):
    """float32 views: identical but for values on a rounding boundary"""
    a, b = np.asarray(got, dtype=np.float32), np.asarray(ref, dtype=np.float32)
    assert a.shape == b.shape, what
    finite = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), finite), what
    if a.size == 0:
        return
    ulps = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))[finite]
    assert ulps.size == 0 or ulps.max() <= 1, "%s: %d float32 ulps apart" % (what, ulps.max())
    assert (ulps != 0).sum() <= max(1, ulps.size // 1000), "%s: %d of %d values differ" % (what, (ulps != 0).sum(), ulps.size)


def test_feature_computer_matches_reference_golden(gold):
    from tfkaldi_amd.processing import feat
    for name in gold["case_names"]:
        meta = json.loads(str(gold["case_%s_meta" % name]))
        sig, ref = gold["case_%s_signal" % name], gold["case_%s_features" % name]
        comp = feat.FeatureComputer(meta["type"], meta["dynamic"], meta["conf"])
        got = comp(sig, meta["rate"])
        close64(got, ref, name)
        same32(got, ref, name)
        got32 = comp.compute_batch([sig], meta["rate"])[0]
        assert got32.dtype == np.float32
        same32(got32, ref, name + " (float32 output)")


def test_batch_equals_one_by_one_and_mixed_sample_types(gold):
    from tfkaldi_amd.processing import feat
    rng = np.random.default_rng(11)
    comp = feat.FeatureComputer("mfcc", "ddelta", dict(AURORA_GMM, include_energy='True', snip_edges='False'))
    sigs = [(rng.standard_normal(n) * 2500).astype(np.int16) for n in (16000, 401, 7, 1234, 5000, 400, 9999)]
    batch = comp.compute_batch(sigs, 16000, dtype=np.float64)
    for s, b in zip(sigs, batch):
        assert np.array_equal(comp(s, 16000), b)  # the batch layout changes nothing, bit for bit
        with np.errstate(all="ignore"):
            close64(b, fo.compute_features(s, 16000, "mfcc", "ddelta", comp.conf))
    # float64 samples next to int16 ones: each sample class goes in its own device pass, the results come back in order
    mixed = comp.compute_batch([sigs[0], sigs[3].astype(np.float64)], 16000, dtype=np.float64)
    assert np.array_equal(mixed[0], batch[0]) and np.array_equal(mixed[1], batch[3])
    assert comp.compute_batch([], 16000) == []


@pytest.mark.parametrize("ftype,dyn,conf,rate", [
    ("fbank", "nodelta", AURORA_DNN, 16000),
    ("mfcc", "delta", AURORA_GMM, 16000),
    ("ssc", "ddelta", dict(AURORA_DNN, nfilt='26', include_energy='True'), 16000),
    ("fbank", "ddelta", dict(AURORA_DNN, nfft='256', nfilt='24', snip_edges='False', lowfreq='64', highfreq='3700'), 8000),
    ("mfcc", "nodelta", dict(AURORA_GMM, nfft='2048', winlen='0.032', winstep='0.008', nfilt='64', numcep='40'), 44100),
    ("fbank", "delta", dict(AURORA_DNN, nfft='4096', winlen='0.064', nfilt='128', preemph='0.9'), 48000),
])
def test_random_utterances_against_oracle(ftype, dyn, conf, rate):
    """a few hundred utterances of ragged lengths in one device pass (about a minute of audio)"""
    from tfkaldi_amd.processing import feat
    rng = np.random.default_rng(zlib.crc32(("%s/%s/%d" % (ftype, dyn, rate)).encode()))
    lens = rng.integers(1, 3 * rate // 4, size=160)
    t = [np.arange(n) / rate for n in lens]
    sigs = [np.round(3000 * np.sin(2 * np.pi * rng.uniform(80, 3000) * ti) + 400 * rng.standard_normal(ti.size)).astype(np.int16)
            for ti in t]
    comp = feat.FeatureComputer(ftype, dyn, conf)
    got = comp.compute_batch(sigs, rate, dtype=np.float64)
    frames = 0
    for s, g in zip(sigs, got):
        with np.errstate(all="ignore"):
            ref = fo.compute_features(s, rate, ftype, dyn, conf)
        close64(g, ref, "%s/%s len %d" % (ftype, dyn, s.size))
        frames += g.shape[0]
    same32(np.concatenate(got), np.concatenate([fo.compute_features(s, rate, ftype, dyn, conf) for s in sigs]))
    assert frames > 3000


def test_sigproc_blocks(gold):
    from tfkaldi_amd.processing import sigproc
    x = gold["blk_signal"]
    pre = sigproc.preemphasis(x, 0.97)
    assert pre.dtype == np.float64 and np.array_equal(pre, gold["blk_preemph"])           # bit-exact
    assert np.array_equal(sigproc.preemphasis(x.astype(np.float64), 0.0), x.astype(np.float64))
    x32 = (x / 32768.0).astype(np.float32)          # a float32 wav: numpy keeps the pre-emphasis in float32
    pre32 = sigproc.preemphasis(x32, 0.97)
    assert pre32.dtype == np.float32 and np.array_equal(pre32, fo.preemphasis(x32, 0.97).astype(np.float32))
    assert np.array_equal(pre32, np.append(x32[0], x32[1:] - 0.97 * x32[:-1]))
    fr = sigproc.framesig(gold["blk_preemph"], 400.0, 160.0)
    assert np.array_equal(fr, gold["blk_frames"])                                          # bit-exact
    assert np.array_equal(sigproc.framesig(x[:333].astype(np.float64), 100.4, 33.6), gold["blk_frames_odd"])
    ham = sigproc.framesig(gold["blk_preemph"], 400.0, 160.0, winfunc=np.hamming)
    assert np.array_equal(ham, gold["blk_frames"] * np.hamming(400))
    scale = np.abs(gold["blk_magspec"]).max()
    assert np.allclose(sigproc.magspec(gold["blk_frames"], 512), gold["blk_magspec"], rtol=0, atol=1e-12 * scale)
    pscale = gold["blk_powspec"].max()
    assert np.allclose(sigproc.powspec(gold["blk_frames"], 512), gold["blk_powspec"], rtol=1e-10, atol=1e-12 * pscale)
    assert np.allclose(sigproc.powspec(gold["blk_frames"], 256), gold["blk_powspec_trunc"], rtol=1e-10, atol=1e-12 * pscale)
    with pytest.raises(IndexError):
        sigproc.preemphasis(np.zeros(0), 0.97)


def test_deframesig_and_logpowspec(gold):
    from tfkaldi_amd.processing import sigproc
    fr = gold["blk2_frames"]
    assert np.array_equal(sigproc.deframesig(fr, 1500, 400.0, 160.0), gold["blk2_deframe"])           # bit-exact
    assert np.array_equal(sigproc.deframesig(fr, 0, 400.0, 160.0), gold["blk2_deframe_full"])
    assert np.array_equal(sigproc.deframesig(fr * np.hamming(400), 1500, 400.0, 160.0, winfunc=np.hamming),
                          gold["blk2_deframe_hamming"])
    x = np.random.default_rng(8).standard_normal(5000)
    assert np.allclose(sigproc.deframesig(sigproc.framesig(x, 256, 64), 5000, 256, 64), x, rtol=0, atol=1e-12)  # round trip
    lp = sigproc.logpowspec(fr, 512)
    assert np.allclose(lp, gold["blk2_logpowspec"], rtol=0, atol=1e-9) and lp.max() == 0.0
    assert np.allclose(sigproc.logpowspec(fr, 512, norm=0), gold["blk2_logpowspec_raw"], rtol=0, atol=1e-9)
    assert np.array_equal(sigproc.logpowspec(np.zeros((3, 400)), 512, norm=0), gold["blk2_logpowspec_silence"])
    big = np.random.default_rng(9).standard_normal((3000, 400))   # more values than one reduction block
    assert np.allclose(sigproc.logpowspec(big, 512), fo.logpowspec(big, 512), rtol=0, atol=1e-9)


def test_parseval_and_linearity_at_full_length():
    """size-independent properties of the spectrum on ten minutes of audio: sum_k c_k P[k] = sum_n y[n]^2 / nfft * ...
    (Parseval for the real transform) and P(a x) = a^2 P(x)"""
    from tfkaldi_amd.processing import sigproc
    rng = np.random.default_rng(5)
    frames = rng.standard_normal((60000, 400))
    p = sigproc.powspec(frames, 512)
    weights = np.full(257, 2.0)
    weights[0] = weights[256] = 1.0
    assert np.allclose((p * weights).sum(1), (frames ** 2).sum(1), rtol=1e-11)
    assert np.allclose(sigproc.powspec(4.0 * frames[:1000], 512), 16.0 * p[:1000], rtol=1e-14, atol=0)


def test_base_functions_return_the_reference_pairs(gold):
    from tfkaldi_amd.processing import base
    sig = gold["case_aurora_fbank40_signal"]
    conf = dict(AURORA_GMM, nfilt='26')
    for name in ("fbank", "logfbank", "mfcc", "ssc"):
        feat_d, energy_d = getattr(base, name)(sig, 16000, conf)
        feat_o, energy_o = getattr(fo, name)(sig, 16000, conf)
        close64(feat_d, feat_o, name)
        close64(energy_d, energy_o, name + " energy")


def test_deltas_are_bit_exact(gold):
    from tfkaldi_amd import features
    from tfkaldi_amd.processing import base
    m = gold["blk_matrix"]
    assert np.array_equal(base.deriv(m), gold["blk_deriv"])
    assert np.array_equal(base.delta(m), gold["blk_delta"])
    assert np.array_equal(base.ddelta(m), gold["blk_ddelta"])
    for n in (1, 2, 3, 4, 5):
        assert np.array_equal(base.deriv(m[:n]), gold["blk_deriv_n%d" % n])
    from scipy.ndimage import convolve1d
    m32 = m.astype(np.float32)                       # float32 features: scipy answers in float32, stage by stage
    d32 = convolve1d(m32, [2, 1, 0, -1, -2], 0)
    assert base.deriv(m32).dtype == np.float32 and np.array_equal(base.deriv(m32), d32)
    assert np.array_equal(base.ddelta(m32), np.concatenate((m32, d32, convolve1d(d32, [2, 1, 0, -1, -2], 0)), 1))
    rng = np.random.default_rng(2)
    mats = [rng.standard_normal((n, 7)) for n in (1, 2, 3, 50, 1, 977, 4)]
    for got, x in zip(features.dynamic(mats, 2), mats):
        assert np.array_equal(got, fo.ddelta(x))  # reflection never crosses an utterance boundary


def golden_datadir(tmp, gold):
    d = os.path.join(tmp, "data")
    os.makedirs(d)
    utts = [str(u) for u in gold["prep_utts"]]
    for u in utts:
        shutil.copy(os.path.join(GOLD, "feat_data_%s.wav.bin" % u), os.path.join(d, u + ".wav"))
    open(os.path.join(d, "wav.scp"), "w").write("".join("%s %s\n" % (u, os.path.join(d, u + ".wav")) for u in utts))
    open(os.path.join(d, "utt2spk"), "w").write("".join("%s %s\n" % (u, u[:4]) for u in utts))
    open(os.path.join(d, "spk2utt"), "w").write("spkA spkA_u1 spkA_u2\nspkB spkB_u1 spkB_u2 spkB_u3\n")
    open(os.path.join(d, "text"), "w").write("".join("%s HELLO WORLD\n" % u for u in utts))
    return d, utts


def golden_featdir(tmp):
    g = os.path.join(tmp, "golden_feats")
    os.makedirs(g)
    for n in ("feats.ark", "cmvn.ark"):
        shutil.copy(os.path.join(GOLD, "feat_prep_%s.bin" % n), os.path.join(g, n))
    for n in ("feats.scp", "cmvn.scp"):
        open(os.path.join(g, n), "w").write(open(os.path.join(GOLD, "feat_prep_%s.txt" % n)).read().replace("@DIR@", g))
    return g


def test_prepare_data_and_cmvn_match_the_reference_files(tmp_path, gold, monkeypatch):
    from tfkaldi_amd.processing import ark, prepare_data
    tmp = str(tmp_path)
    d, utts = golden_datadir(tmp, gold)
    g = golden_featdir(tmp)
    for batch in (1 << 26, 6000):  # one device pass for everything / a pass every utterance or two
        monkeypatch.setattr(prepare_data, "BATCH_SAMPLES", batch)
        f = os.path.join(tmp, "feats%d" % batch)
        buf = io.StringIO()
        with redirect_stdout(buf):
            prepare_data.prepare_data(d, f, AURORA_DNN, "fbank", "nodelta")
            prepare_data.compute_cmvn(f)
        assert buf.getvalue() == str(gold["prep_stdout"])
        assert sorted(os.listdir(f)) == list(gold["prep_copied"])
        assert open(os.path.join(f, "maxlength")).read() == str(gold["prep_maxlength"])
        assert open(os.path.join(f, "feats.scp")).read().replace(f, "@DIR@") == \
            open(os.path.join(GOLD, "feat_prep_feats.scp.txt")).read()  # same order, same byte offsets
        assert open(os.path.join(f, "cmvn.scp")).read().replace(f, "@DIR@") == open(os.path.join(GOLD, "feat_prep_cmvn.scp.txt")).read()
        mine, ref = ark.ArkReader(os.path.join(f, "feats.scp")), ark.ArkReader(os.path.join(g, "feats.scp"))
        for u in utts:
            same32(mine.read_utt(u), ref.read_utt(u), u)
        cm, cr = ark.ArkReader(os.path.join(f, "cmvn.scp")), ark.ArkReader(os.path.join(g, "cmvn.scp"))
        for s in ("spkA", "spkB"):
            assert np.allclose(cm.read_utt(s), cr.read_utt(s), rtol=1e-6)
    # the CMVN sums over the REFERENCE's feature file are the reference's sums, bit for bit
    shutil.copy(os.path.join(d, "spk2utt"), os.path.join(g, "spk2utt"))
    os.remove(os.path.join(g, "cmvn.ark")); os.remove(os.path.join(g, "cmvn.scp"))
    monkeypatch.setattr(prepare_data, "BATCH_CMVN_BYTES", 1)  # one speaker per device pass
    prepare_data.compute_cmvn(g)
    assert open(os.path.join(g, "cmvn.ark"), "rb").read() == open(os.path.join(GOLD, "feat_prep_cmvn.ark.bin"), "rb").read()


def test_prepare_data_with_segments(tmp_path, gold):
    """the branch the reference cannot run (prepare_data.py:61 swaps write_next_utt's arguments): one utterance per
    segment, named by the segment, features of the sample slice [int(begin*rate), int(end*rate))"""
    from tfkaldi_amd.processing import ark, prepare_data
    import scipy.io.wavfile as wav
    tmp = str(tmp_path)
    d, utts = golden_datadir(tmp, gold)
    segs = [("segA", "spkA_u1", 0.0, 0.2), ("segB", "spkB_u1", 0.1, 0.4321), ("segC", "spkA_u1", 0.2, 0.3)]
    open(os.path.join(d, "segments"), "w").write("".join("%s %s %s %s\n" % s for s in segs))
    f = os.path.join(tmp, "feats")
    conf = dict(AURORA_GMM, include_energy='True')
    prepare_data.prepare_data(d, f, conf, "mfcc", "delta")
    reader = ark.ArkReader(os.path.join(f, "feats.scp"))
    assert reader.utt_ids == ["segA", "segC", "segB"]  # wav.scp order, then segment order within a recording
    longest = 0
    for name, rec, begin, end in segs:
        rate, samples = wav.read(os.path.join(d, rec + ".wav"))
        ref = fo.compute_features(samples[int(begin * rate):int(end * rate)], rate, "mfcc", "delta", conf)
        same32(reader.read_utt(name), ref, name)
        longest = max(longest, ref.shape[0])
    assert open(os.path.join(f, "maxlength")).read() == str(longest)


def test_cmvn_stats_large_speakers_bit_exact():
    from tfkaldi_amd import features
    rng = np.random.default_rng(9)
    speakers = [[(rng.standard_normal((int(n), 40)) * 3 + 5).astype(np.float32) for n in rng.integers(1, 900, size=k)]
                for k in (1, 7, 30, 2)]
    got = features.cmvn_stats(speakers)
    for spk, g in zip(speakers, got):
        assert np.array_equal(g, fo.cmvn_stats(np.concatenate(spk)))
    wide = [[rng.standard_normal((50, 130)).astype(np.float32)]]  # more columns than one wavefront
    assert np.array_equal(features.cmvn_stats(wide)[0], fo.cmvn_stats(wide[0][0]))


@pytest.mark.parametrize("dtype", [np.int32, np.float32, np.uint8, np.int64])
def test_other_wav_sample_types(dtype):
    """scipy.io.wavfile hands back int32 / float32 / uint8 arrays for such files: numpy promotes them to float64 in the
    reference's pre-emphasis, and so does the packer here (only int16 travels as it is)"""
    from tfkaldi_amd.processing import feat
    rng = np.random.default_rng(12)
    if dtype is np.uint8:
        sig = rng.integers(0, 256, size=6000).astype(dtype)
    elif dtype is np.float32:
        sig = (rng.standard_normal(6000) * 0.2).astype(dtype)
    else:
        sig = rng.integers(-2 ** 20, 2 ** 20, size=6000).astype(dtype)
    comp = feat.FeatureComputer("mfcc", "delta", AURORA_GMM)
    with np.errstate(all="ignore"):
        ref = fo.compute_features(sig, 16000, "mfcc", "delta", AURORA_GMM)
    close64(comp(sig, 16000), ref, str(dtype))


def test_plan_errors_are_loud():
    from tfkaldi_amd._lib import EngineError
    from tfkaldi_amd.processing import feat
    sig = np.zeros(4000, dtype=np.int16)
    with pytest.raises(ValueError, match="power of two"):  # (host-side, at construction: tests/test_feat_oracle.py)
        feat.FeatureComputer("fbank", "nodelta", dict(AURORA_DNN, nfft='400'))
    from tfkaldi_amd import features
    with pytest.raises(EngineError, match="power of two"):  # the C ABI itself refuses as well
        features.FeaturePlan("fbank", "nodelta", 400, 160, 400, 40, np.zeros((40, 201)))
    with pytest.raises(ValueError, match="one-dimensional"):
        feat.FeatureComputer("fbank", "nodelta", AURORA_DNN)(np.zeros((4000, 2), dtype=np.int16), 16000)
