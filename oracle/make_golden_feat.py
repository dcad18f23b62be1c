"""Generate tests/golden/feat_* from the REFERENCE's own feature code (run in the build container only):
processing/{feat,base,sigproc,prepare_data}.py imported through lib2to3 in a scratch directory under /tmp
(oracle/make_golden_io.py::import_reference_processing; nothing of the reference enters this repository).
The fixtures are DATA: synthetic signals / wav files made here and the arrays / ark bytes the reference produced.

    python oracle/make_golden_feat.py        # needs /root/reference; writes tests/golden/feat_golden.npz, feat_data_*
"""
import contextlib
import io
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_io import GOLD, import_reference_processing  # noqa: E402

AURORA_DNN = dict(winlen='0.025', winstep='0.01', nfilt='40', nfft='512', lowfreq='0', highfreq='-1', preemph='0.97',
                  include_energy='False', snip_edges='True')                     # config_AURORA4.cfg:56-76
AURORA_GMM = dict(AURORA_DNN, nfilt='23', numcep='13', ceplifter='22')           # config_AURORA4.cfg:30-54


def speechlike(rng, n, rate, amp=4000.0):
    t = np.arange(n) / float(rate)
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in
            zip((1.0, 0.6, 0.3, 0.2), (140.0, 730.0, 1210.0, 2950.0), rng.uniform(0, 6.28, 4)))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + 0.15 * rng.standard_normal(n)
    return np.clip(np.round(x * amp / 2.1), -32768, 32767).astype(np.int16)


def cases(rng):
    """name -> (signal, rate, feature type, dynamic, conf)"""
    c = {}
    c["aurora_fbank40"] = (speechlike(rng, 8000, 16000), 16000, "fbank", "nodelta", AURORA_DNN)
    c["aurora_mfcc13"] = (speechlike(rng, 6100, 16000), 16000, "mfcc", "nodelta", AURORA_GMM)
    c["mfcc_ddelta_energy_pad"] = (speechlike(rng, 5923, 16000), 16000, "mfcc", "ddelta",
                                   dict(AURORA_GMM, include_energy='True', snip_edges='False'))
    c["ssc_delta_8k"] = (speechlike(rng, 3000, 8000), 8000, "ssc", "delta",
                         dict(AURORA_DNN, nfilt='15', nfft='256', lowfreq='100', highfreq='3800'))
    c["fbank_delta_energy"] = (speechlike(rng, 4321, 16000), 16000, "fbank", "delta", dict(AURORA_DNN, include_energy='True'))
    c["short_one_frame_pad"] = (speechlike(rng, 300, 16000), 16000, "fbank", "ddelta", dict(AURORA_DNN, snip_edges='False'))
    c["short_one_frame_snip"] = (speechlike(rng, 300, 16000), 16000, "fbank", "delta", AURORA_DNN)
    c["exactly_one_frame"] = (speechlike(rng, 400, 16000), 16000, "mfcc", "ddelta", AURORA_GMM)
    c["two_frames"] = (speechlike(rng, 560, 16000), 16000, "fbank", "ddelta", AURORA_DNN)
    c["three_frames"] = (speechlike(rng, 725, 16000), 16000, "mfcc", "ddelta", AURORA_GMM)
    c["four_frames_pad"] = (speechlike(rng, 801, 16000), 16000, "fbank", "ddelta", dict(AURORA_DNN, snip_edges='False'))
    c["silence_fbank"] = (np.zeros(2000, dtype=np.int16), 16000, "fbank", "delta", dict(AURORA_DNN, include_energy='True'))
    c["silence_ssc"] = (np.zeros(1200, dtype=np.int16), 16000, "ssc", "nodelta", AURORA_DNN)
    c["float_signal_truncating_fft"] = (rng.standard_normal(5000) * 0.3, 16000, "fbank", "nodelta",
                                        dict(AURORA_DNN, winlen='0.02', winstep='0.0125', nfft='256', preemph='0',
                                             nfilt='20', snip_edges='False'))
    c["wide_fft_1024"] = (speechlike(rng, 5000, 16000), 16000, "ssc", "ddelta",
                          dict(AURORA_DNN, nfft='1024', nfilt='64', lowfreq='64', highfreq='7600'))
    c["many_filters_mfcc"] = (speechlike(rng, 4000, 16000), 16000, "mfcc", "delta",
                              dict(AURORA_GMM, nfilt='80', numcep='20', ceplifter='0', include_energy='True'))
    return c


def main():
    os.makedirs(GOLD, exist_ok=True)
    scratch = import_reference_processing()[0]
    import processing.base as base
    import processing.feat as feat
    import processing.prepare_data as prep
    import processing.sigproc as sigproc
    import scipy.io.wavfile as wav
    rng = np.random.default_rng(777)
    out = {}

    # ---- FeatureComputer.__call__ (feat.py:42-69) ----
    names = []
    for name, (sig, rate, ftype, dyn, conf) in cases(rng).items():
        with np.errstate(all="ignore"):
            got = feat.FeatureComputer(ftype, dyn, conf)(sig, rate)
        out["case_%s_signal" % name] = sig
        out["case_%s_features" % name] = got
        out["case_%s_meta" % name] = np.array(json.dumps({"rate": rate, "type": ftype, "dynamic": dyn, "conf": conf}))
        names.append(name)
    out["case_names"] = np.array(names)

    # ---- the building blocks (sigproc.py, base.py) on small inputs ----
    x = speechlike(rng, 1000, 16000)
    out["blk_signal"] = x
    out["blk_preemph"] = sigproc.preemphasis(x, 0.97)
    fr = sigproc.framesig(out["blk_preemph"], 400.0, 160.0)
    out["blk_frames"] = fr
    out["blk_frames_odd"] = sigproc.framesig(x[:333].astype(np.float64), 100.4, 33.6)  # round() -> 100, 34
    out["blk_magspec"] = sigproc.magspec(fr, 512)
    out["blk_powspec"] = sigproc.powspec(fr, 512)
    out["blk_powspec_trunc"] = sigproc.powspec(fr, 256)
    out["blk_fb_40_512_16k"] = base.get_filterbanks(40, 512, 16000, 0, 8000)
    out["blk_fb_23_512_16k"] = base.get_filterbanks(23, 512, 16000, 0, 8000)
    out["blk_fb_15_256_8k"] = base.get_filterbanks(15, 256, 8000, 100, 3800)
    out["blk_fb_default"] = base.get_filterbanks()
    out["blk_hz"] = np.array([0.0, 100.0, 1000.0, 3800.0, 8000.0])
    out["blk_hz2mel"] = base.hz2mel(out["blk_hz"])
    out["blk_mel2hz"] = base.mel2hz(out["blk_hz2mel"])
    cep = rng.standard_normal((5, 13))
    out["blk_cepstra"] = cep
    out["blk_lifter22"] = base.lifter(cep, 22.0)
    out["blk_lifter0"] = base.lifter(cep, 0.0)
    m = rng.standard_normal((9, 4))
    out["blk_matrix"] = m
    out["blk_deriv"] = base.deriv(m)
    out["blk_delta"] = base.delta(m)
    out["blk_ddelta"] = base.ddelta(m)
    for n in (1, 2, 3, 4, 5):
        out["blk_deriv_n%d" % n] = base.deriv(m[:n])
    out["blk_snip"] = np.array([len(feat.snip(np.zeros(n), 16000, 0.025, 0.01)) for n in (300, 400, 559, 560, 561, 16000)])

    # ---- prepare_data + compute_cmvn + shuffle_examples on a small kaldi data directory (prepare_data.py:13-139) ----
    d = os.path.join(scratch, "data")
    f = os.path.join(scratch, "feats")
    os.makedirs(d)
    utts = [("spkA_u1", "spkA", 5000), ("spkA_u2", "spkA", 3456), ("spkB_u1", "spkB", 7777), ("spkB_u2", "spkB", 2100),
            ("spkB_u3", "spkB", 4000)]
    for uid, _, n in utts:
        wav.write(os.path.join(d, uid + ".wav"), 16000, speechlike(rng, n, 16000))
        shutil.copy(os.path.join(d, uid + ".wav"), os.path.join(GOLD, "feat_data_%s.wav.bin" % uid))
    open(os.path.join(d, "wav.scp"), "w").write("".join("%s %s\n" % (u, os.path.join(d, u + ".wav")) for u, _, _ in utts))
    open(os.path.join(d, "utt2spk"), "w").write("".join("%s %s\n" % (u, s) for u, s, _ in utts))
    open(os.path.join(d, "spk2utt"), "w").write("spkA spkA_u1 spkA_u2\nspkB spkB_u1 spkB_u2 spkB_u3\n")
    open(os.path.join(d, "text"), "w").write("".join("%s HELLO WORLD\n" % u for u, _, _ in utts))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        prep.prepare_data(d, f, AURORA_DNN, "fbank", "nodelta")
        prep.compute_cmvn(f)
    out["prep_stdout"] = np.array(buf.getvalue())
    for name in ("feats.ark", "cmvn.ark"):
        shutil.copy(os.path.join(f, name), os.path.join(GOLD, "feat_prep_%s.bin" % name))
    for name in ("feats.scp", "cmvn.scp"):
        open(os.path.join(GOLD, "feat_prep_%s.txt" % name), "w").write(open(os.path.join(f, name)).read().replace(f, "@DIR@"))
    out["prep_maxlength"] = np.array(open(os.path.join(f, "maxlength")).read())
    out["prep_copied"] = np.array(sorted(os.listdir(f)))
    out["prep_utts"] = np.array([u for u, _, _ in utts])
    import random
    random.seed(5)
    prep.shuffle_examples(f)
    out["prep_shuffled_sorted"] = np.array(sorted(open(os.path.join(f, "feats_shuffled.scp")).read().replace(f, "@DIR@").split("\n")))

    # ---- the two sigproc functions nothing in the reference calls (sigproc.py:69-123, 155-178) ----
    rng2 = np.random.default_rng(4242)
    fr2 = sigproc.framesig(speechlike(rng2, 1500, 16000).astype(np.float64), 400.0, 160.0)
    out["blk2_frames"] = fr2
    out["blk2_deframe"] = sigproc.deframesig(fr2, 1500, 400.0, 160.0)
    out["blk2_deframe_full"] = sigproc.deframesig(fr2, 0, 400.0, 160.0)
    out["blk2_deframe_hamming"] = sigproc.deframesig(fr2 * np.hamming(400), 1500, 400.0, 160.0, winfunc=np.hamming)
    out["blk2_logpowspec"] = sigproc.logpowspec(fr2, 512)
    out["blk2_logpowspec_raw"] = sigproc.logpowspec(fr2, 512, norm=0)
    out["blk2_logpowspec_silence"] = sigproc.logpowspec(np.zeros((3, 400)), 512, norm=0)

    np.savez_compressed(os.path.join(GOLD, "feat_golden.npz"), **out)
    shutil.rmtree(scratch)
    print("wrote %d arrays to %s" % (len(out), os.path.abspath(GOLD)))


if __name__ == "__main__":
    main()
