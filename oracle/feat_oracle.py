"""CPU oracle of the reference's FEATURE COMPUTATION (wav samples -> fbank / mfcc / ssc (+ deltas) -> per-speaker
CMVN statistics): SURVEY.md section 8(f), the step in front of the feature reader.

TEST INFRASTRUCTURE ONLY: imported by tests/, by the generator of the golden vectors and by the cpu_baseline leg of
tools/feature_bench.py -- never by tfkaldi_amd/.  It restates, in float64 numpy exactly as the reference computes,

    processing/feat.py:7-90          FeatureComputer.__call__, snip
    processing/base.py:39-284        mfcc, fbank, logfbank, ssc, hz2mel, mel2hz, get_filterbanks, lifter, deriv, delta, ddelta
    processing/sigproc.py:33-191     framesig, deframesig, magspec, powspec, logpowspec, preemphasis
    processing/prepare_data.py:80-118 compute_cmvn (float32 running sums, see cmvn_stats)

Parity is PINNED: tests/golden/feat_golden.npz holds outputs of the reference's own code for the same inputs
(oracle/make_golden_feat.py imports /root/reference/processing through lib2to3 in a scratch directory); this file is
checked against every one of them in tests/test_feat_oracle.py.

Python-2 semantics the reference relies on and that are kept here: `/` between ints floors (`nfft/2+1`,
`samplerate/2`: base.py:76,205,217), `round()` rounds halves away from zero (sigproc.py:50-51).
"""
import math

import numpy as np

EPS = np.finfo(float).eps


def py2_round(x):
    """round() of Python 2: halves away from zero (sigproc.py:50-51 uses it on frame_len / frame_step)"""
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def preemphasis(signal, coeff=0.95):
    """sigproc.py:180-191: y[0] = x[0], y[n] = x[n] - coeff * x[n-1] (float64 as soon as coeff is a float)"""
    signal = np.asarray(signal)
    head = signal[0]  # an empty signal raises IndexError here, as in the reference
    return np.concatenate(([head], signal[1:] - coeff * signal[:-1])).astype(np.float64, copy=False)


def num_frames(slen, frame_len, frame_step):
    """sigproc.py:49-55"""
    if slen <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))


def framesig(sig, frame_len, frame_step):
    """sigproc.py:33-67 with the default (rectangular) window -- the only one the reference ever passes: the signal is
    zero-padded to the end of the last frame"""
    frame_len, frame_step = py2_round(frame_len), py2_round(frame_step)
    n = num_frames(len(sig), frame_len, frame_step)
    padded = np.zeros((n - 1) * frame_step + frame_len)
    padded[:len(sig)] = sig
    idx = np.arange(frame_len)[None, :] + (np.arange(n) * frame_step)[:, None]
    return padded[idx]


def magspec(frames, nfft):
    """sigproc.py:125-139: |rfft| (frames longer than nfft are truncated, shorter ones zero-padded, by numpy)"""
    return np.absolute(np.fft.rfft(frames, nfft))


def powspec(frames, nfft):
    """sigproc.py:141-153"""
    return 1.0 / nfft * np.square(magspec(frames, nfft))


def deframesig(frames, siglen, frame_len, frame_step, winfunc=lambda n: np.ones((n,))):
    """sigproc.py:69-123: overlap-add; the frames and the window (+ 1e-15) are accumulated frame after frame"""
    frame_len, frame_step = py2_round(frame_len), py2_round(frame_step)
    n = frames.shape[0]
    assert frames.shape[1] == frame_len
    padlen = (n - 1) * frame_step + frame_len
    if siglen <= 0:
        siglen = padlen
    rec, corr, win = np.zeros(padlen), np.zeros(padlen), winfunc(frame_len)
    for i in range(n):
        seg = slice(i * frame_step, i * frame_step + frame_len)
        corr[seg] = corr[seg] + win + 1e-15
        rec[seg] = rec[seg] + frames[i]
    return (rec / corr)[0:siglen]


def logpowspec(frames, nfft, norm=1):
    """sigproc.py:155-178"""
    ps = powspec(frames, nfft)
    ps[ps <= 1e-30] = 1e-30
    lps = 10 * np.log10(ps)
    return lps - np.max(lps) if norm else lps


def hz2mel(rate):
    return 2595 * np.log10(1 + rate / 700.0)  # base.py:156-167


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)  # base.py:169-180


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """base.py:182-224: triangular filters on floor()-ed FFT-bin edges, [nfilt, nfft//2 + 1]"""
    highfreq = highfreq or samplerate // 2
    assert highfreq <= samplerate // 2, "highfreq is greater than samplerate/2"
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def _spectrum(signal, samplerate, conf):
    """the shared head of fbank / ssc (base.py:74-90, 132-146)"""
    highfreq = int(conf['highfreq'])
    if highfreq < 0:
        highfreq = samplerate // 2
    signal = preemphasis(signal, float(conf['preemph']))
    frames = framesig(signal, float(conf['winlen']) * samplerate, float(conf['winstep']) * samplerate)
    pspec = powspec(frames, int(conf['nfft']))
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, EPS, energy)
    fb = get_filterbanks(int(conf['nfilt']), int(conf['nfft']), samplerate, int(conf['lowfreq']), highfreq)
    return pspec, energy, fb


def fbank(signal, samplerate, conf):
    """base.py:59-98"""
    pspec, energy, fb = _spectrum(signal, samplerate, conf)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, EPS, feat)
    return feat, energy


def logfbank(signal, samplerate, conf):
    feat, energy = fbank(signal, samplerate, conf)  # base.py:100-114
    return np.log(feat), np.log(energy)


def dct_matrix(nfilt, numcep):
    """scipy.fftpack.dct(type=2, norm='ortho') along axis 1, first numcep outputs, as a [nfilt, numcep] matrix"""
    n = np.arange(nfilt)[:, None]
    k = np.arange(numcep)[None, :]
    m = 2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * nfilt))
    scale = np.where(k == 0, math.sqrt(1.0 / (4 * nfilt)), math.sqrt(1.0 / (2 * nfilt)))
    return m * scale


def lifter_weights(ncoeff, liftering):
    """base.py:226-246 (liftering arrives as a float: `liftering/2` is a true division)"""
    if liftering > 0:
        return 1 + (liftering / 2) * np.sin(np.pi * np.arange(ncoeff) / liftering)
    return np.ones(ncoeff)


def mfcc(signal, samplerate, conf):
    """base.py:39-57"""
    feat, energy = fbank(signal, samplerate, conf)
    feat = np.log(feat)
    numcep = int(conf['numcep'])
    feat = np.dot(feat, dct_matrix(feat.shape[1], min(numcep, feat.shape[1])))
    feat = lifter_weights(feat.shape[1], float(conf['ceplifter'])) * feat
    return feat, np.log(energy)


def ssc(signal, samplerate, conf):
    """base.py:116-154 (the denominator is NOT guarded against zero there)"""
    pspec, energy, fb = _spectrum(signal, samplerate, conf)
    feat = np.dot(pspec, fb.T)
    tiles = np.linspace(1, samplerate // 2, pspec.shape[1])[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.dot(pspec * tiles, fb.T) / feat, np.log(energy)


def _reflect(i, n):
    """index of scipy.ndimage's default boundary mode 'reflect' (d c b a | a b c d | d c b a)"""
    i = i % (2 * n)
    return 2 * n - 1 - i if i >= n else i


def deriv(features):
    """base.py:248-258: scipy.ndimage.convolve1d(features, [2, 1, 0, -1, -2], axis 0), boundary mode 'reflect':
    d[t] = 2 x[t+2] + x[t+1] - x[t-1] - 2 x[t-2], accumulated as scipy's anti-symmetric kernel path does"""
    x = np.asarray(features, dtype=np.float64)
    n = x.shape[0]
    out = np.empty_like(x)
    for t in range(n):
        acc = 0.0 * x[t]
        acc = acc + (-2.0) * (x[_reflect(t - 2, n)] - x[_reflect(t + 2, n)])
        acc = acc + (-1.0) * (x[_reflect(t - 1, n)] - x[_reflect(t + 1, n)])
        out[t] = acc
    return out


def delta(features):
    return np.concatenate((features, deriv(features)), 1)  # base.py:260-270


def ddelta(features):
    d = deriv(features)  # base.py:272-284
    return np.concatenate((features, d, deriv(d)), 1)


def snip(sig, rate, winlen, winstep):
    """feat.py:71-90"""
    n = int((len(sig) - winlen * rate) / (winstep * rate))
    return sig[0:int(n * winstep * rate + winlen * rate)]


def compute_features(sig, rate, feat_type, dynamic, conf):
    """feat.py:42-69 (FeatureComputer.__call__)"""
    comp = {"fbank": logfbank, "mfcc": mfcc, "ssc": ssc}
    if feat_type not in comp:
        raise Exception('unknown feature type')
    dyn = {"nodelta": lambda x: x, "delta": delta, "ddelta": ddelta}
    if dynamic not in dyn:
        raise Exception('unknown dynamic type')
    if conf['snip_edges'] == 'True':
        sig = snip(sig, rate, float(conf['winlen']), float(conf['winstep']))
    feat, energy = comp[feat_type](sig, rate, conf)
    if conf['include_energy'] == 'True':
        feat = np.append(feat, energy[:, np.newaxis], 1)
    return dyn[dynamic](feat)


def cmvn_stats(spk_data):
    """prepare_data.py:103-111 on the speaker's stacked utterances as the ark reader returns them (float32): numpy
    reduces axis 0 of a C-contiguous float32 matrix row after row IN FLOAT32, and squares in float32; the [2, D+1]
    result is float64 only as a container (row 0: sums | frame count, row 1: sums of squares | 0)."""
    x = np.ascontiguousarray(spk_data, dtype=np.float32)
    dim = x.shape[1]
    s1 = np.zeros(dim, dtype=np.float32)
    s2 = np.zeros(dim, dtype=np.float32)
    for row in x:
        s1 = s1 + row
        s2 = s2 + row * row
    stats = np.zeros([2, dim + 1])
    stats[0, :dim] = s1
    stats[1, :dim] = s2
    stats[0, dim] = x.shape[0]
    return stats
