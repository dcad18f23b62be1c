"""fbank / mfcc / ssc features and their deltas with the reference's interface (processing/base.py; the reference's
module derives from python_speech_features), computed on the GPU (csrc/features.hip; no CPU path).  What stays on the
host is table construction: mel scale, filterbank, DCT matrix and lifter weights (built once per plan)."""
import math

import numpy

from .. import features


def hz2mel(rate):
    """base.py:156-167"""
    return 2595 * numpy.log10(1 + rate / 700.0)


def mel2hz(mel):
    """base.py:169-180"""
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """base.py:182-224: [nfilt, nfft/2 + 1] triangular mel filters whose edges are floor()-ed FFT bins
    (the reference runs under Python 2: `nfft/2` and `samplerate/2` floor)"""
    highfreq = highfreq or samplerate // 2
    assert highfreq <= samplerate // 2, "highfreq is greater than samplerate/2"
    edges = numpy.floor((nfft + 1) * mel2hz(numpy.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)) / samplerate)
    fbanks = numpy.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        left, centre, right = edges[j], edges[j + 1], edges[j + 2]
        rising = numpy.arange(int(left), int(centre))
        falling = numpy.arange(int(centre), int(right))
        fbanks[j, rising] = (rising - left) / (centre - left)
        fbanks[j, falling] = (right - falling) / (right - centre)
    return fbanks


def dct_matrix(nfilt, numcep):
    """the matrix of scipy.fftpack.dct(type=2, norm='ortho') along a row of nfilt values, first numcep outputs
    (base.py:55): [nfilt, numcep]"""
    n = numpy.arange(nfilt)[:, None]
    k = numpy.arange(numcep)[None, :]
    scale = numpy.where(k == 0, math.sqrt(1.0 / (4 * nfilt)), math.sqrt(1.0 / (2 * nfilt)))
    return 2.0 * numpy.cos(numpy.pi * k * (2 * n + 1) / (2.0 * nfilt)) * scale


def lifter_weights(ncoeff, liftering):
    """base.py:240-246: 1 + (L/2) sin(pi n / L); L <= 0 disables the lifter"""
    if liftering > 0:
        return 1 + (liftering / 2) * numpy.sin(numpy.pi * numpy.arange(ncoeff) / liftering)
    return numpy.ones(ncoeff)


def make_plan(kind, dynamic, samplerate, conf, include_energy):
    """the device plan of one feature configuration at one sample rate (base.py:74-90 for the geometry)"""
    highfreq = int(conf['highfreq'])
    if highfreq < 0:
        highfreq = samplerate // 2
    nfft, nfilt = int(conf['nfft']), int(conf['nfilt'])
    frame_len = features.py2_round(float(conf['winlen']) * samplerate)
    frame_step = features.py2_round(float(conf['winstep']) * samplerate)
    fb = get_filterbanks(nfilt, nfft, samplerate, int(conf['lowfreq']), highfreq)
    extra = {}
    numcep = 0
    if kind == "mfcc":
        numcep = min(int(conf['numcep']), nfilt)  # `dct(...)[:, :numcep]` cannot give more than nfilt columns
        extra = dict(dct=dct_matrix(nfilt, numcep), lifter=lifter_weights(numcep, float(conf['ceplifter'])))
    elif kind == "ssc":
        extra = dict(bin_weight=numpy.linspace(1, samplerate // 2, nfft // 2 + 1))  # base.py:151
    return features.FeaturePlan(kind, dynamic, frame_len, frame_step, nfft, nfilt, fb, numcep=numcep,
                                include_energy=include_energy, preemph=float(conf['preemph']), **extra)


def _static(kind, signal, samplerate, conf):
    plan = make_plan(kind, "nodelta", samplerate, conf, include_energy=True)
    out = plan.compute([numpy.asarray(signal)])[0]
    plan.close()
    return numpy.ascontiguousarray(out[:, :-1]), numpy.ascontiguousarray(out[:, -1])


def mfcc(signal, samplerate, conf):
    """base.py:39-57 -> ([NUMFRAMES, numcep] liftered cepstra, log frame energy)"""
    return _static("mfcc", signal, samplerate, conf)


def fbank(signal, samplerate, conf):
    """base.py:59-98 -> ([NUMFRAMES, nfilt] filterbank energies, frame energy)"""
    return _static("fbank_raw", signal, samplerate, conf)


def logfbank(signal, samplerate, conf):
    """base.py:100-114 -> (log filterbank energies, log frame energy)"""
    return _static("fbank", signal, samplerate, conf)


def ssc(signal, samplerate, conf):
    """base.py:116-154 -> (spectral subband centroids, log frame energy)"""
    return _static("ssc", signal, samplerate, conf)


def lifter(cepstra, liftering=22):
    """base.py:226-246"""
    if liftering > 0:
        return lifter_weights(numpy.shape(cepstra)[1], liftering) * cepstra
    return cepstra


def _same_kind(result, like):
    """scipy.ndimage.convolve1d returns its input's dtype: float32 features give float32 derivatives (computed in double,
    rounded on the way out); everything else is float64 here as it is in the feature pipeline"""
    return result.astype(numpy.float32) if numpy.asarray(like).dtype == numpy.float32 else result


def deriv(features_):
    """base.py:248-258: first-order derivative over time (kernel [2, 1, 0, -1, -2], reflected edges)"""
    return _same_kind(features.dynamic([features_], 1, deriv_only=True)[0], features_)


def delta(features_):
    """base.py:260-270: [features | derivative]"""
    return _same_kind(features.dynamic([features_], 1)[0], features_)


def ddelta(features_):
    """base.py:272-284: [features | derivative | second derivative]"""
    if numpy.asarray(features_).dtype == numpy.float32:  # the second derivative is taken of the ROUNDED first one there
        first = deriv(features_)
        return numpy.concatenate((features_, first, deriv(first)), 1)
    return features.dynamic([features_], 2)[0]
