"""FeatureComputer with the reference's interface (processing/feat.py), running on the GPU."""
import numpy as np

from . import base


class FeatureComputer(object):
    """computes one type of features (feat.py:7-69).  `__call__` is the reference's per-utterance entry point;
    `compute_batch` does a whole list of utterances in one device pass."""

    def __init__(self, featureType, dynamic, conf):
        if featureType not in ('fbank', 'mfcc', 'ssc'):
            raise Exception('unknown feature type')
        if dynamic not in ('nodelta', 'delta', 'ddelta'):
            raise Exception('unknown dynamic type')
        self.feature_type, self.dynamic, self.conf = featureType, dynamic, conf
        self._plans = {}
        # What the device transform supports is checked HERE, on the host, before any file is touched (the device plan is
        # built at the first batch, per sample rate): numpy's rfft takes any length, csrc/features.hip a power of two in
        # [32, 4096] with nfilt <= nfft / 2, and there is no CPU path to fall back to.
        nfft, nfilt = int(conf.get('nfft', 512)), int(conf.get('nfilt', 1))  # (missing keys: KeyError at first use, as the reference)
        if nfft < 32 or nfft > 4096 or nfft & (nfft - 1):
            raise ValueError("nfft = %d: the GPU feature computation needs a power of two in [32, 4096] (the reference's "
                             "configurations use 512)" % nfft)
        if nfilt < 1 or nfilt > nfft // 2:
            raise ValueError("nfilt = %d must lie in [1, nfft / 2 = %d]" % (nfilt, nfft // 2))

    def plan(self, rate):
        if rate not in self._plans:
            self._plans[rate] = base.make_plan(self.feature_type, self.dynamic, rate, self.conf,
                                               include_energy=self.conf['include_energy'] == 'True')
        return self._plans[rate]

    def _prepared(self, sig, rate):
        if self.conf['snip_edges'] == 'True':
            sig = snip(sig, rate, float(self.conf['winlen']), float(self.conf['winstep']))
        if len(sig) == 0:  # the reference's pre-emphasis reads `signal[0]` (sigproc.py:191)
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        return sig

    def __call__(self, sig, rate):
        """feat.py:42-69: [NUMFRAMES, dim] float64"""
        sig = self._prepared(np.asarray(sig), rate)
        return self.plan(rate).compute([sig])[0]

    def compute_batch(self, sigs, rate, dtype=np.float32):
        """features of many utterances recorded at one sample rate; float32 is what the ark files store"""
        sigs = [self._prepared(np.asarray(s), rate) for s in sigs]
        return self.plan(rate).compute(sigs, dtype=dtype)


def snip(sig, rate, winlen, winstep):
    """feat.py:71-90: cut the tail that does not fill a whole window step"""
    num_frames = int((len(sig) - winlen * rate) / (winstep * rate))
    return sig[0:int(num_frames * winstep * rate + winlen * rate)]
