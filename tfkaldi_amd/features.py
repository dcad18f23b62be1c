"""Device plans for the feature computation (include/tfkaldi_hip.h: tfk_feat_*, tfk_cmvn_stats): the layer between the
reference-named modules processing/{sigproc,base,feat,prepare_data}.py and the HIP kernels of csrc/features.hip.

A plan fixes the frame geometry, the transform length and the tables (mel filterbank, DCT, lifter); `compute` takes a
whole BATCH of utterance signals -- concatenated, uploaded once -- and returns one matrix per utterance.  torch holds
the device buffers and the stream; all arithmetic happens in the library.  There is no CPU path."""
import ctypes
import math
from ctypes import byref, c_int32, c_void_p

import numpy as np

from . import _lib
from ._lib import check


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("feature computation runs on the GPU (csrc/features.hip); no HIP device is visible and "
                           "there is no CPU fallback")
    return torch


def py2_round(x):
    """Python 2's round(): halves away from zero -- sigproc.py:50-51 rounds frame_len / frame_step with it"""
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def count_frames(slen, frame_len, frame_step):
    """sigproc.py:49-55"""
    if slen <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))


def _dptr(t):
    return c_void_p(t.data_ptr())


def sample_class(sig):
    """how numpy treats a wav array in `signal[1:] - coeff * signal[:-1]` (sigproc.py:191): int16 travels as it is and is
    widened on the device; float32 STAYS float32 through the pre-emphasis (the Python float is cast down); every other
    type (int32, uint8, float64, ...) is promoted to float64"""
    dt = np.asarray(sig).dtype
    return "i16" if dt == np.int16 else "f32" if dt == np.float32 else "f64"


def by_sample_class(signals):
    """indices of the signals of each class, in order"""
    groups = {}
    for i, s in enumerate(signals):
        groups.setdefault(sample_class(s), []).append(i)
    return groups


class Packed(object):
    """a batch of signals concatenated in HBM with its utterance / frame offsets"""

    def __init__(self, signals, frame_len, frame_step, device):
        torch = _torch()
        sigs = [np.asarray(s) for s in signals]
        for s in sigs:
            if s.ndim != 1:
                raise ValueError("a signal must be one-dimensional (mono), got shape %s" % (s.shape,))
        classes = set(sample_class(s) for s in sigs)
        if len(classes) > 1:
            raise ValueError("one batch holds one kind of samples (int16, float32 or float64-promoted): %s" % sorted(classes))
        kind = classes.pop() if classes else "f64"
        self.sample_type, dtype = {"i16": (_lib.SAMPLE_I16, np.int16), "f32": (_lib.SAMPLE_F32, np.float32),
                                   "f64": (_lib.SAMPLE_F64, np.float64)}[kind]
        lens = np.array([s.size for s in sigs], dtype=np.int64)
        self.sig_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        frames = np.array([count_frames(int(n), frame_len, frame_step) for n in lens], dtype=np.int64)
        self.frame_off = np.concatenate(([0], np.cumsum(frames))).astype(np.int64)
        self.n_utts = len(sigs)
        self.n_frames = int(self.frame_off[-1])
        dev = torch.device("cuda", device)
        total = max(int(self.sig_off[-1]), 1)  # (one element at least: a valid pointer for all-empty batches)
        # the samples are gathered straight into pinned memory: one pass over them on the host, then a DMA at PCIe speed
        # (a pageable array would be staged through the driver's bounce buffers at a fraction of it)
        tdt = {np.int16: torch.int16, np.float32: torch.float32, np.float64: torch.float64}[dtype]
        staging = torch.zeros(total, dtype=tdt, pin_memory=True)
        flat = staging.numpy()
        for s, lo, hi in zip(sigs, self.sig_off[:-1], self.sig_off[1:]):
            flat[lo:hi] = s
        self.signal = staging.to(dev, non_blocking=True)
        self.d_sig_off = torch.from_numpy(self.sig_off).to(dev)
        self.d_frame_off = torch.from_numpy(self.frame_off).to(dev)
        self._staging = staging  # alive until the copy has been consumed (the compute calls synchronise before returning)

    def split(self, matrix):
        return [matrix[self.frame_off[u]:self.frame_off[u + 1]] for u in range(self.n_utts)]


class FeaturePlan(object):
    """tfk_feat: one feature configuration at one sample rate"""

    def __init__(self, kind, dynamic, frame_len, frame_step, nfft, nfilt, filterbank, numcep=0, include_energy=False,
                 preemph=0.0, bin_weight=None, dct=None, lifter=None, device=0):
        self.lib = _lib.load()
        cfg = _lib.TfkFeatConfig()
        cfg.struct_size = ctypes.sizeof(_lib.TfkFeatConfig)
        cfg.device = device
        cfg.kind, cfg.dynamic = _lib.FEAT_KIND[kind], _lib.FEAT_DYNAMIC[dynamic]
        cfg.frame_len, cfg.frame_step, cfg.nfft, cfg.nfilt, cfg.numcep = int(frame_len), int(frame_step), int(nfft), int(nfilt), int(numcep)
        cfg.include_energy = int(bool(include_energy))
        cfg.preemph = float(preemph)
        self.cfg, self.device = cfg, device
        tabs = []

        def host(a, shape):
            if a is None:
                return c_void_p(None)
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != shape:
                raise ValueError("table of shape %s, expected %s" % (a.shape, shape))
            tabs.append(a)
            return a.ctypes.data_as(c_void_p)
        nbins = int(nfft) // 2 + 1
        self._h = c_void_p()
        _torch()  # binds the HIP runtime / fails loudly without a GPU before the library is asked for device memory
        check(self.lib.tfk_feat_create(byref(cfg), host(filterbank, (int(nfilt), nbins)), host(bin_weight, (nbins,)),
                                       host(dct, (int(nfilt), int(numcep))), host(lifter, (int(numcep),)), byref(self._h)))
        d = c_int32()
        check(self.lib.tfk_feat_dim(self._h, byref(d)))
        self.dim, self.nbins = d.value, nbins

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.tfk_feat_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def pack(self, signals):
        return Packed(signals, self.cfg.frame_len, self.cfg.frame_step, self.device)

    def compute_device(self, packed, dtype=np.float64):
        """[n_frames, dim] torch tensor in HBM (float32: what the ark files hold; float64: what the reference returns)"""
        torch = _torch()
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
        out = torch.empty((packed.n_frames, self.dim), dtype=tdt, device=packed.signal.device)
        stream = torch.cuda.current_stream(packed.signal.device).cuda_stream
        check(self.lib.tfk_feat_compute(self._h, c_void_p(stream), _dptr(packed.signal), packed.sample_type,
                                        _dptr(packed.d_sig_off), _dptr(packed.d_frame_off), packed.n_utts,
                                        packed.n_frames, _dptr(out), self.dim, int(tdt == torch.float64)))
        return out

    def compute(self, signals, dtype=np.float64):
        """one [frames, dim] array per signal (signals of different sample classes go in separate device passes)"""
        torch = _torch()
        signals = [np.asarray(s) for s in signals]
        result = [None] * len(signals)
        for _, idx in sorted(by_sample_class(signals).items()):
            packed = self.pack([signals[i] for i in idx])
            out = self.compute_device(packed, dtype)
            host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            host.copy_(out, non_blocking=True)
            torch.cuda.current_stream(out.device).synchronize()
            for i, m in zip(idx, packed.split(host.numpy())):
                result[i] = m
        return result

    def stage(self, stage, signals):
        """TFK_STAGE_*: frames / magnitude spectrum / power spectrum of each signal, float64"""
        torch = _torch()
        signals = [np.asarray(s) for s in signals]
        groups = by_sample_class(signals)
        if len(groups) > 1:
            result = [None] * len(signals)
            for _, idx in sorted(groups.items()):
                for i, m in zip(idx, self.stage(stage, [signals[i] for i in idx])):
                    result[i] = m
            return result
        packed = self.pack(signals)
        cols = self.cfg.frame_len if stage == _lib.STAGE_FRAMES else self.nbins
        out = torch.empty((packed.n_frames, cols), dtype=torch.float64, device=packed.signal.device)
        stream = torch.cuda.current_stream(packed.signal.device).cuda_stream
        check(self.lib.tfk_feat_stage(self._h, c_void_p(stream), stage, _dptr(packed.signal), packed.sample_type,
                                      _dptr(packed.d_sig_off), _dptr(packed.d_frame_off), packed.n_utts, packed.n_frames,
                                      _dptr(out), cols))
        return packed.split(out.cpu().numpy())


def dynamic(matrices, order, deriv_only=False, dtype=np.float64, device=0):
    """base.deriv / delta / ddelta for a batch of float64 matrices ('reflect' boundary per matrix)"""
    torch = _torch()
    lib = _lib.load()
    mats = [np.ascontiguousarray(m, dtype=np.float64) for m in matrices]
    if not mats:
        return []
    dim = mats[0].shape[1]
    for m in mats:
        if m.ndim != 2 or m.shape[1] != dim:
            raise ValueError("matrices of one batch must share their column count")
    rows = np.array([m.shape[0] for m in mats], dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(rows))).astype(np.int64)
    n = int(off[-1])
    cols = dim if deriv_only else dim * (1 + order)
    if n == 0 or dim == 0:
        return [np.zeros((m.shape[0], cols), dtype=dtype) for m in mats]
    dev = torch.device("cuda", device)
    x = torch.from_numpy(np.concatenate(mats)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
    out = torch.empty((n, cols), dtype=tdt, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib.tfk_feat_dynamic(c_void_p(stream), _dptr(x), dim, dim, _dptr(d_off), len(mats), n, int(order),
                               int(bool(deriv_only)), _dptr(out), cols, int(tdt == torch.float64)))
    res = out.cpu().numpy()
    return [res[off[i]:off[i + 1]] for i in range(len(mats))]


def cmvn_stats(speakers, device=0):
    """compute_cmvn's statistics for a list of speakers, each a list of float32 [N_i, D] utterance matrices:
    one [2, D+1] float64 array per speaker (prepare_data.py:103-111)"""
    torch = _torch()
    lib = _lib.load()
    utts = [np.ascontiguousarray(u, dtype=np.float32) for spk in speakers for u in spk]
    if not speakers:
        return []
    if not utts:
        raise ValueError("speakers without utterances")
    dim = utts[0].shape[1]
    for u in utts:
        if u.ndim != 2 or u.shape[1] != dim:
            raise ValueError("all utterances must share the feature dimension")
    lens = np.array([u.shape[0] for u in utts], dtype=np.int64)
    rows = np.concatenate(([0], np.cumsum(lens)))[:-1].astype(np.int64)
    spk_off = np.concatenate(([0], np.cumsum([len(s) for s in speakers]))).astype(np.int64)
    dev = torch.device("cuda", device)
    flat = np.concatenate(utts) if lens.sum() else np.zeros((1, dim), dtype=np.float32)
    feats = torch.from_numpy(flat).to(dev)
    d_spk, d_rows, d_lens = (torch.from_numpy(a).to(dev) for a in (spk_off, rows, lens))
    stats = torch.empty((len(speakers), 2, dim + 1), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib.tfk_cmvn_stats(c_void_p(stream), _dptr(feats), dim, dim, _dptr(d_spk), _dptr(d_rows), _dptr(d_lens),
                             len(speakers), _dptr(stats)))
    res = stats.cpu().numpy()
    return [res[i] for i in range(len(speakers))]
