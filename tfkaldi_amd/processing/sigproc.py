"""Signal framing and spectra with the reference's interface (processing/sigproc.py), computed on the GPU
(csrc/features.hip through tfkaldi_amd/features.py; float64 in, float64 out; no CPU path)."""
from ctypes import c_void_p

import numpy as np

from .. import _lib, features


def _ones(n):
    return np.ones((n,))


def _plan(frame_len, frame_step, nfft, preemph=0.0):
    """a plan that is only asked for its stages: the filterbank is a placeholder"""
    nfft = int(nfft)
    return features.FeaturePlan("fbank", "nodelta", frame_len, frame_step, nfft, 1,
                                np.zeros((1, nfft // 2 + 1)), preemph=preemph)


def framesig(sig, frame_len, frame_step, winfunc=_ones):
    """sigproc.py:33-67: [NUMFRAMES, frame_len] overlapping frames of the zero-padded signal, times the window"""
    frame_len, frame_step = features.py2_round(frame_len), features.py2_round(frame_step)
    plan = _plan(frame_len, frame_step, 32)
    frames = plan.stage(_lib.STAGE_FRAMES, [np.asarray(sig)])[0]
    plan.close()
    if winfunc is _ones:
        return frames
    import torch  # an elementwise window product, on the device as well (the reference never passes a window)
    win = torch.from_numpy(np.asarray(winfunc(frame_len), dtype=np.float64)).cuda()
    return (torch.from_numpy(frames).cuda() * win).cpu().numpy()


def _rows_as_signal(frames):
    frames = np.ascontiguousarray(frames, dtype=np.float64)
    if frames.ndim != 2:
        raise ValueError("frames must be a [NUMFRAMES, frame_len] matrix")
    return frames


def _spectrum(frames, nfft, stage):
    frames = _rows_as_signal(frames)
    n, width = frames.shape
    if n == 0:
        return np.zeros((0, int(nfft) // 2 + 1))
    plan = _plan(width, width, nfft)  # every row is one frame: step = length, no pre-emphasis
    out = plan.stage(stage, [frames.reshape(-1)])[0]
    plan.close()
    return out


def magspec(frames, nfft):
    """sigproc.py:125-139: |rfft(frames, nfft)|, [N, nfft/2+1]"""
    return _spectrum(frames, nfft, _lib.STAGE_MAGSPEC)


def powspec(frames, nfft):
    """sigproc.py:141-153: 1/nfft * |rfft(frames, nfft)|^2"""
    return _spectrum(frames, nfft, _lib.STAGE_POWSPEC)


def preemphasis(signal, coeff=0.95):
    """sigproc.py:180-191: y[0] = x[0], y[n] = x[n] - coeff * x[n-1]"""
    signal = np.asarray(signal)
    if signal.size == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # numpy's, from `signal[0]`
    chunk = 1024
    plan = _plan(chunk, chunk, 32, preemph=coeff)
    out = plan.stage(_lib.STAGE_FRAMES, [signal])[0].reshape(-1)[:signal.size]
    plan.close()
    return out.astype(np.float32) if signal.dtype == np.float32 else out  # (numpy stays in float32 for a float32 signal)


def deframesig(frames, siglen, frame_len, frame_step, winfunc=_ones):
    """sigproc.py:69-123: overlap-add that undoes framesig; every sample is divided by the summed window of the frames
    covering it; truncated to siglen samples (siglen <= 0: everything)"""
    torch = features._torch()
    frame_len, frame_step = features.py2_round(frame_len), features.py2_round(frame_step)
    frames = np.ascontiguousarray(frames, dtype=np.float64)
    numframes = frames.shape[0]
    assert frames.shape[1] == frame_len, '"frames" matrix is wrong size, 2nd dim is not equal to frame_len'
    padlen = (numframes - 1) * frame_step + frame_len
    if siglen <= 0:
        siglen = padlen
    if numframes == 0:
        return np.zeros((0,))
    lib = _lib.load()
    d_frames = torch.from_numpy(frames).cuda()
    win = None if winfunc is _ones else torch.from_numpy(np.ascontiguousarray(winfunc(frame_len), dtype=np.float64)).cuda()
    out = torch.empty(padlen, dtype=torch.float64, device=d_frames.device)
    stream = torch.cuda.current_stream(d_frames.device).cuda_stream
    _lib.check(lib.tfk_deframesig(c_void_p(stream), c_void_p(d_frames.data_ptr()), frame_len, numframes, frame_len, frame_step,
                                  c_void_p(win.data_ptr()) if win is not None else c_void_p(None), c_void_p(out.data_ptr())))
    return out.cpu().numpy()[0:int(siglen)]


def logpowspec(frames, nfft, norm=1):
    """sigproc.py:155-178: 10 log10 of the power spectrum (floored at 1e-30), minus its maximum over all frames if norm"""
    torch = features._torch()
    frames = _rows_as_signal(frames)
    n, width = frames.shape
    if n == 0:
        return np.zeros((0, int(nfft) // 2 + 1))
    plan = _plan(width, width, nfft)
    packed = plan.pack([frames.reshape(-1)])
    cols = plan.nbins
    ps = torch.empty((packed.n_frames, cols), dtype=torch.float64, device=packed.signal.device)
    scratch = torch.empty(1024, dtype=torch.float64, device=ps.device)
    stream = c_void_p(torch.cuda.current_stream(ps.device).cuda_stream)
    _lib.check(plan.lib.tfk_feat_stage(plan._h, stream, _lib.STAGE_POWSPEC, c_void_p(packed.signal.data_ptr()), packed.sample_type,
                                       c_void_p(packed.d_sig_off.data_ptr()), c_void_p(packed.d_frame_off.data_ptr()),
                                       packed.n_utts, packed.n_frames, c_void_p(ps.data_ptr()), cols))
    _lib.check(plan.lib.tfk_logpow(stream, c_void_p(ps.data_ptr()), ps.numel(), int(bool(norm)), c_void_p(scratch.data_ptr())))
    out = ps.cpu().numpy()
    plan.close()
    return out
