"""Signal framing and spectra with the reference's interface (processing/sigproc.py), computed on the GPU
(csrc/features.hip through tfkaldi_amd/features.py; float64 in, float64 out; no CPU path).

`deframesig` and `logpowspec` of the reference module have no caller anywhere in the reference and are not provided."""
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
    return out
