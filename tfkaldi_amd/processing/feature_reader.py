"""Per-utterance feature feed: ark read -> per-speaker CMVN -> +-context splice
(interface of the reference's processing/feature_reader.py)."""
import numpy as np

from . import ark, readfiles


class Unspliced(np.ndarray):
    """Frames [N, D] whose +-context splice -- and, when `cmvn` is set, whose mean/variance normalisation -- is
    deferred to the GPU.  Trainer and Decoder recognise the type and send only these N x D values
    (plus the 2 x D speaker table) over PCIe; `spliced()` gives the host result.

    cmvn: None (frames already normalised) or a [2, D] float32 array (mean, standard deviation)."""

    context_width = 0
    cmvn = None

    def __new__(cls, frames, context_width, cmvn=None):
        obj = np.ascontiguousarray(frames, dtype=np.float32).view(cls)
        obj.context_width = int(context_width)
        obj.cmvn = None if cmvn is None else np.ascontiguousarray(cmvn, dtype=np.float32)
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            self.context_width = getattr(obj, "context_width", 0)
            self.cmvn = getattr(obj, "cmvn", None)

    def normalised(self):
        """the frames after mean/variance normalisation, on the host"""
        frames = np.asarray(self)
        if self.cmvn is None:
            return frames
        return np.divide(np.subtract(frames, self.cmvn[0]), self.cmvn[1])

    def spliced(self):
        return splice(self.normalised(), self.context_width)


def cmvn_table(utterances):
    """[U, 2, D] table for the engine's raw entry points, or None when no utterance defers its normalisation
    (utterances that are already normalised get the identity row mean 0 / std 1: (x - 0) / 1 is exact)"""
    if all(u.cmvn is None for u in utterances):
        return None
    dim = utterances[0].shape[1]
    identity = np.stack([np.zeros(dim, dtype=np.float32), np.ones(dim, dtype=np.float32)])
    return np.stack([identity if u.cmvn is None else u.cmvn for u in utterances])


class FeatureReader(object):
    """Reads features from a Kaldi archive, mean/variance-normalises them per speaker and splices them.
    With splice_on_device=True get_utt() returns `Unspliced` frames (same None-when-too-short rule) and the
    splice happens in HBM; cmvn_on_device=True (implies splice_on_device) also defers the normalisation: the
    frames stay as read from the ark and carry their speaker's (mean, std) table.  Archives that are not float32
    are normalised on the host as before (the device arithmetic is float32, like numpy's on float32 input)."""

    def __init__(self, scpfile, cmvnfile, utt2spkfile, context_width, max_input_length, splice_on_device=False,
                 cmvn_on_device=False):
        self.cmvn_on_device = cmvn_on_device
        self.splice_on_device = splice_on_device or cmvn_on_device
        self.reader = ark.ArkReader(scpfile)
        self.reader_cmvn = ark.ArkReader(cmvnfile)
        self.utt2spk = readfiles.read_utt2spk(utt2spkfile)
        self.context_width = context_width
        self.max_input_length = max_input_length

    def get_utt(self):
        """(utt_id, spliced features or None if too short, looped) -- reference feature_reader.py:42-60"""
        utt_id, utt_mat, looped = self.reader.read_next_utt()
        stats = self.reader_cmvn.read_utt(self.utt2spk[utt_id])
        if self.cmvn_on_device and utt_mat.dtype == np.float32 and stats.dtype == np.float32:
            if utt_mat.shape[0] < 1 + 2 * self.context_width:
                return utt_id, None, looped
            return utt_id, Unspliced(utt_mat, self.context_width, cmvn=np.stack(cmvn_params(stats))), looped
        normalised = apply_cmvn(utt_mat, stats)
        if self.splice_on_device:
            if normalised.shape[0] < 1 + 2 * self.context_width:
                return utt_id, None, looped
            return utt_id, Unspliced(normalised, self.context_width), looped
        return utt_id, splice(normalised, self.context_width), looped

    def next_id(self):
        return self.reader.read_next_scp()

    def prev_id(self):
        return self.reader.read_previous_scp()

    def split(self):
        self.reader.split()


def cmvn_params(stats):
    """(mean, standard deviation) from accumulated statistics (reference feature_reader.py:109-113):
    stats[0] = [sum x ..., frame count], stats[1] = [sum x^2 ..., 0]."""
    count = stats[0, -1]
    mean = stats[0, :-1] / count
    variance = stats[1, :-1] / count - np.square(mean)
    return mean, np.sqrt(variance)


def apply_cmvn(utt, stats):
    """Mean/variance normalisation from accumulated statistics (reference feature_reader.py:91-115)"""
    mean, std = cmvn_params(stats)
    return np.divide(np.subtract(utt, mean), std)


def splice(utt, context_width):
    """Concatenate each frame with its `context_width` left and right neighbours; frames beyond the
    utterance edges are ZERO (not replicated).  Block j of the output row t is frame t + j - context_width
    (reference feature_reader.py:117-156).  Returns None when the utterance has fewer than 2c+1 frames."""
    num_frames, dim = utt.shape
    if num_frames < 1 + 2 * context_width:
        return None
    padded = np.zeros((num_frames + 2 * context_width, dim), dtype=np.float32)
    padded[context_width:context_width + num_frames] = utt
    spliced = np.empty((num_frames, dim * (1 + 2 * context_width)), dtype=np.float32)
    for j in range(1 + 2 * context_width):
        spliced[:, j * dim:(j + 1) * dim] = padded[j:j + num_frames]
    return spliced
