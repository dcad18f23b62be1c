"""Per-utterance feature feed: ark read -> per-speaker CMVN -> +-context splice
(interface of the reference's processing/feature_reader.py)."""
import numpy as np

from . import ark, readfiles


class Unspliced(np.ndarray):
    """CMVN-normalised frames [N, D] whose +-context splice is deferred to the GPU.  Trainer and
    Decoder recognise the type and send only these N x D values over PCIe; `spliced()` gives the host result."""

    context_width = 0

    def __new__(cls, frames, context_width):
        obj = np.ascontiguousarray(frames, dtype=np.float32).view(cls)
        obj.context_width = int(context_width)
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            self.context_width = getattr(obj, "context_width", 0)

    def spliced(self):
        return splice(np.asarray(self), self.context_width)


class FeatureReader(object):
    """Reads features from a Kaldi archive, mean/variance-normalises them per speaker and splices them.
    With splice_on_device=True get_utt() returns `Unspliced` frames (same None-when-too-short rule) and the
    splice happens in HBM."""

    def __init__(self, scpfile, cmvnfile, utt2spkfile, context_width, max_input_length, splice_on_device=False):
        self.splice_on_device = splice_on_device
        self.reader = ark.ArkReader(scpfile)
        self.reader_cmvn = ark.ArkReader(cmvnfile)
        self.utt2spk = readfiles.read_utt2spk(utt2spkfile)
        self.context_width = context_width
        self.max_input_length = max_input_length

    def get_utt(self):
        """(utt_id, spliced features or None if too short, looped) -- reference feature_reader.py:42-60"""
        utt_id, utt_mat, looped = self.reader.read_next_utt()
        stats = self.reader_cmvn.read_utt(self.utt2spk[utt_id])
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


def apply_cmvn(utt, stats):
    """Mean/variance normalisation from accumulated statistics (reference feature_reader.py:91-115):
    stats[0] = [sum x ..., frame count], stats[1] = [sum x^2 ..., 0]."""
    count = stats[0, -1]
    mean = stats[0, :-1] / count
    variance = stats[1, :-1] / count - np.square(mean)
    return np.divide(np.subtract(utt, mean), np.sqrt(variance))


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
