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
    (These two flags concern get_utt() and everything built on it -- get_batch(), Decoder input.  The packed feed,
    BatchDispenser.next_packed, always defers both steps to the device for float32 data: same bits, see there.)
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
        self._stats = {}   # speaker -> accumulated statistics (the reference re-reads them for every utterance)
        self._tables = {}  # speaker -> [2, D] float32 (mean, std), or None when the statistics are not float32

    def speaker_stats(self, utt_id):
        """the accumulated CMVN statistics of the utterance's speaker; read from the archive once per speaker"""
        spk = self.utt2spk[utt_id]
        stats = self._stats.get(spk)
        if stats is None:
            stats = self._stats[spk] = self.reader_cmvn.read_utt(spk)
        return stats

    def speaker_table(self, utt_id):
        """[2, D] float32 (mean, standard deviation) of the utterance's speaker in the arithmetic numpy uses on float32
        statistics -- the operand of the device-side normalisation; None when the statistics are stored as float64
        (the host then normalises in float64, as the reference does)"""
        spk = self.utt2spk[utt_id]
        if spk not in self._tables:
            stats = self.speaker_stats(utt_id)
            self._tables[spk] = np.stack(cmvn_params(stats)) if stats.dtype == np.float32 else None
        return self._tables[spk]

    def get_utt(self):
        """(utt_id, spliced features or None if too short, looped) -- reference feature_reader.py:42-60"""
        utt_id, utt_mat, looped = self.reader.read_next_utt()
        return utt_id, self.finish(utt_id, utt_mat), looped

    def finish(self, utt_id, utt_mat):
        """what get_utt returns for the frames `utt_mat` as read from the archive: normalised and spliced, or the
        deferred (`Unspliced`) form"""
        if self.cmvn_on_device and utt_mat.dtype == np.float32:
            table = self.speaker_table(utt_id)
            if table is not None:
                if utt_mat.shape[0] < 1 + 2 * self.context_width:
                    return None
                return Unspliced(utt_mat, self.context_width, cmvn=table)
        normalised = apply_cmvn(utt_mat, self.speaker_stats(utt_id))
        if self.splice_on_device:
            if normalised.shape[0] < 1 + 2 * self.context_width:
                return None
            return Unspliced(normalised, self.context_width)
        return splice(normalised, self.context_width)

    # ---- planning interface of the batch dispenser's packed path: walk the cursor without fetching ----
    def next_entry(self):
        """advance like get_utt but read only the entry's header: (scp index, utt_id, frames, looped)"""
        index, utt_id, looped = self.reader.next_entry()
        return index, utt_id, self.reader.entry(index)[2], looped

    @property
    def min_frames(self):
        """utterances shorter than this cannot be spliced (get_utt returns None for them)"""
        return 1 + 2 * self.context_width

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
