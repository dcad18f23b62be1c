"""Utterance-level mini-batch dispensers (interface of the reference's processing/batchdispenser.py)."""
import gzip
import weakref
from abc import ABCMeta, abstractmethod

import numpy as np

from .feature_reader import apply_cmvn


class _BufferPool(object):
    """Byte buffers for packed batches, recycled when the LAST array that views a lease has gone (a caller may hold a
    batch for as long as it likes -- the held-out validation batches live for the whole run)."""

    def __init__(self, keep=4):
        self.free, self.keep = [], keep

    def _give_back(self, block):
        if len(self.free) < self.keep:
            self.free.append(block)

    def lease(self, nbytes):
        """uint8 array of `nbytes`; every view derived from it keeps it alive (numpy stops collapsing `.base` chains at
        an array whose own base is not an array), and its collection hands the block back"""
        size = 1 << 16  # blocks come in powers of two: batches of similar length share them
        while size < nbytes:
            size <<= 1
        block = None
        for i, cand in enumerate(self.free):
            if size <= cand.nbytes <= 2 * size:
                block = self.free.pop(i)
                break
        if block is None:
            block = np.empty(size, dtype=np.uint8)
        lease = np.frombuffer(memoryview(block)[:nbytes], dtype=np.uint8)
        weakref.finalize(lease, self._give_back, block)
        return lease


class PackedBatch(object):
    """The utterances ONE consumer needs from a batch, in the layout the engine's raw entry points take
    (include/tfkaldi_hip.h: tfk_accumulate_raw): frames as stored in the archive, back to back, their lengths, the
    speakers' (mean, std) rows and the encoded targets back to back.  CMVN and the +-context splice happen in HBM.

    frames       [T, D] float32, C-contiguous
    lens         [U] int32 frames per utterance
    cmvn         [U, 2, D] float32, or None when every utterance is already normalised
    targets      [sum(target_lens)] int32
    target_lens  [U] int32
    groups       [(first utterance, end utterance, first row, end row, first target, end target)] per micro-batch
    info         whatever the selector returned beside the groups (the trainer's: micro-batches in the whole step and
                 this rank's slice of them)
    utt_ids      ids of the packed utterances; batch_utts = usable utterances in the whole batch
    """

    __slots__ = ("frames", "lens", "cmvn", "targets", "target_lens", "groups", "info", "utt_ids", "batch_utts",
                 "context_width", "warnings")

    @property
    def num_frames(self):
        return self.frames.shape[0]


def select_all(rows):
    """selector of a consumer that wants every utterance as one group"""
    return [list(range(len(rows)))], None


class BatchDispenser(object, metaclass=ABCMeta):
    """Hands out batches of `size` utterances as (list of [N_i, F] float32, list of [N_i] uint32).

    Besides the reference's interface there is a PACKED path for the trainer (`next_packed` / `prefetch`): the
    dispenser walks the scp cursor over the batch reading headers only, asks a selector which utterances the consumer
    needs (a data-parallel rank: the utterances of ITS micro-batches), and reads exactly those straight into one batch
    buffer.  `prefetch` produces the next batch ahead of time (the trainer calls it while the GPU works on the current
    one); every call that moves the cursor first puts an unused prefetch back, so the sequence of batches is the one
    get_batch alone would give."""

    @abstractmethod
    def read_target_file(self, target_path):
        """dict utterance id -> target string"""

    def __init__(self, feature_reader, target_coder, size, target_path):
        self.feature_reader = feature_reader
        self.target_dict = self.read_target_file(target_path)
        # every target is encoded once (the reference does the same pass for max_target_length, batchdispenser.py:51-52,
        # throws the result away and encodes again in every get_batch)
        self._encoded = {utt: target_coder.encode(t) for utt, t in self.target_dict.items()}
        self.max_target_length = max(t.size for t in self._encoded.values())
        self.size = size
        self.target_coder = target_coder
        self._pool = _BufferPool()
        self._ahead = None  # (PackedBatch, selector, num_utt, cursor before it was produced)

    # ---- the walk over the scp both paths share ----
    def _plan(self, num_utt):
        """advance the reader over the next `num_utt` usable utterances (headers only): [(scp index, utt_id, frames)]
        and the reference's WARNING lines for the ones passed over (batchdispenser.py:74-91)"""
        reader = self.feature_reader
        plan, warnings = [], []
        while len(plan) < num_utt:
            index, utt_id, frames, _ = reader.next_entry()
            known, long_enough = utt_id in self.target_dict, frames >= reader.min_frames
            if known and long_enough:
                plan.append((index, utt_id, frames))
            else:
                if not known:
                    warnings.append("WARNING no targets for %s" % utt_id)
                if not long_enough:
                    warnings.append("WARNING %s is too short to splice" % utt_id)
        return plan, warnings

    def get_batch(self):
        """Next `size` usable utterances; utterances without targets or too short to splice are skipped
        with the reference's WARNING lines (batchdispenser.py:74-91)."""
        self._put_back()
        if not hasattr(self.feature_reader, "next_entry"):  # a reader with the reference's interface only
            return self._get_batch_serial()
        plan, warnings = self._plan(self.size)
        for line in warnings:
            print(line)
        reader = self.feature_reader
        inputs = [reader.finish(utt_id, reader.reader.read_utt_data(index)) for index, utt_id, _ in plan]
        return inputs, [self._encoded[utt_id].copy() for _, utt_id, _ in plan]

    def _get_batch_serial(self):
        batch_inputs, batch_targets = [], []
        while len(batch_inputs) < self.size:
            utt_id, utt_mat, _ = self.feature_reader.get_utt()
            if utt_id in self.target_dict and utt_mat is not None:
                batch_inputs.append(utt_mat)
                batch_targets.append(self._encoded[utt_id].copy())
            else:
                if utt_id not in self.target_dict:
                    print("WARNING no targets for %s" % utt_id)
                if utt_mat is None:
                    print("WARNING %s is too short to splice" % utt_id)
        return batch_inputs, batch_targets

    # ---- packed path ----
    @property
    def packed(self):
        """can this dispenser produce packed batches (a FeatureReader of this package behind it)?"""
        return hasattr(self.feature_reader, "next_entry")

    def _produce(self, select, num_utt):
        # WHERE the normalisation and the splice happen on this path is fixed: float32 archives with float32 statistics travel
        # raw and are normalised + spliced in HBM (bit-identical to the host's arithmetic: tests/test_gpu_device_splice.py,
        # tests/test_packed_feed.py), float64 data is normalised on the host in float64 as the reference does.  The reader's
        # `cmvn_on_device` / `splice_on_device` flags select the same thing for get_utt() / get_batch() ONLY; a user who wants the
        # host to do this work feeds the reference's way (`[nnet] packed_feed = False`, trainer.update(*dispenser.get_batch())).
        reader = self.feature_reader
        ark_reader = reader.reader
        plan, warnings = self._plan(num_utt)
        groups, info = select([frames for _, _, frames in plan])
        wanted = [plan[i] for group in groups for i in group]
        dim = ark_reader.entry(wanted[0][0])[3] if wanted else 0
        total = sum(frames for _, _, frames in wanted)
        out = PackedBatch()
        out.frames = self._pool.lease(total * dim * 4).view(np.float32).reshape(total, dim)
        out.lens = np.fromiter((frames for _, _, frames in wanted), dtype=np.int32, count=len(wanted))
        tables, row, deferred = [], 0, False
        for index, utt_id, frames in wanted:
            dest = out.frames[row:row + frames]
            row += frames
            entry = ark_reader.entry(index)
            if entry[3] != dim:
                raise ValueError("%s has %d-dimensional frames, the batch started with %d" % (utt_id, entry[3], dim))
            table = reader.speaker_table(utt_id) if entry[4] == np.float32 else None
            if table is not None:
                ark_reader.read_into(index, dest)
                deferred = True
            else:  # float64 archive or statistics: normalised on the host in float64, as the reference does
                dest[...] = apply_cmvn(ark_reader.read_utt_data(index), reader.speaker_stats(utt_id))
            tables.append(table)
        if deferred:
            identity = None
            for i, table in enumerate(tables):
                if table is None:
                    if identity is None:
                        identity = np.stack([np.zeros(dim, dtype=np.float32), np.ones(dim, dtype=np.float32)])
                    tables[i] = identity  # (x - 0) / 1 is exact
            out.cmvn = np.stack(tables)
        else:
            out.cmvn = None
        encoded = [self._encoded[utt_id] for _, utt_id, _ in wanted]
        out.target_lens = np.fromiter((t.size for t in encoded), dtype=np.int32, count=len(encoded))
        out.targets = (np.concatenate(encoded).astype(np.int32) if encoded else np.zeros(0, dtype=np.int32))
        out.groups, utt, row, tgt = [], 0, 0, 0
        for group in groups:
            n = len(group)
            rows = int(out.lens[utt:utt + n].sum())
            labels = int(out.target_lens[utt:utt + n].sum())
            out.groups.append((utt, utt + n, row, row + rows, tgt, tgt + labels))
            utt, row, tgt = utt + n, row + rows, tgt + labels
        out.info, out.utt_ids, out.batch_utts = info, [utt_id for _, utt_id, _ in wanted], len(plan)
        out.context_width, out.warnings = reader.context_width, warnings
        return out

    def next_packed(self, select=select_all, num_utt=None):
        """The next batch (`num_utt` usable utterances, default `size`) as a PackedBatch holding the utterances
        `select` asks for.  select(frames per usable utterance) -> (groups, info): `groups` = lists of positions in the
        batch, one list per micro-batch, packed in that order."""
        num_utt = self.size if num_utt is None else num_utt
        ahead, self._ahead = self._ahead, None
        if ahead is not None:
            if ahead[1] is select and ahead[2] == num_utt:
                batch = ahead[0]
            else:
                self.feature_reader.reader.scp_position = ahead[3]
                batch = self._produce(select, num_utt)
        else:
            batch = self._produce(select, num_utt)
        for line in batch.warnings:
            print(line)
        return batch

    def prefetch(self, select=select_all, num_utt=None):
        """produce the batch the next next_packed(select, num_utt) will hand out; a no-op when one is waiting"""
        if self._ahead is None:
            num_utt = self.size if num_utt is None else num_utt
            cursor = self.feature_reader.reader.scp_position
            self._ahead = (self._produce(select, num_utt), select, num_utt, cursor)

    def _put_back(self):
        """an unused prefetch: the cursor returns to where it was before it was produced"""
        if self._ahead is not None:
            self.feature_reader.reader.scp_position = self._ahead[3]
            self._ahead = None

    def split(self):
        """split off what has been read (used to carve the validation set: nnet.py:89-96)"""
        self._put_back()
        self.feature_reader.split()

    def _move(self, step_fn):
        self._put_back()
        moved = 0
        while moved < self.size:
            if step_fn() in self.target_dict:
                moved += 1

    def skip_batch(self):
        """advance over one batch without reading features (resume: nnet.py:107-108)"""
        self._move(self.feature_reader.next_id)

    def return_batch(self):
        """rewind by one batch (validation rollback: nnet.py:180-181)"""
        self._move(self.feature_reader.prev_id)

    def compute_target_count(self):
        """occurrences of every label over ALL targets (the prior: nnet.py:241-244)"""
        encoded = np.concatenate(list(self._encoded.values()))
        return np.bincount(encoded, minlength=self.target_coder.num_labels)

    @property
    def num_batches(self):
        """number of batches in the data: the reference is Python 2, `num_utt / size` floors
        (batchdispenser.py:148-155)"""
        return self.num_utt // self.size

    @property
    def num_utt(self):
        return len(self.target_dict)

    @property
    def num_labels(self):
        return self.target_coder.num_labels

    @property
    def max_input_length(self):
        return self.feature_reader.max_input_length


def _read_targets(fid):
    targets = {}
    for line in fid:
        fields = line.strip().split(" ")
        targets[fields[0]] = " ".join(fields[1:])
    return targets


class TextBatchDispenser(BatchDispenser):
    """targets from a plain text file "<utt> <words...>" (reference batchdispenser.py:175-198)"""

    def read_target_file(self, target_path):
        with open(target_path, "r") as fid:
            return _read_targets(fid)


class AlignmentBatchDispenser(BatchDispenser):
    """targets from gzipped (possibly concatenated) alignment text "<utt> <pdf> <pdf> ..."
    (reference batchdispenser.py:200-223; main.py:138-140 concatenates the per-job .gz files)"""

    def read_target_file(self, target_path):
        with gzip.open(target_path, "rt") as fid:
            return _read_targets(fid)
