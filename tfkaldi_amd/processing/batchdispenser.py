"""Utterance-level mini-batch dispensers (interface of the reference's processing/batchdispenser.py)."""
import gzip
from abc import ABCMeta, abstractmethod

import numpy as np


class BatchDispenser(object, metaclass=ABCMeta):
    """Hands out batches of `size` utterances as (list of [N_i, F] float32, list of [N_i] uint32)."""

    @abstractmethod
    def read_target_file(self, target_path):
        """dict utterance id -> target string"""

    def __init__(self, feature_reader, target_coder, size, target_path):
        self.feature_reader = feature_reader
        self.target_dict = self.read_target_file(target_path)
        # longest encoded target sequence (reference batchdispenser.py:51-52)
        self.max_target_length = max(target_coder.encode(t).size for t in self.target_dict.values())
        self.size = size
        self.target_coder = target_coder

    def get_batch(self):
        """Next `size` usable utterances; utterances without targets or too short to splice are skipped
        with the reference's WARNING lines (batchdispenser.py:74-91)."""
        batch_inputs, batch_targets = [], []
        while len(batch_inputs) < self.size:
            utt_id, utt_mat, _ = self.feature_reader.get_utt()
            if utt_id in self.target_dict and utt_mat is not None:
                batch_inputs.append(utt_mat)
                batch_targets.append(self.target_coder.encode(self.target_dict[utt_id]))
            else:
                if utt_id not in self.target_dict:
                    print("WARNING no targets for %s" % utt_id)
                if utt_mat is None:
                    print("WARNING %s is too short to splice" % utt_id)
        return batch_inputs, batch_targets

    def split(self):
        """split off what has been read (used to carve the validation set: nnet.py:89-96)"""
        self.feature_reader.split()

    def _move(self, step_fn):
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
        encoded = np.concatenate([self.target_coder.encode(t) for t in self.target_dict.values()])
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
