"""Target coders: target string -> uint32 label vector (interface of the reference's processing/target_coder.py)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class TargetCoder(object, metaclass=ABCMeta):
    """Maps the space-separated symbols of a (normalised) target string to alphabet indices."""

    def __init__(self, target_normalizer):
        self.target_normalizer = target_normalizer
        self.alphabet = list(self.create_alphabet())
        self.lookup = {symbol: index for index, symbol in enumerate(self.alphabet)}

    @abstractmethod
    def create_alphabet(self):
        """the list of symbols, index = label"""

    def encode(self, targets):
        """uint32 labels of the normalised targets (reference target_coder.py:36-55)"""
        normalized = self.target_normalizer(targets, self.lookup.keys())
        return np.array([self.lookup[t] for t in normalized.split(" ")], dtype=np.uint32)

    def decode(self, encoded_targets):
        return " ".join(self.alphabet[int(e)] for e in encoded_targets)

    @property
    def num_labels(self):
        return len(self.lookup)


class TextCoder(TargetCoder):
    """character targets (reference target_coder.py:79-118)"""

    def create_alphabet(self):
        specials = ["<eos>", "<sos>", "<space>", ",", ".", "'", "-", "?", "<unk>"]
        return specials + [chr(c) for c in range(ord("a"), ord("z") + 1)]


class AlignmentCoder(TargetCoder):
    """pdf-id alignments: the alphabet is "0" ... str(num_targets - 1) (reference target_coder.py:120-142)"""

    def __init__(self, target_normalizer, num_targets):
        self.num_targets = num_targets
        super(AlignmentCoder, self).__init__(target_normalizer)

    def create_alphabet(self):
        return [str(t) for t in range(self.num_targets)]
