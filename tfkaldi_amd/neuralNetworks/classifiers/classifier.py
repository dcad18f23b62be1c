"""Abstract classifier (reference neuralNetworks/classifiers/classifier.py)."""
from abc import ABCMeta, abstractmethod


class Classifier(object, metaclass=ABCMeta):
    """a neural-net classifier with `output_dim` outputs"""

    def __init__(self, output_dim):
        self.output_dim = output_dim

    @abstractmethod
    def __call__(self, inputs, seq_length, is_training=False, reuse=False, scope=None):
        """-> (logits, logit sequence lengths, saver, control ops)   (reference classifier.py:16-37)"""
        raise NotImplementedError("Abstract method")

    @abstractmethod
    def engine_config(self, input_dim, **trainer_options):
        """the C-ABI description (tfk_config) of this classifier for `input_dim` inputs"""
