"""Decoding environment with the reference's interface (neuralNetworks/decoder.py)."""
import os

import numpy as np

from .classifiers.dnn import ModelSaver


class _Graph(object):
    def finalize(self):
        pass


class Decoder(object):
    """Forward-only environment: evaluation-mode classifier + softmax (reference decoder.py:11-47)."""

    def __init__(self, classifier, input_dim, max_length, device=None):
        """
        Args:
            classifier: the classifier that will be used for decoding
            input_dim: the input dimension to the nnnetgraph
            max_length: the maximal utterance length
        """
        self.graph = _Graph()
        self.max_length = max_length
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.engine = classifier.create_engine(input_dim, max_frames=min(max(int(max_length), 1), 1 << 16), device=device)
        self.saver = ModelSaver(self.engine)
        self.graph.finalize()

    def _check(self, inputs):
        inputs = np.asarray(inputs)
        if inputs.shape[0] > self.max_length:  # the reference's zero padding has a negative size here
            raise ValueError("negative dimensions are not allowed")
        return inputs

    def __call__(self, inputs):
        """NxF features -> NxO state posteriors (reference decoder.py:49-71)"""
        return self.engine.posteriors(self._check(inputs))

    def log_likelihoods(self, inputs):
        """log(posterior / prior), fused in the softmax kernel; needs set_prior (reference nnet.py:280-286)"""
        return self.engine.posteriors(self._check(inputs), log_div_prior=True)

    def set_prior(self, prior):
        self.engine.set_prior(prior)

    def restore(self, filename):
        """load the saved neural net (reference decoder.py:73-81)"""
        self.saver.restore(None, filename)

    def close(self):
        self.engine.close()
