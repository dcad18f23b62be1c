"""Decoding environment with the reference's interface (neuralNetworks/decoder.py)."""
import os

import numpy as np

from ..processing.feature_reader import Unspliced, cmvn_table
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
            from ..dataparallel import local_device
            device = local_device()
        self.engine = classifier.create_engine(input_dim, max_frames=min(max(int(max_length), 1), 1 << 16), device=device)
        self.saver = ModelSaver(self.engine)
        self.graph.finalize()

    def _check(self, inputs):
        inputs = np.asarray(inputs)
        if inputs.shape[0] > self.max_length:  # the reference's zero padding has a negative size here
            raise ValueError("negative dimensions are not allowed")
        return inputs

    def _run(self, inputs, **kw):
        if isinstance(inputs, Unspliced):  # splice on the device
            self._check(inputs)
            return self.engine.posteriors_raw(np.asarray(inputs), [inputs.shape[0]], inputs.context_width,
                                              cmvn=cmvn_table([inputs]), **kw)
        return self.engine.posteriors(self._check(inputs), **kw)

    def __call__(self, inputs):
        """NxF features -> NxO state posteriors (reference decoder.py:49-71)"""
        return self._run(inputs)

    def log_likelihoods(self, inputs):
        """log(posterior / prior), fused in the softmax kernel; needs set_prior (reference nnet.py:280-286)"""
        return self._run(inputs, log_div_prior=True)

    def decode_batch(self, utterances, log_div_prior=True):
        """Several utterances in ONE forward pass: returns a list of per-utterance [N_i, O] arrays.
        All `Unspliced` -> device-side splice with utterance boundaries; otherwise the spliced matrices are
        simply stacked (frames are independent once spliced)."""
        lens = [u.shape[0] for u in utterances]
        for u in utterances:
            self._check(u)
        if all(isinstance(u, Unspliced) for u in utterances):
            flat = self.engine.posteriors_raw(np.concatenate([np.asarray(u) for u in utterances]), lens,
                                              utterances[0].context_width, log_div_prior=log_div_prior,
                                              cmvn=cmvn_table(utterances))
        else:
            stack = np.concatenate([u.spliced() if isinstance(u, Unspliced) else np.asarray(u, dtype=np.float32)
                                    for u in utterances])
            flat = self.engine.posteriors(stack, log_div_prior=log_div_prior)
        return np.split(flat, np.cumsum(lens)[:-1])

    def set_prior(self, prior):
        self.engine.set_prior(prior)

    def restore(self, filename):
        """load the saved neural net (reference decoder.py:73-81)"""
        self.saver.restore(None, filename)

    def close(self):
        self.engine.close()
