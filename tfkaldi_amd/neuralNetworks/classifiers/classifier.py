"""Abstract classifier (reference neuralNetworks/classifiers/classifier.py).

In the reference a classifier is a callable that appends TensorFlow ops to the graph under construction
(classifier.py:16-37) and the Trainer / Decoder call it to build their graphs.  Here the graph + session is one
HIP engine, so a classifier DESCRIBES the network to the engine instead of executing it op by op.  The contract a
subclass has to fulfil to be trainable, decodable and checkpointable is ONE method:

    engine_config(input_dim, **trainer_options) -> tfk_config     (tfkaldi_amd._lib.make_config(...))

Everything the Trainer / Decoder / Nnet need is built on it in this base class and can be overridden:
    create_engine(input_dim, torch_state=False, **trainer_options)   the engine ("graph + session") for this net
    initialize(engine, rng)      the variable initialisers of layer.py:39-48 / dnn.py:67-68
    control_ops(engine)          {'add', 'init'} with layer-wise initialisation, else None (dnn.py:114-122)
    __call__(inputs, seq_length, is_training=False, reuse=False, scope=None)
                                 the reference's call signature, evaluated eagerly in inference mode
tests/test_gpu_nnet_e2e.py::test_minimal_classifier_subclass_trains checks a subclass that defines nothing else.
"""
from abc import ABCMeta, abstractmethod

import numpy as np


class ControlOp(object):
    """a graph operation with a .run() (the reference's control_ops values are tf Operations)"""

    def __init__(self, fn):
        self._fn = fn

    def run(self, feed_dict=None, session=None):
        self._fn()


class Classifier(object, metaclass=ABCMeta):
    """a neural-net classifier with `output_dim` outputs"""

    def __init__(self, output_dim):
        self.output_dim = output_dim
        self._scopes = {}

    @abstractmethod
    def engine_config(self, input_dim, **trainer_options):
        """the C-ABI description (tfk_config) of this classifier for `input_dim` inputs; `trainer_options` are the
        keyword arguments of tfkaldi_amd._lib.make_config that belong to the training environment
        (init_learning_rate, learning_rate_decay, num_steps, max_frames, seed, device)"""

    def create_engine(self, input_dim, torch_state=False, **trainer_options):
        """the engine that holds this classifier's variables and executes it (reference: the graph the Trainer /
        Decoder build by calling the classifier, trainer.py:72-79, decoder.py:36-38)"""
        from ...engine import Engine
        return Engine(self.engine_config(input_dim, **trainer_options), torch_state=torch_state)

    def initialize(self, engine, rng):
        """run the variable initialisers: hidden weights N(0, 1/sqrt(d_in)), output weights N(0, 0) = 0, biases 0,
        beta 0, moving mean 0 / variance 1 (layer.py:39-48, dnn.py:67-68)"""
        from ... import _lib
        L, H = engine.L, engine.H
        for l in range(L):
            d_in = engine.F if l == 0 else H
            engine.set(_lib.WEIGHTS, l, (rng.standard_normal((d_in, H)) * (1.0 / np.sqrt(d_in))).astype(np.float32))
            engine.set(_lib.BIASES, l, np.zeros(H, dtype=np.float32))
            if engine.batch_norm:
                engine.set(_lib.BN_BETA, l, np.zeros(H, dtype=np.float32))
                engine.set(_lib.BN_MOVING_MEAN, l, np.zeros(H, dtype=np.float32))
                engine.set(_lib.BN_MOVING_VAR, l, np.ones(H, dtype=np.float32))
        engine.set(_lib.WEIGHTS, L, np.zeros((H, engine.O), dtype=np.float32))
        engine.set(_lib.BIASES, L, np.zeros(engine.O, dtype=np.float32))
        engine.set_scalar(_lib.INITIALISED_LAYERS, 0)

    def control_ops(self, engine):
        """{'add': ..., 'init': ...} with layer-wise initialisation, else None (dnn.py:114-122)"""
        if not engine.cfg.layerwise_init:
            return None
        return {"add": ControlOp(engine.add_layer), "init": ControlOp(engine.init_last_layer)}

    def saver(self, engine):
        from .dnn import ModelSaver
        return ModelSaver(engine)

    # ---- the reference's call signature, evaluated eagerly ----
    def __call__(self, inputs, seq_length, is_training=False, reuse=False, scope=None):
        """Forward computation on sequential data: `inputs` is a list with a [batch, input_dim] array per
        time step, `seq_length` the utterance lengths.  Returns (sequential logits, seq_length, saver,
        control_ops) like reference dnn.py:37-131.  Variables live in `scope`; reuse=False creates them
        (freshly initialised), reuse=True shares the ones created earlier.  Only inference mode is
        available through this entry point: the training graph (batch-statistics batch norm, dropout, the
        gradient accumulators) is what a Trainer builds around the classifier -- here Trainer.update."""
        from . import seq_convertors
        if is_training:
            raise NotImplementedError("training-mode evaluation runs through neuralNetworks.trainer.Trainer")
        scope = scope or type(self).__name__
        if reuse:
            if scope not in self._scopes:
                raise ValueError("Variable scope %s does not exist, reuse=True" % scope)
            engine = self._scopes[scope]
        else:
            if scope in self._scopes:
                raise ValueError("Variable scope %s already exists, did you mean to set reuse=True?" % scope)
            engine = self.create_engine(int(np.asarray(inputs[0]).shape[1]))
            self.initialize(engine, np.random.default_rng())
            self._scopes[scope] = engine
        flat = seq_convertors.seq2nonseq([np.asarray(x, dtype=np.float32) for x in inputs], seq_length)
        logits = engine.posteriors(flat, raw_logits=True)
        seq_logits = seq_convertors.nonseq2seq(logits, seq_length, len(inputs))
        return seq_logits, seq_length, self.saver(engine), self.control_ops(engine)
