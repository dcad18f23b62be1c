"""The DNN classifier (reference neuralNetworks/classifiers/dnn.py): `num_layers` hidden FFLayers sharing
one activation chain plus a linear output layer, executed by the HIP engine."""
import os

import numpy as np

from ... import _lib
from .activation import TfActivation, engine_options
from .classifier import Classifier, ControlOp  # noqa: F401  (ControlOp: re-exported)
from .layer import FFLayer


class ModelSaver(object):
    """Stands in for the `tf.train.Saver` the reference creates inside the classifier scope (dnn.py:129):
    it holds every Classifier variable -- weights, biases, batch-norm beta and moving statistics,
    `initialisedlayers` -- and nothing of the optimiser.  One file per checkpoint, written at exactly the
    path prefix the caller gives (numpy .npz container)."""

    def __init__(self, engine):
        self.engine = engine

    def save(self, sess, filename):
        # written next to the target and renamed: a crash while saving cannot damage the previous checkpoint
        # (nnet.py's `validated` fallback model lives at a fixed path)
        tmp = "%s.tmp%d" % (filename, os.getpid())
        with open(tmp, "wb") as fid:
            np.savez(fid, **self.engine.model_tensors())
        os.replace(tmp, filename)

    def restore(self, sess, filename):
        with np.load(filename) as data:
            self.engine.load_model_tensors({k: data[k] for k in data.files})


class DNN(Classifier):
    """feed-forward fully connected network"""

    def __init__(self, output_dim, num_layers, num_units, activation, layerwise_init=True,
                 compute_dtype="float32"):
        """(reference dnn.py:17-35) + compute_dtype: "float32" = the reference's arithmetic (emulated exactly-split on the
        bf16 matrix pipe; "float32_mfma" = the exact fp32 matrix instructions), "bfloat16" = mixed precision (bf16 MFMA
        contractions, everything else fp32; BASELINE cfg3 / cfg4) -- tfkaldi_amd/_lib.py: DTYPES"""
        super(DNN, self).__init__(output_dim)
        self.num_layers = num_layers
        self.num_units = num_units
        self.activation = activation
        self.layerwise_init = layerwise_init
        self.compute_dtype = compute_dtype

    # ---- structure ----
    def layers(self):
        """the FFLayer objects of dnn.py:64-68: one shared hidden layer description and the output layer,
        whose weights start at zero (weights_std = 0)"""
        hidden = FFLayer(self.num_units, self.activation)
        out = FFLayer(self.output_dim, TfActivation(None, "linear"), 0)
        return hidden, out

    def engine_config(self, input_dim, init_learning_rate=1e-3, learning_rate_decay=1.0, num_steps=1,
                      max_frames=1024, seed=0, device=0):
        opts = engine_options(self.activation)
        return _lib.make_config(input_dim, self.num_layers, self.num_units, self.output_dim,
                                layerwise_init=self.layerwise_init, init_learning_rate=init_learning_rate,
                                learning_rate_decay=learning_rate_decay, num_steps=num_steps, max_frames=max_frames,
                                seed=seed, device=device, compute_dtype=self.compute_dtype, **opts)

    def initialize(self, engine, rng):
        """run the variable initialisers: hidden weights N(0, 1/sqrt(d_in)), output weights N(0, 0) = 0,
        biases 0, beta 0, moving mean 0 / variance 1 (layer.py:39-48, dnn.py:67-68)"""
        hidden, out = self.layers()
        for l in range(self.num_layers):
            d_in = engine.F if l == 0 else self.num_units
            engine.set(_lib.WEIGHTS, l, hidden.initial_weights(d_in, rng))
            engine.set(_lib.BIASES, l, np.zeros(self.num_units, dtype=np.float32))
            if engine.batch_norm:
                engine.set(_lib.BN_BETA, l, np.zeros(self.num_units, dtype=np.float32))
                engine.set(_lib.BN_MOVING_MEAN, l, np.zeros(self.num_units, dtype=np.float32))
                engine.set(_lib.BN_MOVING_VAR, l, np.ones(self.num_units, dtype=np.float32))
        engine.set(_lib.WEIGHTS, self.num_layers, out.initial_weights(self.num_units, rng))
        engine.set(_lib.BIASES, self.num_layers, np.zeros(self.output_dim, dtype=np.float32))
        engine.set_scalar(_lib.INITIALISED_LAYERS, 0)
