"""Activation wrappers with the reference's constructors (neuralNetworks/classifiers/activation.py).

In the reference these objects append TensorFlow ops to a graph when called.  Here they DESCRIBE the
chain; `chain_spec()` flattens it and the HIP engine executes it fused with the affine layers
(tfkaldi_amd/csrc/kernels.hip).  A wrapper applies the wrapped activation first and its own function
last (reference activation.py:22-42), so nnet.py:42-72 builds
    affine -> Batchnorm -> nonlinearity -> L2Norm -> Dropout.
"""
from abc import ABCMeta, abstractmethod

import numpy as np

_ENGINE_ONLY = ("activation objects describe the network; the computation runs inside the HIP engine "
                "(create a Trainer / Decoder or call the DNN classifier)")


class Activation(object, metaclass=ABCMeta):
    """base class: wraps another activation (or None)"""

    def __init__(self, activation=None):
        self.activation = activation

    def __call__(self, inputs, is_training=False, reuse=False):
        raise TypeError(_ENGINE_ONLY)

    @abstractmethod
    def _apply_func(self, activations, is_training, reuse):
        """own stage of the chain"""
        raise NotImplementedError("Abstract method")

    @abstractmethod
    def _stage(self):
        """(stage name, parameter)"""

    def chain_spec(self):
        """innermost-first list of (stage, parameter) tuples"""
        inner = self.activation.chain_spec() if self.activation is not None else []
        return inner + [self._stage()]


def _classify_nonlinearity(fn):
    """Accepts 'relu' / 'sigmoid' / 'tanh' / 'linear' or a callable (the reference passes tf.nn.relu,
    tf.nn.sigmoid, tf.nn.tanh or an identity lambda: nnet.py:48-62); callables are recognised by probing."""
    if isinstance(fn, str):
        if fn not in ("relu", "sigmoid", "tanh", "linear"):
            raise Exception('unkown nonlinearity')
        return fn
    probe = np.array([-2.0, -0.5, 0.0, 0.75, 3.0])
    try:
        out = np.asarray(fn(probe), dtype=np.float64)
    except Exception as exc:
        raise TypeError("cannot recognise the nonlinearity %r: %s" % (fn, exc))
    table = {"relu": np.maximum(probe, 0), "sigmoid": 1 / (1 + np.exp(-probe)), "tanh": np.tanh(probe),
             "linear": probe}
    for name, want in table.items():
        if out.shape == want.shape and np.allclose(out, want, atol=1e-6):
            return name
    raise Exception('unkown nonlinearity')


class TfActivation(Activation):
    """adds an element-wise nonlinearity (reference activation.py:58-84)"""

    def __init__(self, activation, tfActivation):
        super(TfActivation, self).__init__(activation)
        self.tf_activation = tfActivation
        self.nonlin = _classify_nonlinearity(tfActivation)

    def _apply_func(self, activations, is_training, reuse):
        raise TypeError(_ENGINE_ONLY)

    def _stage(self):
        return ("nonlin", self.nonlin)


class L2Norm(Activation):
    """divides each frame by its mean square where that exceeds 1 (reference activation.py:87-111)"""

    def _apply_func(self, activations, is_training, reuse):
        raise TypeError(_ENGINE_ONLY)

    def _stage(self):
        return ("l2_norm", None)


class Dropout(Activation):
    """dropout in training mode; `dropout` is the KEEP probability in (0, 1] (reference activation.py:113-143)"""

    def __init__(self, activation, dropout):
        super(Dropout, self).__init__(activation)
        assert dropout > 0 and dropout <= 1
        self.dropout = dropout

    def _apply_func(self, activations, is_training, reuse):
        raise TypeError(_ENGINE_ONLY)

    def _stage(self):
        return ("dropout", float(self.dropout))


class Batchnorm(Activation):
    """batch normalisation, tf.contrib.layers.batch_norm defaults (reference activation.py:145-161)"""

    def _apply_func(self, activations, is_training, reuse):
        raise TypeError(_ENGINE_ONLY)

    def _stage(self):
        return ("batch_norm", None)


def engine_options(activation):
    """Map an activation chain onto the engine's fused hidden-layer pipeline
    (batch_norm -> nonlin -> l2_norm -> dropout, each optional but in this order)."""
    opts = {"batch_norm": False, "nonlin": "linear", "l2_norm": False, "keep_prob": 1.0}
    order = {"batch_norm": 0, "nonlin": 1, "l2_norm": 2, "dropout": 3}
    last = -1
    spec = activation.chain_spec() if activation is not None else []
    for stage, param in spec:
        if order[stage] <= last:
            raise NotImplementedError(
                "activation chain %s is not the batch_norm -> nonlin -> l2_norm -> dropout order the "
                "engine fuses" % [s for s, _ in spec])
        last = order[stage]
        if stage == "batch_norm":
            opts["batch_norm"] = True
        elif stage == "nonlin":
            opts["nonlin"] = param
        elif stage == "l2_norm":
            opts["l2_norm"] = True
        else:
            opts["keep_prob"] = param
    return opts
