"""Feed-forward layer description (reference neuralNetworks/classifiers/layer.py)."""
import numpy as np


class FFLayer(object):
    """fully connected layer: outputs = activation(inputs . weights + biases) with weights [d_in, d_out]
    (reference layer.py:36-56).  The product is computed by the MFMA GEMM of the HIP engine."""

    def __init__(self, output_dim, activation, weights_std=None):
        self.output_dim = output_dim
        self.activation = activation
        self.weights_std = weights_std

    def initial_weights(self, input_dim, rng):
        """N(0, weights_std) or N(0, 1/sqrt(d_in)) when weights_std is None (reference layer.py:39-44)"""
        std = self.weights_std if self.weights_std is not None else 1.0 / np.sqrt(input_dim)
        return (rng.standard_normal((input_dim, self.output_dim)) * std).astype(np.float32)

    def __call__(self, inputs, is_training=False, reuse=False, scope=None):
        raise TypeError("layers describe the network; the computation runs inside the HIP engine")
