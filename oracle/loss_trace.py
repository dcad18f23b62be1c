"""CPU ORACLE leg (test / bench infrastructure, NOT product code): the per-step `average_loss` of the first optimiser
steps of a training run in FLOAT64, the referee between two fp32 implementations of the same step.

SURVEY.md 8d asks for CE-loss parity over the first 20 steps (reference neuralNetworks/trainer.py:336-346: the value
Trainer.update returns, batch_loss / num_frames before the parameter update).  Two fp32 implementations that sum in different
orders drift apart once Adam has amplified their rounding noise (|g| ~ sqrt(v) makes the first updates ~lr * sign(g)); their
mutual distance says nothing about which of them follows the specified arithmetic.  This module runs oracle/dnn_oracle.py
(float64 numpy, every function citing the reference line it restates) over the SAME weights and micro-batch sequence, so
that bench.py and tests/test_gpu_loss_trace.py can put the engine's and the CPU stand-in's distance to it side by side.
"""
import numpy as np

from .dnn_oracle import OracleDNN


def f64_loss_trace(batches, hidden_weights, steps, input_dim, num_layers, num_units, output_dim, nonlin="relu",
                   batch_norm=True, init_learning_rate=1e-3):
    """`steps` optimiser steps, one micro-batch each (batches[i % len(batches)]), from the reference initialisation with the
    given hidden weights (output layer zero, dnn.py:67-68); returns the list of average losses"""
    o = OracleDNN(input_dim, num_layers, num_units, output_dim, nonlin=nonlin, batch_norm=batch_norm,
                  init_learning_rate=init_learning_rate, learning_rate_decay=1.0, num_steps=max(1, 3 * steps))
    for l, w in enumerate(hidden_weights):
        o.W[l] = np.asarray(w, dtype=np.float64).copy()
    trace = []
    for i in range(steps):
        X, y = batches[i % len(batches)]
        o.accumulate(np.asarray(X, dtype=np.float64), np.asarray(y))
        trace.append(float(o.apply()))
    return trace


def distances(trace, referee):
    """per-step relative distance |trace - referee| / |referee| and its running maximum"""
    n = min(len(trace), len(referee))
    rel = [abs(a - b) / max(abs(b), 1e-300) for a, b in zip(trace[:n], referee[:n])]
    return rel, list(np.maximum.accumulate(rel)) if rel else []
