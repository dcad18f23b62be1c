"""Sequential <-> flat frame layouts (reference neuralNetworks/classifiers/seq_convertors.py).

Sequential data is a list with one [batch, dim] array per time step (zero padded); non-sequential data is
one [T, dim] array, T = sum of the sequence lengths, utterance-major with the frame order kept.  The engine
consumes the flat layout directly, so these are host-side helpers only (pure data movement)."""
import numpy as np


def seq2nonseq(tensorlist, seq_length, name=None):
    """list of Tmax [U, F] arrays -> [T, F] (reference seq_convertors.py:12-39)"""
    stacked = np.stack(tensorlist)  # [Tmax, U, F]
    return np.concatenate([stacked[:int(n), s] for s, n in enumerate(seq_length)], axis=0)


def nonseq2seq(tensor, seq_length, length, name=None):
    """[T, F] -> list of `length` [U, F] arrays, zero padded (reference seq_convertors.py:41-80)"""
    seq_length = [int(n) for n in seq_length]
    out = np.zeros((length, len(seq_length), tensor.shape[1]), dtype=tensor.dtype)
    start = 0
    for s, n in enumerate(seq_length):
        out[:n, s] = tensor[start:start + n]
        start += n
    return [out[t] for t in range(length)]
