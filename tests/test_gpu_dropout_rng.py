"""The dropout keep masks (reference activation.py:140-141: tf.nn.dropout draws a fresh mask in every session.run) as
the engine regenerates them from Philox(key = seed, counter = (column / 4, row, layer, call)): independent across
layers, across accumulate calls and across data-parallel ranks (dataparallel.rank_seed), reproducible for equal
coordinates, and unstructured along rows and columns."""
import numpy as np
import pytest

from util import batch

pytestmark = pytest.mark.gpu


def _masks(eng, X, y, calls, T):
    from tfkaldi_amd import _lib
    out = []
    for _ in range(calls):
        eng.accumulate(X, y)
        out.append([eng.debug_fetch(_lib.DBG_DROPOUT_MASK, l, T) for l in range(eng.L)])
    return out


def test_dropout_masks_are_independent(gpu):
    from tfkaldi_amd import _lib
    from tfkaldi_amd.dataparallel import rank_seed
    from tfkaldi_amd.engine import Engine
    T, F, L, H, O, keep = 512, 24, 3, 256, 10, 0.5
    rng = np.random.default_rng(5)
    X, y = batch(rng, T, F, O)

    def engine(seed):
        eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=keep, max_frames=T, seed=seed))
        eng.init_hidden_weights(np.random.default_rng(1))
        return eng

    seed = 0x1234ABCD
    assert rank_seed(seed, 0) == seed and len({rank_seed(seed, r) for r in range(8)}) == 8
    a, a2, b = engine(seed), engine(seed), engine(rank_seed(seed, 1))
    ma, ma2, mb = _masks(a, X, y, 2, T), _masks(a2, X, y, 2, T), _masks(b, X, y, 2, T)
    n = T * H
    tol = 5.0 * 0.5 / np.sqrt(n)  # five sigma of the agreement rate of two independent fair masks
    for call in range(2):
        for l in range(L):
            m = ma[call][l]
            assert set(np.unique(m)) == {0.0, 1.0} and abs(m.mean() - keep) < tol
            np.testing.assert_array_equal(m, ma2[call][l])  # same coordinates -> same mask (backward regenerates it)
            # rows and columns: every row / column keeps about half of its units, neighbours are uncorrelated
            assert np.abs(m.mean(axis=0) - keep).max() < 6.0 * 0.5 / np.sqrt(T)
            assert np.abs(m.mean(axis=1) - keep).max() < 6.0 * 0.5 / np.sqrt(H)
            assert abs((m[:, 1:] == m[:, :-1]).mean() - 0.5) < tol and abs((m[1:] == m[:-1]).mean() - 0.5) < tol
    agree = lambda p, q: float((p == q).mean())
    pairs = {"layers": (ma[0][0], ma[0][1]), "layers 1-2": (ma[0][1], ma[0][2]), "calls": (ma[0][0], ma[1][0]),
             "ranks": (ma[0][0], mb[0][0]), "ranks, later call": (ma[1][2], mb[1][2]),
             "rank x layer": (ma[0][0], mb[0][1])}
    for name, (p, q) in pairs.items():
        assert abs(agree(p, q) - 0.5) < tol, (name, agree(p, q))
    for e in (a, a2, b):
        e.close()
