"""Stacked passes (include/tfkaldi_hip.h: tfk_accumulate_stacked / _raw): k micro-batches of one optimiser step in ONE
pass of the GEMMs must give what k sequential tfk_accumulate calls give (reference neuralNetworks/trainer.py:310-332: the
micro-batch loop; classifiers/activation.py:159-161: batch-norm statistics per session run = per micro-batch) -- against
the float64 oracle fed micro-batch by micro-batch, and against the engine's own sequential path."""
import os

import numpy as np
import pytest

from util import assert_close, batch, engine_grads, engine_params, make_pair

pytestmark = pytest.mark.gpu

KW = dict(input_dim=40, num_layers=3, num_units=72, output_dim=24, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-3, num_steps=10, max_frames=64)


def _segments(rng, rows, F, O):
    parts = [batch(rng, n, F, O) for n in rows]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), parts


@pytest.mark.parametrize("rows", [[128, 128], [100, 37, 260, 5], [64] * 8, [300, 129, 1]])
@pytest.mark.parametrize("dtype,keep", [("float32", 1.0), ("float32", 0.7), ("bfloat16", 1.0)])
def test_stacked_matches_oracle_and_sequential(gpu, rows, dtype, keep):
    """ragged segments (padding rows inside the pass), aligned segments (no padding), single-row segments; gradient sums,
    loss, BN moving averages (sequential per-segment updates) and the parameters after Adam"""
    rng = np.random.default_rng(11)
    kw = dict(KW, keep_prob=keep, compute_dtype=dtype, seed=77)
    eng, oracle = make_pair(rng, **kw)
    seq, _ = make_pair(np.random.default_rng(11), **kw)
    X, y, parts = _segments(rng, rows, kw["input_dim"], kw["output_dim"])
    for step in range(2):
        eng.accumulate_stacked(X, y, rows, last=True)
        for i, (Xs, ys) in enumerate(parts):
            seq.accumulate(Xs, ys, last=(i == len(parts) - 1))
        if keep >= 1.0:
            for Xs, ys in parts:
                oracle.accumulate(Xs, ys)
        g_stk, g_seq = engine_grads(eng), engine_grads(seq)
        # same kernels, same per-segment statistics: only the summation order inside dW / the loss differs.  Scale: the
        # largest gradient of the LAYER (the bias gradient in front of a batch norm is a sum of dz that is zero in exact
        # arithmetic: pure round-off, to be judged on the scale of the layer's weight gradient)
        # Step 0 starts from identical parameters: tight.  From step 1 on the two engines' parameters differ where Adam took
        # its first step on a round-off-sized gradient the other way (~lr * sign(g): tests/test_gpu_engine_parity.py), and a
        # pre-activation within round-off of zero may sit on the other side of the ReLU: bounded loosely.
        for k in g_seq:
            layer = "".join(ch for ch in k if ch.isdigit())
            scale = max(np.abs(v).max() for kk, v in g_seq.items() if kk.endswith(layer)) + 1e-30
            tol = (2e-6 if dtype == "float32" else 2e-2) if step == 0 else 5e-2
            assert np.abs(g_stk[k] - g_seq[k]).max() <= tol * scale, (step, k, np.abs(g_stk[k] - g_seq[k]).max() / scale)
        if keep >= 1.0 and dtype == "float32" and step == 0:
            for k, want in oracle.G.items():
                layer = "".join(ch for ch in k if ch.isdigit())
                scale = max(np.abs(v).max() for kk, v in oracle.G.items() if kk.endswith(layer))
                assert_close("G %s step %d" % (k, step), g_stk[k], want, 2e-4, 2e-5 * scale)
        l_stk, l_seq = eng.apply(), seq.apply()
        assert abs(l_stk - l_seq) <= ((2e-6 if dtype == "float32" else 2e-3) if step == 0 else 5e-3) * abs(l_seq), (
            step, l_stk, l_seq)
        if keep >= 1.0 and dtype == "float32":
            assert_close("loss %d" % step, l_stk, oracle.apply(), 2e-5 if step == 0 else 1e-3, 0)
    from tfkaldi_amd import _lib
    for l in range(kw["num_layers"]):  # moving averages: k sequential updates, in segment order
        for kind in (_lib.BN_MOVING_MEAN, _lib.BN_MOVING_VAR):
            a, b = eng.get(kind, l), seq.get(kind, l)
            assert np.abs(a - b).max() <= (1e-6 if dtype == "float32" else 1e-3) * (np.abs(b).max() + 1e-30), (l, kind)
    if keep >= 1.0 and dtype == "float32":
        for l in range(kw["num_layers"]):
            assert_close("mov_mean", eng.get(_lib.BN_MOVING_MEAN, l), oracle.mov_mean[l], 1e-4, 1e-6)
            assert_close("mov_var", eng.get(_lib.BN_MOVING_VAR, l), oracle.mov_var[l], 1e-4, 1e-6)
    p_stk, p_seq = engine_params(eng), engine_params(seq)
    for k in p_seq:
        assert np.abs(p_stk[k] - p_seq[k]).max() <= (5e-4 if dtype == "float32" else 5e-3), k
    eng.close(); seq.close()


def test_dropout_stream_is_the_sequential_one(gpu):
    """the keep mask of segment i of a stacked pass is the mask the i-th sequential call draws (call index + row inside the
    micro-batch): with aligned segments and fp32 the layer outputs are bit-identical"""
    from tfkaldi_amd import _lib
    rng = np.random.default_rng(5)
    kw = dict(KW, keep_prob=0.6, seed=9)
    eng, _ = make_pair(rng, **kw)
    seq, _ = make_pair(np.random.default_rng(5), **kw)
    rows = [128, 128, 128]
    X, y, parts = _segments(rng, rows, kw["input_dim"], kw["output_dim"])
    eng.accumulate_stacked(X, y, rows)
    stacked = eng.debug_fetch(_lib.DBG_HIDDEN, 1, sum(rows))
    for i, (Xs, ys) in enumerate(parts):
        seq.accumulate(Xs, ys)
        want = seq.debug_fetch(_lib.DBG_HIDDEN, 1, rows[i])
        got = stacked[sum(rows[:i]):sum(rows[:i + 1])]
        assert (want == 0).mean() > 0.3  # (dropout really dropped)
        np.testing.assert_array_equal(got, want)
    eng.close(); seq.close()


def test_stacked_raw_equals_sequential_raw(gpu):
    """unspliced frames + utterance lengths (the packed feed's entry point): CMVN + splice on the device into the padded
    layout of the pass"""
    rng = np.random.default_rng(3)
    D, C = 8, 2
    kw = dict(KW, input_dim=D * (2 * C + 1), seed=1)
    eng, _ = make_pair(rng, **kw)
    seq, _ = make_pair(np.random.default_rng(3), **kw)
    lens = rng.integers(5, 40, size=11).astype(np.int32)
    seg_utts = [4, 3, 4]
    T = int(lens.sum())
    raw = rng.standard_normal((T, D)).astype(np.float32)
    y = rng.integers(0, kw["output_dim"], size=T).astype(np.int32)
    cmvn = np.stack([np.stack([rng.standard_normal(D), 0.5 + rng.random(D)]) for _ in lens]).astype(np.float32)
    eng.accumulate_stacked_raw(raw, y, lens, C, seg_utts, last=True, cmvn=cmvn)
    u = r = 0
    for i, n in enumerate(seg_utts):
        rows = int(lens[u:u + n].sum())
        seq.accumulate_raw(raw[r:r + rows], y[r:r + rows], lens[u:u + n], C, last=(i == len(seg_utts) - 1), cmvn=cmvn[u:u + n])
        u, r = u + n, r + rows
    g_stk, g_seq = engine_grads(eng), engine_grads(seq)
    scale = max(np.abs(v).max() for v in g_seq.values())
    for k in g_seq:
        assert np.abs(g_stk[k] - g_seq[k]).max() <= 2e-6 * scale, k
    assert abs(eng.apply() - seq.apply()) <= 2e-6
    eng.close(); seq.close()


def test_chains_outside_the_stacked_pass_run_sequentially(gpu):
    """tanh chains, no batch norm, TFK_STACK=0 (the switch is read at engine creation): the call is still valid and equals
    the sequential calls BIT FOR BIT (it is them)"""
    rng = np.random.default_rng(8)
    for over in (dict(nonlin="tanh"), dict(batch_norm=False), dict(l2_norm=True)):
        kw = dict(KW, **over)
        eng, _ = make_pair(np.random.default_rng(1), **kw)
        seq, _ = make_pair(np.random.default_rng(1), **kw)
        rows = [70, 130, 9]
        X, y, parts = _segments(rng, rows, kw["input_dim"], kw["output_dim"])
        eng.accumulate_stacked(X, y, rows, last=True)
        for i, (Xs, ys) in enumerate(parts):
            seq.accumulate(Xs, ys, last=(i == 2))
        for k, v in engine_grads(seq).items():
            np.testing.assert_array_equal(engine_grads(eng)[k], v, err_msg=str((over, k)))
        assert eng.apply() == seq.apply()
        eng.close(); seq.close()
    with pytest.raises(Exception, match="hold 3 rows"):
        eng2, _ = make_pair(np.random.default_rng(1), **KW)
        try:
            eng2.accumulate_stacked(X[:4], y[:4], [1, 2], last=True)
        finally:
            eng2.close()


@pytest.mark.timeout(600)
def test_stacked_at_cfg2_size_against_sequential(gpu):
    """8 micro-batches x 1024 frames on BASELINE cfg2's network (the shape tools/multi_mb_bench.py times), fp32 and bf16"""
    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    for dtype, tol in (("float32", 1e-5), ("bfloat16", 3e-2)):
        engines = []
        for _ in range(2):
            eng = Engine(_lib.make_config(440, 6, 2048, 2000, nonlin="relu", batch_norm=True, max_frames=1024, num_steps=100,
                                          compute_dtype=dtype))
            eng.init_hidden_weights(np.random.default_rng(7))
            rng = np.random.default_rng(2)
            eng.set(_lib.WEIGHTS, 6, (rng.standard_normal((2048, 2000)) * 0.02).astype(np.float32))
            engines.append(eng)
        g = torch.Generator(device="cuda").manual_seed(1)
        X = torch.randn(8192, 440, device="cuda", generator=g)
        y = torch.randint(0, 2000, (8192,), device="cuda", dtype=torch.int32, generator=g)
        torch.cuda.synchronize()
        stk, seq = engines
        stk.accumulate_stacked_device(X.data_ptr(), 440, y.data_ptr(), 8192, [1024] * 8, last=True)
        for i in range(8):
            seq.accumulate_device(X[i * 1024:].data_ptr(), 440, y[i * 1024:].data_ptr(), 1024, last=(i == 7))
        # The stacked pass multiplies 8192 rows at once, so the heuristic takes 128x128 tiles (one accumulator chain per
        # element) where the 1024-row passes take 64x64 tiles (four chains, summed at the end): z differs in the last
        # bit, and of the 10^8 batch-normalised values a few dozen lie so close to zero that they land on the other side of
        # the ReLU -- each moves ONE column of the layer's gradient by one frame's contribution (measured with the float64
        # oracle as referee: tools/relu_flip_probe.py; the reference's TensorFlow kernels have the same freedom).  So: the
        # output layer (no ReLU behind it) tight; hidden layers by the share of columns that moved at all, and in norm.
        a, b = stk.get(_lib.WEIGHTS, 6, _lib.SLOT_GRAD), seq.get(_lib.WEIGHTS, 6, _lib.SLOT_GRAD)
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (dtype, np.abs(a - b).max() / np.abs(b).max())
        for l in (0, 3, 5):
            a, b = stk.get(_lib.WEIGHTS, l, _lib.SLOT_GRAD), seq.get(_lib.WEIGHTS, l, _lib.SLOT_GRAD)
            moved = (np.abs(a - b).max(axis=0) > 10 * tol * np.abs(b).max()).mean()
            rel = np.linalg.norm(a - b) / np.linalg.norm(b)
            # (a flip in layer l moves one column there and, through dA, a little of EVERY column below: the share of
            # moved columns is a meaningful bound for the top hidden layer only; measured rel: 0.8e-3 .. 2.4e-3 in fp32)
            assert rel <= (8e-3 if dtype == "float32" else 1e-1), (dtype, l, moved, rel)  # (bf16: measured up to 4.2e-2)
            if l == 5 and dtype == "float32":
                assert moved <= 0.10, (l, moved)
        l_stk, l_seq = stk.apply(), seq.apply()
        assert abs(l_stk - l_seq) <= tol * abs(l_seq)
        assert np.abs(stk.get(_lib.BN_MOVING_VAR, 4) - seq.get(_lib.BN_MOVING_VAR, 4)).max() <= 1e-5 + tol
        stk.close(); seq.close()


@pytest.mark.parametrize("kw_over", [{}, {"nonlin": "tanh", "l2_norm": True}, {"keep_prob": 0.7}, {"compute_dtype": "bfloat16"}])
def test_stacked_evaluation_equals_one_pass_per_microbatch(gpu, kw_over, monkeypatch):
    """Trainer.evaluate over k micro-batches in ONE engine call (tfk_eval_accumulate_stacked, reference trainer.py:356-441): in
    evaluation mode rows are independent, so the stack is one pass over the concatenation -- for every activation chain, ragged
    segments and single-row ones included.  Same frame count, the loss to the fp32 order of its sum; also when the stack is cut
    into several passes (TFK_EVAL_PASS_ROWS)."""
    from util import make_pair, batch
    kw = dict(input_dim=40, num_layers=3, num_units=96, output_dim=30, nonlin="relu", batch_norm=True, init_learning_rate=1e-3,
              num_steps=10)
    kw.update(kw_over)
    rng = np.random.default_rng(5)
    eng, _ = make_pair(rng, **kw)
    X0, y0 = batch(rng, 64, 40, 30)
    eng.accumulate(X0, y0, last=True)  # (a step, so that the moving statistics are not the initial ones)
    eng.apply()
    rows = [57, 1, 130, 256, 33]
    mbs = [batch(rng, r, 40, 30) for r in rows]
    for X, y in mbs:
        eng.eval_accumulate(X, y)
    frames_seq = eng.scalar(5)  # NUM_FRAMES
    want = eng.eval_finish()
    Xs, ys = np.concatenate([m[0] for m in mbs], 0), np.concatenate([m[1] for m in mbs], 0)
    tol = 2e-6 if kw.get("compute_dtype", "float32") != "bfloat16" else 2e-5
    for cap in (None, "200", "1"):
        if cap:
            monkeypatch.setenv("TFK_EVAL_PASS_ROWS", cap)
        eng.eval_accumulate_stacked(Xs, ys, rows)
        assert eng.scalar(5) == frames_seq == sum(rows)
        got = eng.eval_finish()
        assert abs(got - want) <= tol * abs(want), (cap, got, want)
    with pytest.raises(RuntimeError, match="hold"):
        eng.eval_accumulate_stacked(Xs, ys, rows[:-1])
    eng.close()
