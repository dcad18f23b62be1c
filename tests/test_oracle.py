"""The float64 oracle itself: cross-checked against PyTorch autograd (an independent implementation of
the same graph) and against the known-answer properties of SURVEY.md section 8c."""
import numpy as np
import pytest
import torch

from oracle.dnn_oracle import OracleDNN
from util import randomize

KW = dict(input_dim=9, num_layers=3, num_units=11, output_dim=7, init_learning_rate=1e-3, num_steps=50)


def torch_loss(o, X, y, masks=None, train=True):
    """the same network in torch (float64); returns (summed CE, leaf tensors, batch stats)"""
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in o.params().items()}
    a = torch.tensor(np.asarray(X, dtype=np.float64))
    stats = []
    nact = o.num_active()
    nfw = o.L if (train and o.layerwise and o.bn) else nact
    outs = []
    for l in range(nfw):
        z = a @ P["W%d" % l] + P["b%d" % l]
        if o.bn:
            if train:
                mu, var = z.mean(0), z.var(0, unbiased=False)
            else:
                mu, var = torch.tensor(o.mov_mean[l]), torch.tensor(o.mov_var[l])
            stats.append((mu.detach().numpy(), var.detach().numpy()))
            z = (z - mu) / torch.sqrt(var + o.bn_eps) + P["beta%d" % l]
        v = {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "linear": lambda t: t}[o.nonlin](z)
        if o.l2:
            s = (v ** 2).mean(1, keepdim=True)
            v = torch.where(s > 1, v / s, v)
        if o.dropout and train:
            v = v * torch.tensor(masks[l]) / o.keep
        a = v
        outs.append(a)
    logits = outs[nact - 1] @ P["W%d" % o.L] + P["b%d" % o.L]
    loss = torch.nn.functional.cross_entropy(logits, torch.tensor(np.asarray(y, dtype=np.int64)), reduction="sum")
    return loss, P, stats, logits


CHAINS = [dict(nonlin="relu", batch_norm=True), dict(nonlin="sigmoid"), dict(nonlin="tanh", l2_norm=True),
          dict(nonlin="relu", l2_norm=True),
          dict(nonlin="relu", batch_norm=True, l2_norm=True, keep_prob=0.6), dict(nonlin="linear", keep_prob=0.8),
          dict(nonlin="relu", batch_norm=True, layerwise_init=True)]


@pytest.mark.parametrize("chain", CHAINS, ids=lambda c: "-".join("%s=%s" % kv for kv in sorted(c.items())))
def test_gradients_match_autograd(chain):
    rng = np.random.default_rng(0)
    o = OracleDNN(**dict(KW, **chain))
    randomize(o, rng, scale=3.0)  # large output weights so the L2 branch (s > 1) is exercised
    if o.l2:
        for l in range(o.L):
            o.W[l] = o.W[l] * 3
    T = 23
    X = rng.standard_normal((T, KW["input_dim"])) * 2
    y = rng.integers(0, KW["output_dim"], size=T)
    masks = [(rng.random((T, KW["num_units"])) < o.keep).astype(np.float64) for _ in range(o.L)] if o.dropout else None
    loss, P, stats, logits = torch_loss(o, X, y, masks)
    loss.backward()
    g = o.accumulate(X, y, masks)
    assert abs(o.batch_loss - loss.item()) < 1e-9 * max(1, abs(loss.item()))
    assert np.allclose(o.last_logits, logits.detach().numpy(), rtol=1e-10, atol=1e-12)
    if o.l2 and o.nonlin == "relu":  # tanh / sigmoid outputs never have a mean square above 1
        assert any((c["s"] > 1).any() for c in o.last_cache) and any((c["s"] <= 1).any() for c in o.last_cache)
    for k in g:
        want = P[k].grad.numpy() if P[k].grad is not None else np.zeros_like(g[k])
        assert np.allclose(g[k], want, rtol=1e-8, atol=1e-10), k
    if o.bn:  # EMA of the biased batch statistics, once per accumulate
        o2 = OracleDNN(**dict(KW, **chain))
        for l, (mu, var) in enumerate(stats):
            assert np.allclose(o.mov_mean[l] - 0.999 * np.zeros(1), o.mov_mean[l])
            assert mu.shape == (KW["num_units"],)


@pytest.mark.parametrize("chain", [c for c in CHAINS if c.get("nonlin") != "relu" or not c.get("l2_norm")],
                         ids=lambda c: "-".join("%s=%s" % kv for kv in sorted(c.items())))
def test_gradients_match_finite_differences(chain):
    """A third, torch-free pin of the backward pass: central differences of the oracle's OWN summed loss (batch statistics,
    L2 branch selection and dropout masks held as the forward pass holds them) against its analytic gradient sums.  Chains
    with a kink that a finite step can cross (ReLU next to the L2 threshold) are left to the autograd test."""
    import copy
    rng = np.random.default_rng(11)
    o = OracleDNN(**dict(KW, **chain))
    randomize(o, rng, scale=1.5)
    T = 17
    X = rng.standard_normal((T, KW["input_dim"]))
    y = rng.integers(0, KW["output_dim"], size=T)
    masks = [(rng.random((T, KW["num_units"])) < o.keep).astype(np.float64) for _ in range(o.L)] if o.dropout else None

    def loss_at(name, idx, delta):
        q = copy.deepcopy(o)
        q.params()[name][idx] += delta
        q.accumulate(X, y, masks)
        return q.batch_loss

    ref = copy.deepcopy(o)
    g = ref.accumulate(X, y, masks)
    h = 1e-5
    checked = kinks = 0
    for name, arr in o.params().items():
        if not np.any(g[name]):  # (layers above the active depth: the zero branch)
            continue
        for _ in range(4):
            idx = tuple(int(rng.integers(0, n)) for n in arr.shape)
            fd = (loss_at(name, idx, h) - loss_at(name, idx, -h)) / (2 * h)
            if o.nonlin == "relu" and abs(fd - g[name][idx]) > 1e-5 * max(1.0, abs(fd)):
                kinks += 1  # a ReLU kink inside the step: not a statement about the gradient (rare: bounded below)
                continue
            assert abs(fd - g[name][idx]) <= 2e-6 * max(1.0, abs(fd)), (name, idx, fd, g[name][idx])
            checked += 1
    assert checked >= 8 and kinks <= 2, (checked, kinks)


def test_adam_matches_torch_adam():
    """TF's Adam (epsilon outside the bias correction, folded lr_t) vs torch.optim.Adam, whose epsilon is added
    to sqrt(v_hat): equal when epsilon is rescaled, which pins the formula."""
    rng = np.random.default_rng(1)
    o = OracleDNN(**dict(KW, nonlin="relu"))
    randomize(o, rng)
    p0 = {k: v.copy() for k, v in o.params().items()}
    X = rng.standard_normal((15, KW["input_dim"]))
    y = rng.integers(0, KW["output_dim"], size=15)
    traj = []
    for _ in range(3):
        o.accumulate(X, y)
        traj.append({k: np.clip(v / 15.0, -1, 1) for k, v in o.G.items()})
        o.apply()
    for k in p0:
        m = np.zeros_like(p0[k]); v = np.zeros_like(p0[k]); w = p0[k].copy()
        for t, g in enumerate(traj, start=1):
            m = 0.9 * m + 0.1 * g[k]; v = 0.999 * v + 0.001 * g[k] ** 2
            mhat, vhat = m / (1 - 0.9 ** t), v / (1 - 0.999 ** t)
            w -= 1e-3 * mhat / (np.sqrt(vhat) + 1e-8 / np.sqrt(1 - 0.999 ** t))
        assert np.allclose(o.params()[k], w, rtol=1e-9, atol=1e-12), k


def test_known_answers():
    rng = np.random.default_rng(2)
    o = OracleDNN(**dict(KW, nonlin="relu", batch_norm=True))
    o.init_hidden_weights(rng)  # reference initialisation: output layer zero
    X = rng.standard_normal((20, KW["input_dim"]))
    y = rng.integers(0, KW["output_dim"], size=20)
    w0 = [w.copy() for w in o.W]
    o.accumulate(X, y)
    for l in range(o.L):  # KAT 8c-2: zero output weights => no gradient reaches the hidden layers
        assert (o.G["W%d" % l] == 0).all() and (o.G["beta%d" % l] == 0).all()
    loss = o.apply()
    assert abs(loss - np.log(KW["output_dim"])) < 1e-12  # KAT 8c-1
    for l in range(o.L):
        assert (o.W[l] == w0[l]).all()
    moved = np.abs(o.b[o.L])
    assert np.allclose(moved[moved > 0], 1e-3, rtol=1e-3)  # first Adam step ~ lr * sign(g)
    # KAT 8c-4: clipping saturates at +-1 before Adam
    o.G["b%d" % o.L] = np.full(KW["output_dim"], 1e6); o.num_frames = 10; o.batch_loss = 1.0
    b_before = o.b[o.L].copy(); m_before = o.m["b%d" % o.L].copy()
    o.apply()
    assert np.allclose(o.m["b%d" % o.L], 0.9 * m_before + 0.1 * 1.0)
    # KAT 8c-9: eval-mode BN at initialisation = z / sqrt(1 + 1e-3)
    o2 = OracleDNN(**dict(KW, nonlin="linear", batch_norm=True, num_layers=1))
    o2.init_hidden_weights(rng)
    _, cache, _ = o2._forward(X, False)
    assert np.allclose(cache[0]["a"], X.dot(o2.W[0]) / np.sqrt(1 + 1e-3))
    # learning-rate schedule: exponential decay over num_steps and halving
    o3 = OracleDNN(**dict(KW, learning_rate_decay=0.1))
    o3.global_step = 25
    assert np.isclose(o3.learning_rate(), 1e-3 * 0.1 ** 0.5)
    o3.halve_learning_rate()
    assert np.isclose(o3.learning_rate(), 0.5e-3 * 0.1 ** 0.5)


def test_layerwise_selects_depth():
    rng = np.random.default_rng(3)
    o = OracleDNN(**dict(KW, nonlin="relu", layerwise_init=True))
    randomize(o, rng)
    assert o.num_active() == 1
    X = rng.standard_normal((5, KW["input_dim"]))
    l1 = o.posteriors(X)
    o.add_layer(); assert o.num_active() == 2
    assert not np.allclose(o.posteriors(X), l1)
    o.add_layer(); o.add_layer(); o.add_layer(); assert o.num_active() == 3  # tf.case default branch
    o.init_last_layer()
    assert np.allclose(o.posteriors(X), 1.0 / KW["output_dim"])


def test_torch_cpu_baseline_matches_oracle():
    """bench.py's cpu_baseline stand-in computes the same step as the oracle (fp32 vs float64)"""
    from oracle.torch_cpu_step import TorchCpuTrainer
    rng = np.random.default_rng(5)
    kw = dict(input_dim=20, num_layers=2, num_units=16, output_dim=9)
    o = OracleDNN(nonlin="relu", batch_norm=True, init_learning_rate=1e-3, num_steps=100, **kw)
    ws = o.init_hidden_weights(rng)
    t = TorchCpuTrainer(nonlin="relu", batch_norm=True, init_learning_rate=1e-3, **kw)
    t.set_hidden_weights(ws)
    for step in range(4):
        for _ in range(2):
            X = rng.standard_normal((30, 20)).astype(np.float32)
            y = rng.integers(0, 9, size=30)
            o.accumulate(X, y); t.accumulate(X, y)
        assert abs(o.apply() - t.apply()) < 1e-4
    assert np.allclose(t.mov_var[1].numpy(), o.mov_var[1], rtol=1e-4, atol=1e-6)
