"""GPU tool (diagnostic): why two correct fp32 GEMM tile configurations give gradients that differ in whole columns.
One micro-batch through a 2 x 2048 ReLU + BN net with the forward GEMM forced to the 128x128 tile (TFK_GEMM_CFG_NN=12: one
accumulator chain per element) or left at the 64x64 tile (four chains summed at the end): z, the batch statistics and the
layer outputs agree with float64 to round-off in both -- and yet a hidden layer's gradient moves by ~1 % in ONE column:
a batch-normalised value within round-off of zero lands on the other side of the ReLU, and d relu flips for that frame.
This is the freedom any fp32 implementation has (the reference's TensorFlow kernels included); it is what bounds the
stacked-vs-sequential comparison at full size (tests/test_gpu_stacked.py).

    TFK_GEMM_CFG_NN=12 python tools/relu_flip_probe.py ; python tools/relu_flip_probe.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import batch, make_pair
kw = dict(input_dim=440, num_layers=2, num_units=2048, output_dim=512, nonlin="relu", batch_norm=True,
          init_learning_rate=1e-3, num_steps=10, max_frames=1024)
rng = np.random.default_rng(1)
eng, oracle = make_pair(rng, **kw)
X, y = batch(rng, 1024, 440, 512)
eng.accumulate(X, y, last=True)
# float64 forward of layer 0 by hand
z0 = X.astype(np.float64) @ oracle.W[0] + oracle.b[0]
mu, var = z0.mean(0), z0.var(0)
rstd = 1.0 / np.sqrt(var + 1e-3)
gz = eng.debug_fetch(3, 0, 1024); gm = eng.debug_fetch(4, 0, 1024)[0]; gr = eng.debug_fetch(5, 0, 1024)[0]
print("z0 max err %.3e (max|z| %.3g)  rows with err>1e-4: %d" % (np.abs(gz - z0).max(), np.abs(z0).max(), (np.abs(gz - z0).max(1) > 1e-4).sum()))
print("mean0 err %.3e  rstd0 rel err %.3e  (rstd assumes eps 1e-3: min rel %.3e)" % (np.abs(gm - mu).max(), (np.abs(gr - rstd) / rstd).max(), (np.abs(gr - rstd) / rstd).min()))
bad = np.argwhere(np.abs(gz - z0) > 1e-4)
print("bad z elements:", len(bad), bad[:10].tolist())
from util import engine_grads
oracle.accumulate(X, y)
g = engine_grads(eng)
for name in ("W0", "W1", "W2", "beta0", "beta1"):
    w = oracle.G[name]
    print("%-6s vs f64 %.2e" % (name, np.abs(g[name] - w).max() / np.abs(w).max()))
a0 = eng.debug_fetch(1, 0, 1024); a1 = eng.debug_fetch(1, 1, 1024)
a0r = np.maximum((z0 - mu) * rstd + oracle.beta[0], 0)
print("a0 err %.3e" % np.abs(a0 - a0r).max())
z1 = a0r @ oracle.W[1] + oracle.b[1]
gz1 = eng.debug_fetch(3, 1, 1024)
print("z1 err %.3e  mean1 err %.3e rstd1 relerr %.3e" % (np.abs(gz1 - z1).max(), np.abs(eng.debug_fetch(4, 1, 1024)[0] - z1.mean(0)).max(),
      (np.abs(eng.debug_fetch(5, 1, 1024)[0] - 1 / np.sqrt(z1.var(0) + 1e-3)) * np.sqrt(z1.var(0) + 1e-3)).max()))
from tfkaldi_amd import _lib
for l in range(3):
    w = eng.get(_lib.WEIGHTS, l)
    print("W%d changed by accumulate: %d elements (max %.3e)" % (l, (w != oracle.W[l].astype(np.float32)).sum(), np.abs(w - oracle.W[l]).max()))
for l in range(2):
    print("beta%d changed: %d; m-slot nonzero: %d" % (l, (eng.get(_lib.BN_BETA, l) != oracle.beta[l].astype(np.float32)).sum(),
          (eng.get(_lib.WEIGHTS, l, _lib.SLOT_ADAM_M) != 0).sum()))
gb = g["beta1"]; wb = oracle.G["beta1"]
bad = np.flatnonzero(np.abs(gb - wb) > 1e-4 * np.abs(wb).max())
print("beta1 bad columns: %d of %d; first %s ... last %s" % (len(bad), wb.size, bad[:24].tolist(), bad[-8:].tolist()))
print("  bad mod 128 histogram:", np.bincount(bad % 128, minlength=128).nonzero()[0][:40].tolist())
gw = g["W1"]; ww = oracle.G["W1"]
badw = np.argwhere(np.abs(gw - ww) > 1e-4 * np.abs(ww).max())
print("W1 bad elements: %d; bad columns %d bad rows %d; cols first %s" % (len(badw), len(set(badw[:,1])), len(set(badw[:,0])), sorted(set(badw[:,1]))[:16]))
