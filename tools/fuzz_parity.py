"""GPU tool: randomised parity sweep -- random shapes / activation chains / micro-batch sizes, engine vs the float64
oracle (loss, every gradient, loss after the optimiser step, evaluation loss).  usage: fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import batch, engine_grads, make_pair  # noqa: E402
from tfkaldi_amd import _lib  # noqa: E402


def rel(got, want):
    return float(np.linalg.norm(np.asarray(got, dtype=np.float64) - want) / max(np.linalg.norm(want), 1e-12))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(cases):
        kw = dict(input_dim=int(rng.integers(1, 90)), num_layers=int(rng.integers(1, 5)),
                  num_units=int(rng.integers(1, 150)), output_dim=int(rng.integers(2, 120)),
                  nonlin=str(rng.choice(["relu", "sigmoid", "tanh", "linear"])), batch_norm=bool(rng.integers(0, 2)),
                  l2_norm=bool(rng.integers(0, 4) == 0), keep_prob=float(rng.choice([1.0, 1.0, 0.8, 0.5])),
                  layerwise_init=bool(rng.integers(0, 4) == 0), init_learning_rate=1e-3, num_steps=50)
        if kw["nonlin"] == "relu" and not kw["batch_norm"]:
            kw["nonlin"] = "tanh"  # (ReLU kinks at exact zeros of an un-normalised net make fp32/fp64 disagree)
        sizes = [int(rng.integers(2, 400)) for _ in range(int(rng.integers(1, 4)))]
        if rng.integers(0, 6) == 0:
            sizes = [int(rng.integers(2100, 3000))]  # long contraction: split-K weight gradients
        dtype = os.environ.get("TFK_FUZZ_DTYPE", "float32")  # bfloat16: mixed precision vs the operand-rounding oracle
        tol = 2e-3 if dtype == "float32" else 2e-2
        eng, oracle = make_pair(np.random.default_rng(case), max_frames=max(sizes), compute_dtype=dtype, **kw)
        bad = []
        for T in sizes:
            X, y = batch(rng, T, kw["input_dim"], kw["output_dim"])
            eng.accumulate(X, y)
            masks = None
            if kw["keep_prob"] < 1:
                masks = [eng.debug_fetch(_lib.DBG_DROPOUT_MASK, l, T).astype(np.float64) for l in range(eng.L)]
            oracle.accumulate(X, y, masks)
        lo = abs(eng.scalar(_lib.BATCH_LOSS) - oracle.batch_loss) / abs(oracle.batch_loss)
        if lo > (5e-5 if dtype == "float32" else 2e-3):
            bad.append(("loss", lo))
        got = engine_grads(eng)
        for k, want in oracle.G.items():
            if oracle.bn and k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
                continue
            # a LINEAR batch-normalised net: a shift of layer l's output is a column shift of z_{l+1}, which the next
            # batch norm removes -- d loss / d beta_l is identically zero below the last hidden layer and both sides
            # hold round-off only
            if oracle.bn and kw["nonlin"] == "linear" and not kw["l2_norm"] and k.startswith("beta") and \
                    k != "beta%d" % (oracle.L - 1):
                continue
            # a batch-normalised layer fed by ONE input: its output is invariant to the scale of its (single-row)
            # weight, so that gradient is identically zero as well
            if oracle.bn and k.startswith("W") and k != "W%d" % oracle.L and want.shape[0] == 1:
                continue
            if np.abs(want).max() < 1e-9:
                if np.abs(got[k]).max() > 1e-4:
                    bad.append((k, float(np.abs(got[k]).max())))
                continue
            r = rel(got[k], want)
            worst = max(worst, r)
            if r > tol:
                bad.append((k, r))
        la, lb = eng.apply(), oracle.apply()
        if abs(la - lb) > (5e-5 if dtype == "float32" else 2e-3) * abs(lb):
            bad.append(("avg loss", la, lb))
        X, y = batch(rng, sizes[0], kw["input_dim"], kw["output_dim"])
        eng.eval_accumulate(X, y); oracle.eval_accumulate(X, y)
        ea, eb = eng.eval_finish(), oracle.eval_finish()
        if abs(ea - eb) > (5e-4 if dtype == "float32" else 5e-3) * abs(eb):
            bad.append(("eval", ea, eb))
        eng.close()
        print("case %3d %s sizes %s: %s" % (case, {k: v for k, v in kw.items() if k not in ("init_learning_rate", "num_steps")},
                                              sizes, "OK" if not bad else "FAIL %s" % bad))
    print("worst gradient relative error %.2e" % worst)


if __name__ == "__main__":
    main()
