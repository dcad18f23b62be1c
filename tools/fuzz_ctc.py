"""GPU tool: randomised CTC sweep -- random utterance / label lengths (incl. empty, repeated and infeasible label
sequences), class counts and nets; loss, dLogits and the parameter gradients against oracle/ctc_oracle.py.
usage: fuzz_ctc.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.ctc_oracle import ctc_batch  # noqa: E402
from util import engine_grads, make_pair  # noqa: E402
from tfkaldi_amd import _lib  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(cases):
        O = int(rng.integers(2, 60))
        kw = dict(input_dim=int(rng.integers(1, 40)), num_layers=int(rng.integers(1, 4)), num_units=int(rng.integers(2, 70)),
                  output_dim=O, nonlin=str(rng.choice(["sigmoid", "tanh"])), batch_norm=bool(rng.integers(0, 2)),
                  init_learning_rate=1e-3, num_steps=50)
        U = int(rng.integers(1, 6))
        utt = [int(rng.integers(1, 90)) for _ in range(U)]
        lab = [int(rng.integers(0, max(1, min(t + 3, 40)))) for t in utt]  # some longer than their utterance
        hi = max(1, O - 1) if rng.integers(0, 3) else max(1, min(2, O - 1))  # few classes -> many repeats
        labels = np.concatenate([rng.integers(0, hi, size=n) for n in lab] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
        T = int(sum(utt))
        eng, oracle = make_pair(np.random.default_rng(case), max_frames=T, **kw)
        X = (rng.standard_normal((T, kw["input_dim"])) * 1.5).astype(np.float32)
        eng.accumulate_ctc(X, utt, labels, lab)
        loss, dlog, n_labels = ctc_batch(oracle.forward_logits(X), utt, labels, lab)
        got_loss = eng.scalar(_lib.BATCH_LOSS)
        bad = []
        if np.isfinite(loss):
            if abs(got_loss - loss) > 5e-5 * max(abs(loss), 1.0):
                bad.append(("loss", got_loss, loss))
        elif got_loss != np.inf:
            bad.append(("loss should be inf", got_loss))
        d = eng.debug_fetch(_lib.DBG_LOGITS, 0, T)
        err = float(np.abs(d - dlog).max())
        worst = max(worst, err)
        if err > 5e-5 or np.isnan(d).any():
            bad.append(("dlogits", err))
        if np.isfinite(loss):
            oracle.backward_from_dlogits(dlog, loss, n_labels)
            got = engine_grads(eng)
            for k, want in oracle.G.items():
                if oracle.bn and k.startswith("b") and not k.startswith("beta") and k != "b%d" % oracle.L:
                    continue
                scale = max(np.abs(want).max(), 1e-3)
                if np.abs(got[k] - want).max() > 5e-4 * scale:
                    bad.append((k, float(np.abs(got[k] - want).max() / scale)))
        eng.close()
        print("case %3d O=%d utt=%s lab=%s: %s" % (case, O, utt, lab, "OK" if not bad else "FAIL %s" % bad))
    print("worst dlogits error %.2e" % worst)


if __name__ == "__main__":
    main()
