"""GPU tool: time the CTC training step on a BASELINE configs[4]-like workload (4x512 DNN, 440 in, 35 characters +
blank, 16 utterances x 800 frames, 100 labels each) and print the per-kernel-family HIP-event profile."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def main():
    U, Tu, S, F, L, H, O = 16, 800, 100, 440, 4, 512, 36
    T = U * Tu
    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, max_frames=T, num_steps=1000,
                           compute_dtype=os.environ.get("TFK_QB_DTYPE", "float32"))
    eng = Engine(cfg)
    rng = np.random.default_rng(7)
    eng.init_hidden_weights(rng)
    X = rng.standard_normal((T, F)).astype(np.float32)
    raw = rng.standard_normal((T, F // 11)).astype(np.float32)  # 40-dim unspliced frames, context 5
    labels = rng.integers(0, O - 1, size=U * S).astype(np.int32)
    utt, lab = [Tu] * U, [S] * U
    K = 10
    for name, step in (("host-spliced frames, 22 MB over PCIe", lambda: eng.accumulate_ctc(X, utt, labels, lab, last=True)),
                       ("unspliced frames, splice on the device",
                        lambda: eng.accumulate_ctc_raw(raw, utt, 5, labels, lab, last=True))):
        for _ in range(3):
            step()
            eng.apply()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
            loss = eng.apply()
        dt = (time.perf_counter() - t0) / K
        print("ctc step (%s): %.3f ms  %.0f frames/s  loss/label %.4f" % (name, dt * 1e3, T / dt, loss))
    eng.profile_begin()
    for _ in range(K):
        eng.accumulate_ctc_raw(raw, utt, 5, labels, lab, last=True)
        eng.apply()
    for s in eng.profile_end():
        print("  %-28s n=%4d %9.3f ms/step" % (s["name"], s["launches"] // K, s["total_ms"] / K))
    eng.close()


if __name__ == "__main__":
    main()
