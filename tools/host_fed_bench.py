"""GPU tool: BASELINE cfg2 step fed from HOST memory (the C ABI's host-pointer entry points: pinned double-buffered
staging + H2D on a copy stream), spliced frames vs unspliced frames with the splice on the device.  The numbers for
DESIGN.md's PCIe-inclusive note."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def main():
    T, D, C, L, H, O = 1024, 40, 5, 6, 2048, 2000
    F = D * (2 * C + 1)
    eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, max_frames=T, num_steps=1000))
    rng = np.random.default_rng(7)
    eng.init_hidden_weights(rng)
    X = rng.standard_normal((T, F)).astype(np.float32)
    raw = rng.standard_normal((T, D)).astype(np.float32)
    y = rng.integers(0, O, size=T).astype(np.int32)
    lens = [64] * 16
    for name, step in (("spliced [1024, 440] from host", lambda: eng.accumulate(X, y, last=True)),
                       ("unspliced [1024, 40] from host", lambda: eng.accumulate_raw(raw, y, lens, C, last=True))):
        for _ in range(5):
            step(); eng.apply()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            step(); eng.apply()
        dt = (time.perf_counter() - t0) / K
        print("%-32s %.3f ms/step  %.0f frames/s" % (name, dt * 1e3, T / dt))
    eng.close()


if __name__ == "__main__":
    main()
