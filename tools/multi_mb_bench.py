"""GPU tool: the reference's own batch configuration on BASELINE cfg2's network -- 128 utterances per optimiser step in
8 micro-batches of 16 (config_AURORA4.cfg:134-137), here 8 x 1024 frames: G accumulates over the micro-batches
(EPI_ACCUM from the second on), one Adam per step."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def main():
    T, F, L, H, O, MB = 1024, 440, 6, 2048, 2000, 8
    eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, max_frames=T, num_steps=1000))
    eng.init_hidden_weights(np.random.default_rng(7))
    Xs = [torch.randn(T, F, device="cuda") for _ in range(MB)]
    ys = [torch.randint(0, O, (T,), device="cuda", dtype=torch.int32) for _ in range(MB)]
    torch.cuda.synchronize()

    def step():
        for i in range(MB):
            eng.accumulate_device(Xs[i].data_ptr(), F, ys[i].data_ptr(), T, last=(i == MB - 1))
        return eng.apply()

    for _ in range(3):
        step()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        loss = step()
    dt = (time.perf_counter() - t0) / K
    print("8 micro-batches x 1024 frames per step: %.3f ms/step  %.0f frames/s  loss %.4f" % (dt * 1e3, MB * T / dt, loss))
    eng.close()


if __name__ == "__main__":
    main()
