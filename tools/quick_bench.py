"""GPU tool: time the engine on BASELINE cfg2 (6x2048 ReLU+BN, 440 in, 2000 pdfs, 1024 frames) and print the
per-kernel-family HIP-event profile."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def main():
    # TFK_QB_SHAPE="T,F,L,H,O" (default BASELINE cfg2), TFK_QB_KEEP=<dropout keep prob>, TFK_QB_BN=0 disables BN
    T, F, L, H, O = [int(x) for x in os.environ.get("TFK_QB_SHAPE", "1024,440,6,2048,2000").split(",")]
    dtype = os.environ.get("TFK_QB_DTYPE", "float32")  # bfloat16: mixed-precision mode
    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=os.environ.get("TFK_QB_BN", "1") != "0",
                           keep_prob=float(os.environ.get("TFK_QB_KEEP", "1")), max_frames=T, num_steps=1000,
                           compute_dtype=dtype)
    eng = Engine(cfg)
    rng = np.random.default_rng(7)
    eng.init_hidden_weights(rng)
    X = torch.randn(T, F, device="cuda")
    y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    for _ in range(5):
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        loss = eng.apply()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        loss = eng.apply()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("cfg2: %.3f ms/step  %.0f frames/s  loss %.4f" % (dt * 1e3, T / dt, loss))
    eng.profile_begin()
    for _ in range(10):
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        eng.apply()
    tot = 0
    for s in eng.profile_end():
        tot += s["total_ms"]
        print("  %-28s n=%4d  %8.3f ms/step  %7.1f us/launch  %6.1f TF  %7.1f GB/s" % (
            s["name"], s["launches"] // 10, s["total_ms"] / 10, 1e3 * s["total_ms"] / s["launches"],
            s["flops"] / s["total_ms"] / 1e9, s["bytes"] / s["total_ms"] / 1e6))
    print("  sum of kernels: %.3f ms/step" % (tot / 10))


if __name__ == "__main__":
    main()
