"""GPU tool: the reference's own batch configuration -- 128 utterances per optimiser step in 8 micro-batches of 16
(config_AURORA4.cfg:134-141), here k x 1024 frames from HBM-resident data -- run the reference's way (one pass per
micro-batch: G accumulates, EPI_ACCUM from the second on, one Adam per step) and as ONE stacked pass
(tfk_accumulate_stacked: the GEMMs over all k micro-batches at once, batch-norm statistics / dropout per micro-batch).
Reference seam: neuralNetworks/trainer.py:310-332 (the micro-batch loop of Trainer.update).

    python tools/multi_mb_bench.py [cfg2|cfg3|cfg4 ...] [--k 8] > profiles/rNN_multi_mb.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402

CONFIGS = {  # frames per micro-batch, F, L, H, O, keep_prob, dtype
    "cfg2": (1024, 440, 6, 2048, 2000, 1.0, "float32"),
    "cfg2-bf16": (1024, 440, 6, 2048, 2000, 1.0, "bfloat16"),
    "cfg2-x3": (1024, 440, 6, 2048, 2000, 1.0, "float32x3"),  # fp32 emulated on the bf16 pipe
    "cfg3": (1024, 440, 6, 2048, 4000, 1.0, "bfloat16"),
    "cfg4": (2048, 440, 8, 4096, 8000, 0.5, "bfloat16"),
}


def run(name, k, steps=20):
    T, F, L, H, O, keep, dtype = CONFIGS[name]
    out = {}
    for mode in ("sequential", "stacked"):
        eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=keep, max_frames=T,
                                      num_steps=1000, compute_dtype=dtype, seed=3))
        eng.init_hidden_weights(np.random.default_rng(7))
        g = torch.Generator(device="cuda").manual_seed(1)
        X = torch.randn(k * T, F, device="cuda", generator=g)
        y = torch.randint(0, O, (k * T,), device="cuda", dtype=torch.int32, generator=g)
        torch.cuda.synchronize()

        def step():
            if mode == "stacked":
                eng.accumulate_stacked_device(X.data_ptr(), F, y.data_ptr(), k * T, [T] * k, last=True)
            else:
                for i in range(k):
                    eng.accumulate_device(X[i * T:].data_ptr(), F, y[i * T:].data_ptr(), T, last=(i == k - 1))
            return eng.apply()

        losses = [step() for _ in range(3)]
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            losses.append(step())
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[mode] = (dt, losses)
        eng.close()
    seq, stk = out["sequential"], out["stacked"]
    drift = max(abs(a - b) / abs(a) for a, b in zip(seq[1], stk[1]))
    print("%-10s %d x %4d frames/step  sequential %7.3f ms (%9.0f frames/s)   stacked %7.3f ms (%9.0f frames/s)   %+5.1f %%   "
          "loss %.4f vs %.4f (max rel. difference over %d steps %.1e)"
          % (name, k, T, seq[0] * 1e3, k * T / seq[0], stk[0] * 1e3, k * T / stk[0], 100.0 * (seq[0] / stk[0] - 1.0),
             seq[1][-1], stk[1][-1], len(seq[1]), drift))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["cfg2", "cfg2-bf16", "cfg3", "cfg4"])
    ap.add_argument("--k", type=int, nargs="*", default=[2, 4, 8])
    args = ap.parse_args()
    print("# k micro-batches per optimiser step from HBM-resident data: one pass per micro-batch vs one stacked pass")
    for name in args.configs:
        for k in args.k:
            if CONFIGS[name][0] * k > 8192 and False:
                continue
            run(name, k)


if __name__ == "__main__":
    main()
