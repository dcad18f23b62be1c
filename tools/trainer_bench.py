"""GPU tool: CrossEnthropyTrainer.update through the Python API at the reference's batch shape on BASELINE cfg2's
network (128 utterances x 64 frames per step, 16 per micro-batch), host-resident numpy inputs: spliced vs deferred
(CMVN + splice on the device).  Compare with tools/multi_mb_bench.py (the same step from device-resident data)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd.neuralNetworks.classifiers import activation as act  # noqa: E402
from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN  # noqa: E402
from tfkaldi_amd.neuralNetworks.trainer import CrossEnthropyTrainer  # noqa: E402
from tfkaldi_amd.processing.feature_reader import Unspliced, splice  # noqa: E402


def main():
    D, C, O, U, N = 40, 5, 2000, 128, 64
    F = D * (2 * C + 1)
    rng = np.random.default_rng(0)
    raw = [rng.standard_normal((N, D)).astype(np.float32) for _ in range(U)]
    ys = [rng.integers(0, O, size=N).astype(np.uint32) for _ in range(U)]
    cmvn = np.stack([np.zeros(D, np.float32), np.ones(D, np.float32)])
    inputs = {"spliced on the host": [splice(r, C) for r in raw],
              "deferred to the device": [Unspliced(r, C, cmvn=cmvn) for r in raw]}
    dnn = DNN(O, 6, 2048, act.TfActivation(act.Batchnorm(None), "relu"), False)
    for name, xs in inputs.items():
        tr = CrossEnthropyTrainer(dnn, F, N, N, 1e-3, 1.0, 1000, 16, seed=1)
        tr.initialize()
        for _ in range(3):
            tr.update(xs, ys)
        K = 10
        t0 = time.perf_counter()
        for _ in range(K):
            loss = tr.update(xs, ys)
        dt = (time.perf_counter() - t0) / K
        print("Trainer.update, inputs %-24s %.2f ms/step  %.0f frames/s  (loss %.4f)" % (name + ":", dt * 1e3, U * N / dt, loss))
        tr.close()


if __name__ == "__main__":
    main()
