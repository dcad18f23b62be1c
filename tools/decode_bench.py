"""GPU tool: Decoder throughput on BASELINE cfg2's network -- log(posterior / prior) for 96 utterances of 300 frames,
one utterance per pass (the reference's loop, nnet.py:270-286) vs decode_batch in passes of up to 8192 frames, with
host-spliced and device-spliced inputs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd.neuralNetworks.classifiers import activation as act  # noqa: E402
from tfkaldi_amd.neuralNetworks.classifiers.dnn import DNN  # noqa: E402
from tfkaldi_amd.neuralNetworks.decoder import Decoder  # noqa: E402
from tfkaldi_amd.processing.feature_reader import Unspliced, splice  # noqa: E402


def main():
    D, C, O, U, N = 40, 5, 2000, 96, 300
    F = D * (2 * C + 1)
    rng = np.random.default_rng(0)
    raw = [rng.standard_normal((N, D)).astype(np.float32) for _ in range(U)]
    dnn = DNN(O, 6, 2048, act.TfActivation(act.Batchnorm(None), "relu"), False)
    dec = Decoder(dnn, F, 8192)
    dec.engine.init_hidden_weights(rng)
    dec.set_prior(np.full(O, 1.0 / O, dtype=np.float32))
    spliced = [splice(r, C) for r in raw]
    deferred = [Unspliced(r, C) for r in raw]

    def groups(xs, budget):
        out, cur, n = [], [], 0
        for x in xs:
            if cur and n + x.shape[0] > budget:
                out.append(cur); cur, n = [], 0
            cur.append(x); n += x.shape[0]
        return out + [cur]

    for name, fn in (("one utterance per pass, host-spliced", lambda: [dec.log_likelihoods(x) for x in spliced]),
                     ("8192-frame passes, host-spliced", lambda: [dec.decode_batch(g) for g in groups(spliced, 8192)]),
                     ("8192-frame passes, device-spliced", lambda: [dec.decode_batch(g) for g in groups(deferred, 8192)])):
        fn()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        dt = (time.perf_counter() - t0) / 3
        print("%-40s %.1f ms  %.0f frames/s" % (name, dt * 1e3, U * N / dt))
    dec.close()


if __name__ == "__main__":
    main()
