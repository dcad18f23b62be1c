"""GPU tool (round 6): how long before the END of an optimiser step does the host hold the step's loss?

step_finish writes (loss, frames, #micro-batches) to mapped pinned memory in front of the optimiser launch (~130 us at cfg2), so
that the host can enqueue the next step under it.  This measures, per step, the host time between tfk_apply_end returning
(the loss is there) and the engine stream running dry (hipStreamSynchronize returning), for the two hand-overs:
    TFK_LOSS_EVENT=1   event record behind step_finish + hipEventSynchronize (rounds 1-5)
    default            sequence word behind the scalars, polled by the host
usage: python tools/loss_handover_latency.py      (run once per setting of TFK_LOSS_EVENT)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def main():
    T, F, L, H, O = 1024, 440, 6, 2048, 2000
    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=1.0, max_frames=T, num_steps=1000, compute_dtype="float32")
    eng = Engine(cfg)
    eng.init_hidden_weights(np.random.default_rng(7))
    X = torch.randn(T, F, device="cuda")
    y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    lead, whole = [], []
    for it in range(60):
        t0 = time.perf_counter()
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        eng.apply_enqueue()
        eng.apply_end()
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        if it >= 10:
            lead.append((t2 - t1) * 1e6)
            whole.append((t2 - t0) * 1e6)
    lead.sort()
    print("TFK_LOSS_EVENT=%s: the host holds the loss %.1f us (median; min %.1f, max %.1f) before the stream runs dry; step %.1f us "
          "(each step synchronised: not the pipelined rate)" % (os.environ.get("TFK_LOSS_EVENT", "unset"), lead[len(lead) // 2], lead[0], lead[-1],
                                                                 sorted(whole)[len(whole) // 2]))


if __name__ == "__main__":
    main()
