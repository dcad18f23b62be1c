"""SURVEY.md 8d, CE-loss parity over the first 20 optimiser steps at BASELINE cfg2 (6x2048 ReLU + BN, 1024 frames per step,
the bench's weights and micro-batches), with the float64 oracle as the referee: the engine's distance to it is bounded by the
distance of the PyTorch-CPU fp32 restatement IN THE SAME RUN (the same arithmetic in another summation order: Adam amplifies
either one's rounding noise) -- not by the engine's own history.  Reference: neuralNetworks/trainer.py:336-346 (the value
Trainer.update returns)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# for the record only (no bound is derived from these): worst distance of each arithmetic to the float64 referee over the 20 steps
# as measured on MI355X -- exact fp32 MFMA 2.2e-4; emulated with the truncating split (round 5) 5.2e-4 .. 6.1e-4, with the
# round-to-nearest split (round 6) see profiles/r06_loss_trace_f64.json; PyTorch-CPU fp32 6.2e-4
FLOOR = 3e-4  # a CPU trace that happens to stay close must not make the bound tighter than two fp32 random walks can meet


@pytest.mark.timeout(900)
def test_engine_tracks_the_float64_oracle_over_20_steps(gpu):
    """both arithmetics of `compute_dtype = float32` -- the exact fp32 matrix instructions and the emulation on three bf16
    planes -- against ONE float64 referee run (22 s of host numpy)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from loss_trace_f64 import traces
    from oracle.loss_trace import distances
    gpu_t, cpu_t, ref = traces(20, ("float32_mfma", "float32"))
    c_rel, c_run = distances(cpu_t, ref)
    rel = {d: distances(t, ref)[0] for d, t in gpu_t.items()}
    print("step  fp32_mfma-vs-f64  fp32(emulated)-vs-f64  cpu_fp32-vs-f64")
    for k in range(20):
        print("%4d  %.3e         %.3e              %.3e" % (k, rel["float32_mfma"][k], rel["float32"][k], c_rel[k]))
    assert abs(ref[0] - np.log(2000)) < 1e-9  # KAT 8c-1: zero output layer -> ln O exactly
    # Tolerances.  Steps 0-2 are the pure round-off regime (a 1024-frame sum of per-frame losses in fp32: ~1e-7 relative;
    # measured 5e-8 .. 3e-7 for every implementation; at step 3 the exact-fp32 engine is still there, the emulated one and the
    # CPU stand-in are at 3.3e-6 / 1.5e-6): bounded absolutely.  From then on Adam has amplified the summation-order noise of the
    # near-zero gradients (the first updates are ~lr * sign(g)) and EVERY fp32 trace wanders around the float64 one chaotically:
    # a step-by-step comparison of random walks is a coin toss, so the whole trace of each engine arithmetic is bounded by TWICE
    # the worst distance the CPU fp32 trace reaches in this very run (or FLOOR) -- an fp32 implementation the engine had no part
    # in.  (Round 5 bounded each arithmetic by 3x its own measured history: a tolerance calibrated on the output it checks.)
    bound = 2.0 * max(max(c_rel), FLOOR)
    print("bound on every step of either engine trace: 2 x max(cpu worst %.3e, %.1e) = %.3e; engine worst: %s" % (
        max(c_rel), FLOOR, bound, {d: "%.3e" % max(r) for d, r in rel.items()}))
    for d, g_rel in rel.items():
        for k in range(3):
            assert g_rel[k] <= 2e-6, (d, k, g_rel[k])
        assert g_rel[3] <= 1e-5, (d, g_rel[3])
        for k in range(20):
            assert g_rel[k] <= bound, (d, k, g_rel[k], bound)
    assert max(c_rel) <= 2e-3, max(c_rel)  # (if the CPU stand-in itself moves this far, the comparison lost its meaning)
