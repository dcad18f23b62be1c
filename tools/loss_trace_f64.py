"""GPU tool: the first 20 optimiser steps of bench.py's workload (BASELINE cfg2, the bench's weights and micro-batches)
through the engine, the PyTorch-CPU fp32 stand-in and the float64 oracle; prints the three traces and each fp32
implementation's distance to the float64 referee (reference neuralNetworks/trainer.py:336-346).

    python tools/loss_trace_f64.py [steps] > profiles/rNN_loss_trace_f64.json
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.loss_trace import distances, f64_loss_trace  # noqa: E402
from oracle.torch_cpu_step import TorchCpuTrainer  # noqa: E402
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402


def _workload(steps):
    wl = bench.Workload("cfg2")
    with tempfile.TemporaryDirectory(prefix="tfkaldi_trace_") as d:
        batches = bench.make_batches(wl, 0, 1, min(steps, bench.MAX_RING), d)
    rng = np.random.default_rng(7)
    hidden = [(rng.standard_normal((bench.F if l == 0 else wl.H, wl.H)) / np.sqrt(bench.F if l == 0 else wl.H)).astype(np.float32)
              for l in range(wl.L)]
    return wl, batches, hidden


def engine_trace(steps, dtype, wl, batches, hidden):
    cfg = _lib.make_config(bench.F, wl.L, wl.H, wl.O, nonlin="relu", batch_norm=True, init_learning_rate=1e-3,
                           num_steps=3 * steps, max_frames=wl.T, compute_dtype=dtype)
    eng = Engine(cfg)
    for l, w in enumerate(hidden):
        eng.set(_lib.WEIGHTS, l, w)
    out = []
    for i in range(steps):
        X, y = batches[i % len(batches)]
        eng.accumulate(X, y, last=True)
        out.append(eng.apply())
    eng.close()
    return out


def traces(steps=20, dtype="float32"):
    """(engine trace(s), PyTorch-CPU fp32 trace, float64 referee); dtype: one arithmetic or a tuple of them (then the first
    element is the dict {dtype: trace})"""
    wl, batches, hidden = _workload(steps)
    F, L, H, O = bench.F, wl.L, wl.H, wl.O
    many = not isinstance(dtype, str)
    gpu = {d: engine_trace(steps, d, wl, batches, hidden) for d in (dtype if many else (dtype,))}
    cpu_t = TorchCpuTrainer(F, L, H, O, nonlin="relu", batch_norm=True)
    cpu_t.set_hidden_weights(hidden)
    cpu = []
    for i in range(steps):
        X, y = batches[i % len(batches)]
        cpu_t.accumulate(X, y)
        cpu.append(cpu_t.apply())
    ref = f64_loss_trace(batches, hidden, steps, F, L, H, O)
    return (gpu if many else gpu[dtype]), cpu, ref


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    gpu, cpu, ref = traces(steps, ("float32_mfma", "float32"))
    c_rel, c_run = distances(cpu, ref)
    out = {"steps": steps, "cpu_fp32": cpu, "float64": ref, "cpu_fp32_vs_f64_rel": c_rel, "cpu_fp32_vs_f64_max": max(c_rel)}
    for d, t in gpu.items():
        rel, _ = distances(t, ref)
        out["engine_" + d] = t
        out["engine_%s_vs_f64_rel" % d] = rel
        out["engine_%s_vs_f64_max" % d] = max(rel)
    print(json.dumps(out, indent=1))
