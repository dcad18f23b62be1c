"""GPU tool: one JSON line per BASELINE configuration as ONE rank sees it: step time, frames/s, per-kernel-family
HIP-event times, the dominant contraction against the MFMA roofline of its arithmetic, and the step's parameter-side
HBM traffic (40 * P bytes per step in fp32 master precision: W read fwd + bwd, dW written, Adam 16 read +
12 written; + 2 * P_w for the bf16 weight shadow in mixed precision) against the 8 TB/s HBM3E peak.
usage: python tools/step_line.py cfg2|cfg3|cfg4 [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd.engine import Engine  # noqa: E402

CONFIGS = {  # T per GPU, F, L, H, O, keep_prob, dtype
    "cfg2": (1024, 440, 6, 2048, 2000, 1.0, "float32"),
    "cfg2x3": (1024, 440, 6, 2048, 2000, 1.0, "float32x3"),  # fp32 emulated on the bf16 pipe
    "cfg3": (1024, 440, 6, 2048, 4000, 1.0, "bfloat16"),
    "cfg4": (2048, 440, 8, 4096, 8000, 0.5, "bfloat16"),
}
PEAK = {"float32": 157.3, "bfloat16": 2500.0, "float32x3": 2500.0 / 6}


def main():
    name = sys.argv[1]
    T, F, L, H, O, keep, dtype = CONFIGS[name]
    eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=keep, max_frames=T,
                                  num_steps=1000, compute_dtype=dtype))
    eng.init_hidden_weights(np.random.default_rng(7))
    X = torch.randn(T, F, device="cuda")
    y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()

    def step():
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        return eng.apply()

    for _ in range(5):
        step()
    eng.synchronize()
    K = 40
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / K
    eng.profile_begin()
    for _ in range(10):
        step()
    stats = eng.profile_end()
    macs = F * H + (L - 1) * H * H + H * O
    flop_per_frame = 6 * macs - 2 * F * H
    P = macs + (L * H + O) + L * H  # weights + biases + BN beta
    gemms = [s for s in stats if s["name"].startswith("gemm")]
    dom = max(gemms, key=lambda s: s["total_ms"])
    ach = dom["flops"] / dom["total_ms"] / 1e9
    param_bytes = 40.0 * P + (2.0 * macs if dtype == "bfloat16" else 0.0)
    adam = [s for s in stats if s["name"] == "adam_apply"][0]
    line = {
        "config": name, "workload": "%dx%d ReLU+BN%s, 440 in, %d pdf, %d frames per GPU per step, %s" % (
            L, H, " + dropout %.1f" % keep if keep < 1 else "", O, T,
            {"float32": "fp32 MFMA", "bfloat16": "bf16 MFMA (fp32 accumulate / master / optimiser)",
             "float32x3": "fp32 emulated on the bf16 MFMA pipe (3 bf16 planes per operand, 6 plane products)"}[dtype]),
        "ms_per_step": 1e3 * dt, "frames_per_s_per_gpu": T / dt, "parameters": P,
        "step_tflops": T / dt * flop_per_frame / 1e12, "step_frac_of_mfma_peak": T / dt * flop_per_frame / 1e12 / PEAK[dtype],
        "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": PEAK[dtype], "unit": "TFLOP/s",
                     "frac": ach / PEAK[dtype], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"]},
        "all_gemm_tflops": sum(s["flops"] for s in gemms) / sum(s["total_ms"] for s in gemms) / 1e9,
        "hbm": {"parameter_side_bytes_per_step": param_bytes, "GBps_over_the_step": param_bytes / dt / 1e9,
                "frac_of_8TBps": param_bytes / dt / 8e12,
                "adam_GBps": adam["bytes"] / adam["total_ms"] / 1e6, "adam_us": 1e3 * adam["total_ms"] / adam["launches"]},
        "kernel_ms_per_step": {s["name"]: s["total_ms"] / 10 for s in stats},
    }
    print(json.dumps(line))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fid:
            fid.write(json.dumps(line) + "\n")
    eng.close()


if __name__ == "__main__":
    main()
