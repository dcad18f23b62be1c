"""GPU tool: time every tile configuration of the fp32 MFMA GEMM on the contractions of the BASELINE
configs and print TFLOP/s (peak fp32 MFMA on MI355X: 157.3).  Usage: python tools/gemm_sweep.py [out.json]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402

lib = _lib.load()
ALL = ["128x128/4w2", "128x64/4wp2", "64x128/4wp2", "64x64/4wp2", "128x128/8w3", "256x128/8w2", "64x128/8wp2", "128x64/8wp2",
       "64x64/4wp1", "64x64/dma4", "128x64/dma4", "64x128/dma4", "128x128/dma4"]
# TFK_SWEEP_CFGS="0,3,9,10,11,12": the configurations to time (default: all), interleaved rounds, median
CFGS = [int(x) for x in os.environ.get("TFK_SWEEP_CFGS", ",".join(str(i) for i in range(len(ALL)))).split(",")]
NAMES = [ALL[c] for c in CFGS]
LAY = ["NN", "NT", "TN"]


def shapes(T, F, H, O):
    return [("fwd0", 0, T, H, F), ("fwd", 0, T, H, H), ("fwdO", 0, T, O, H), ("dAO", 1, T, H, O), ("dA", 1, T, H, H),
            ("dWO", 2, H, O, T), ("dW", 2, H, H, T), ("dW0", 2, F, H, T)]


def bench(layout, M, N, K, cfg, iters=20):
    p4 = lambda n: (n + 3) & ~3
    if layout == 0:
        a = torch.randn(M, p4(K), device="cuda"); b = torch.randn(K, p4(N), device="cuda")
    elif layout == 1:
        a = torch.randn(M, p4(K), device="cuda"); b = torch.randn(N, p4(K), device="cuda")
    else:
        a = torch.randn(K, p4(M), device="cuda"); b = torch.randn(K, p4(N), device="cuda")
    c = torch.zeros(M, p4(N), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
            ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0, cfg)
    for _ in range(3):
        assert lib.tfk_gemm_f32(*args) == 0, lib.tfk_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.tfk_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


def main():
    out = {}
    confs = {"cfg2": (1024, 440, 2048, 2000), "cfg4/gpu": (2048, 440, 4096, 8000)}
    if os.environ.get("TFK_SWEEP_ONLY"):  # e.g. TFK_SWEEP_ONLY=cfg2
        confs = {k: v for k, v in confs.items() if k in os.environ["TFK_SWEEP_ONLY"].split(",")}
    for tag, (T, F, H, O) in confs.items():
        print("== %s  T=%d F=%d H=%d O=%d" % (tag, T, F, H, O))
        print("%-6s %-3s %6s %6s %6s | " % ("op", "lay", "M", "N", "K") + " ".join("%11s" % n for n in NAMES))
        for name, layout, M, N, K in shapes(T, F, H, O):
            times = {c: [] for c in CFGS}
            for _ in range(5):
                for c in CFGS:
                    times[c].append(bench(layout, M, N, K, c, iters=10)[0])
            row = []
            for c in CFGS:
                ms = sorted(times[c])[len(times[c]) // 2]
                row.append((ms, 2.0 * M * N * K / ms / 1e9))
            out["%s/%s" % (tag, name)] = row
            print("%-6s %-3s %6d %6d %6d | " % (name, LAY[layout], M, N, K) +
                  " ".join("%5.1fTF%4.0fus" % (tf, ms * 1e3) for ms, tf in row))
    ms, tf = bench(0, 4096, 4096, 4096, 0)
    print("4096^3 NN cfg0: %.1f TF" % tf)
    ms, tf = bench(0, 4096, 4096, 4096, 5)
    print("4096^3 NN cfg5: %.1f TF" % tf)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()
