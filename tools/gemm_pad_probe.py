"""GPU tool: does the LEADING DIMENSION of the GEMM operands matter?  Rows that are a power-of-two number of bytes
apart (2048 floats = 8 KiB, 4096 bf16 = 8 KiB) can map every row of an operand tile onto the same L2 / fabric
channel.  Times the bf16 and fp32 GEMMs on the BASELINE shapes with the operands' leading dimensions padded by
0 .. 256 elements (interleaved rounds, median).  usage: python tools/gemm_pad_probe.py"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402

lib = _lib.load()
LAY = ["NN", "NT", "TN"]


def operands(layout, M, N, K, pad, dtype):
    sa, sb = {0: ((M, K), (K, N)), 1: ((M, K), (N, K)), 2: ((K, M), (K, N))}[layout]
    al = 8 if dtype == torch.bfloat16 else 4
    ld = lambda c: ((c + al - 1) // al * al) + pad
    a = torch.zeros(sa[0], ld(sa[1]), dtype=dtype, device="cuda")
    b = torch.zeros(sb[0], ld(sb[1]), dtype=dtype, device="cuda")
    a[:, :sa[1]] = torch.randn(sa, device="cuda").to(dtype)
    b[:, :sb[1]] = torch.randn(sb, device="cuda").to(dtype)
    c = torch.zeros(M, ((N + 3) & ~3) + (pad if pad % 4 == 0 else 0), device="cuda")
    return a, b, c


def runner(layout, M, N, K, pad, kind):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if kind == "bf16":
        a, b, c = operands(layout, M, N, K, pad, torch.bfloat16)
        args = (st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0)
        fn = lib.tfk_gemm_bf16
    else:
        a, b, c = operands(layout, M, N, K, pad, torch.float32)
        args = (st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0, -1)
        fn = lib.tfk_gemm_f32

    def once(iters=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        assert fn(*args) == 0, lib.tfk_last_error()
        e0.record()
        for _ in range(iters):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    once.keep = (a, b, c)
    return once


def main():
    pads = [0, 8, 32, 64, 128, 256]
    for kind, shapes in (("bf16", [("fwd4", 0, 2048, 4096, 4096), ("dA4", 1, 2048, 4096, 4096), ("dW4", 2, 4096, 4096, 2048),
                                   ("fwd3", 0, 1024, 2048, 2048), ("dA3", 1, 1024, 2048, 2048), ("dW3", 2, 2048, 2048, 1024)]),
                         ("f32", [("fwd2", 0, 1024, 2048, 2048), ("dA2", 1, 1024, 2048, 2048), ("dW2", 2, 2048, 2048, 1024),
                                  ("fwd4", 0, 2048, 4096, 4096)])):
        print("== %s  (TFLOP/s by padding of the operands' leading dimensions, in elements)" % kind)
        print("%-5s %-2s %5s %5s %5s | " % ("op", "ly", "M", "N", "K") + " ".join("%8s" % ("pad %d" % p) for p in pads))
        for name, layout, M, N, K in shapes:
            runs = [runner(layout, M, N, K, p, kind) for p in pads]
            times = [[] for _ in pads]
            for _ in range(5):
                for i, r in enumerate(runs):
                    times[i].append(r())
            tf = [2.0 * M * N * K / statistics.median(t) / 1e9 for t in times]
            print("%-5s %-2s %5d %5d %5d | " % (name, LAY[layout], M, N, K) + " ".join("%8.0f" % x for x in tf), flush=True)


if __name__ == "__main__":
    main()
