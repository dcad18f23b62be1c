"""GPU tool: the fp32-emulating GEMM (gemm_bf16x3: three bf16 planes per operand, six plane products, fp32 accumulate) against
float64 and beside the exact-fp32 MFMA kernel, per layout: max error / sum|ab| (the bound tests/test_gpu_gemm.py holds the fp32
kernel to is 4e-7 * sum|ab| + 1e-6), rms relative error, us per launch, fp32-equivalent TFLOP/s.

    python tools/gemm_f32x3_check.py [quick] > profiles/rNN_gemm_f32x3.txt
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib, x3 as x3l  # noqa: E402

lib = _lib.load()
p8 = lambda n: (n + 7) & ~7
p4 = lambda n: (n + 3) & ~3


def planes(x):
    """fp32 [rows, cols] on the device -> (interleaved three-plane bf16 array, its leading dimension): csrc/x3_layout.h"""
    return x3l.split(lib, x)


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(layout, M, N, K, epi=0, iters=20, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if layout == 0:
        A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g)
        ref = A.double() @ B.double(); sab = A.double().abs() @ B.double().abs()
    elif layout == 1:
        A = torch.randn(M, K, device="cuda", generator=g); B = torch.randn(N, K, device="cuda", generator=g)
        ref = A.double() @ B.double().T; sab = A.double().abs() @ B.double().abs().T
    else:
        A = torch.randn(K, M, device="cuda", generator=g); B = torch.randn(K, N, device="cuda", generator=g)
        ref = A.double().T @ B.double(); sab = A.double().abs().T @ B.double().abs()
    Ap, lda = planes(A)
    Bp, ldb = planes(B)
    # the split is exact
    for X, Xp, ld in ((A, Ap, lda), (B, Bp, ldb)):
        r, c = X.shape
        s = sum(q[:, :c].float() for q in x3l.planes(Xp, r, ld))
        assert torch.equal(s, X), "split3 is not exact"
    ldc = p4(N)
    C0 = torch.randn(M, ldc, device="cuda", generator=g) if epi & 2 else torch.zeros(M, ldc, device="cuda")
    bias = torch.randn(N, device="cuda", generator=g)
    C = C0.clone(); C32 = C0.clone()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    A4 = torch.zeros(A.shape[0], p4(A.shape[1]), device="cuda"); A4[:, :A.shape[1]] = A
    B4 = torch.zeros(B.shape[0], p4(B.shape[1]), device="cuda"); B4[:, :B.shape[1]] = B

    def x3():
        _lib.check(lib.tfk_gemm_bf16x3(st, layout, ctypes.c_void_p(Ap.data_ptr()), lda, ctypes.c_void_p(Bp.data_ptr()), ldb,
                                       ctypes.c_void_p(C.data_ptr()), ldc, M, N, K, ctypes.c_void_p(bias.data_ptr()), epi))

    def f32():
        _lib.check(lib.tfk_gemm_f32(st, layout, ctypes.c_void_p(A4.data_ptr()), A4.shape[1], ctypes.c_void_p(B4.data_ptr()), B4.shape[1],
                                    ctypes.c_void_p(C32.data_ptr()), ldc, M, N, K, ctypes.c_void_p(bias.data_ptr()), epi, -1))

    x3(); f32()
    torch.cuda.synchronize()
    want = ref + (bias.double() if epi & 1 else 0) + (C0[:, :N].double() if epi & 2 else 0)
    ex = ((C[:, :N].double() - want).abs() / (sab + want.abs() + 1e-30)).max().item()
    ef = ((C32[:, :N].double() - want).abs() / (sab + want.abs() + 1e-30)).max().item()
    rx = ((C[:, :N].double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    rf = ((C32[:, :N].double() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    assert (C[:, N:] == C0[:, N:]).all(), "padding columns written"
    if epi & 2:
        C.copy_(C0)
    tx = timed(x3, iters) if iters else 0.0
    tf = timed(f32, iters) if iters else 0.0
    fl = 2.0 * M * N * K
    print("%s %5dx%5dx%5d epi %d | bf16x3 %7.1f us %6.1f TF  fp32-MFMA %7.1f us %6.1f TF | max err/sum|ab|: x3 %.2e  f32 %.2e | rms rel: x3 %.2e  f32 %.2e"
          % (("NN", "NT", "TN")[layout], M, N, K, epi, tx, fl / tx / 1e6 if tx else 0, tf, fl / tf / 1e6 if tf else 0, ex, ef, rx, rf))
    return ex


if __name__ == "__main__":
    quick = len(sys.argv) > 1
    print("# fp32 emulated on three bf16 planes per operand (gemm_bf16x3) vs the exact-fp32 MFMA kernel, both against float64; 1x MI355X")
    worst = 0.0
    for layout in (0, 1, 2):  # ragged shapes: every edge predicate
        for (M, N, K) in ((197, 203, 75), (70, 330, 33), (130, 100, 64), (1, 1, 1), (129, 257, 1027)):
            worst = max(worst, run(layout, M, N, K, epi=(1 if layout == 0 else 2 if layout == 2 else 0), iters=0, seed=M))
    print("# BASELINE cfg2 shapes (1024 frames) and the stacked / cfg4 sizes")
    shapes = [(0, 1024, 2048, 440), (0, 1024, 2048, 2048), (0, 1024, 2000, 2048), (1, 1024, 2048, 2000), (1, 1024, 2048, 2048),
              (2, 2048, 2000, 1024), (2, 2048, 2048, 1024), (2, 440, 2048, 1024)]
    if not quick:
        shapes += [(0, 8192, 2048, 2048), (1, 8192, 2048, 2048), (2, 2048, 2048, 8192), (0, 2048, 4096, 4096), (2, 4096, 4096, 2048)]
    for layout, M, N, K in shapes:
        worst = max(worst, run(layout, M, N, K))
    print("# worst max err / sum|ab| of the emulation: %.2e" % worst)
