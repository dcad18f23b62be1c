"""GPU tool, measurement only: this library's GEMMs beside the vendor library's (torch.mm -> hipBLASLt / rocBLAS) on the
same operands and shapes -- the contractions of BASELINE cfg2 (fp32) and of cfg3 / cfg4 per GPU (bf16 operands).  Nothing
in the product calls the vendor library; the figure says how much of the distance to the MFMA peak is this kernel's and
how much is the shape's (one wave of blocks, K of a few thousand, an fp32 result).
    python tools/gemm_yardstick.py > profiles/rNN_gemm_yardstick.txt
The library writes a bf16 result for bf16 operands (half the bytes of ours, which writes fp32 and, in the engine, a bf16
twin as well); where this torch build can ask for an fp32 result (out_dtype) that column is filled too."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402

lib = _lib.load()
LAY = ["NN", "NT", "TN"]


def shapes(T, F, H, O):
    return [("fwd0", 0, T, H, F), ("fwd", 0, T, H, H), ("fwdO", 0, T, O, H), ("dAO", 1, T, H, O), ("dA", 1, T, H, H),
            ("dWO", 2, H, O, T), ("dW", 2, H, H, T), ("dW0", 2, F, H, T)]


def p8(n):
    return (n + 7) & ~7


def timed(fns, rounds=7, iters=20):
    """Median us per launch of each callable; the rounds are interleaved (a, b, a, b, ...) so that clock ramps and
    neighbours on the box hit every candidate alike."""
    out = [[] for _ in fns]
    for _ in range(rounds):
        for k, fn in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return [statistics.median(o) for o in out]


def operands(layout, M, N, K, dtype, pad):
    g = torch.Generator(device="cuda").manual_seed(layout * 7919 + M + 3 * N + 5 * K)
    sa, sb = {0: ((M, K), (K, N)), 1: ((M, K), (N, K)), 2: ((K, M), (K, N))}[layout]
    a = torch.zeros(sa[0], pad(sa[1]), dtype=dtype, device="cuda")
    b = torch.zeros(sb[0], pad(sb[1]), dtype=dtype, device="cuda")
    a[:, :sa[1]] = torch.randn(sa, generator=g, device="cuda").to(dtype)
    b[:, :sb[1]] = torch.randn(sb, generator=g, device="cuda").to(dtype)
    return a, b, sa, sb


def vendor(layout, a, b, sa, sb, out_dtype=None):
    av, bv = a[:, :sa[1]], b[:, :sb[1]]
    if layout == 1:
        bv = bv.t()
    if layout == 2:
        av = av.t()
    if out_dtype is None:
        return lambda: torch.mm(av, bv)
    return lambda: torch.mm(av, bv, out_dtype=out_dtype)


def main():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("this library vs torch.mm (vendor GEMM), us per launch and TFLOP/s; 1x MI355X")
    print("== BASELINE cfg2, fp32 (peak 157.3)")
    print("%-5s %-2s %5s %5s %5s | %16s | %16s | %22s" % ("op", "ly", "M", "N", "K", "tfk_gemm_f32", "torch.mm fp32",
                                                          "tfk_gemm_bf16x3 (f32x3)"))
    torch.backends.cuda.matmul.allow_tf32 = False

    def planes(x, cols):
        """the exact three-plane bf16 split of x[:, :cols] (tfk_split3, interleaved: csrc/x3_layout.h): what the fp32-emulating
        contraction reads"""
        from tfkaldi_amd import x3
        return x3.split(lib, x[:, :cols])

    for name, layout, M, N, K in shapes(1024, 440, 2048, 2000):
        a, b, sa, sb = operands(layout, M, N, K, torch.float32, lambda n: (n + 3) & ~3)
        c = torch.zeros(M, (N + 3) & ~3, device="cuda")
        args = (st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0, -1)
        ap, lda = planes(a, sa[1])
        bp, ldb = planes(b, sb[1])
        c3 = torch.zeros_like(c)
        args3 = (st, layout, ctypes.c_void_p(ap.data_ptr()), lda, ctypes.c_void_p(bp.data_ptr()), ldb,
                 ctypes.c_void_p(c3.data_ptr()), c3.shape[1], M, N, K, None, 0)

        def ours():
            assert lib.tfk_gemm_f32(*args) == 0

        def ours3():
            assert lib.tfk_gemm_bf16x3(*args3) == 0
        t0, t1, t3 = timed([ours, vendor(layout, a, b, sa, sb), ours3])
        fl = 2.0 * M * N * K
        print("%-5s %-2s %5d %5d %5d | %7.1fus %6.1fTF | %7.1fus %6.1fTF | %7.1fus %6.1fTF-equivalent" % (
            name, LAY[layout], M, N, K, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t3, fl / t3 / 1e6), flush=True)
    for tag, dims in (("cfg3/gpu", (1024, 440, 2048, 4000)), ("cfg4/gpu", (2048, 440, 4096, 8000))):
        print("== BASELINE %s, bf16 operands (peak 2500)" % tag)
        print("%-5s %-2s %5s %5s %5s | %16s | %16s | %16s" % ("op", "ly", "M", "N", "K", "tfk_gemm_bf16->f32", "torch.mm ->bf16",
                                                             "torch.mm ->f32"))
        for name, layout, M, N, K in shapes(*dims):
            a, b, sa, sb = operands(layout, M, N, K, torch.bfloat16, p8)
            c = torch.zeros(M, (N + 3) & ~3, device="cuda")
            args = (st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                    ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0)

            def ours():
                assert lib.tfk_gemm_bf16(*args) == 0
            fns = [ours, vendor(layout, a, b, sa, sb)]
            try:
                f32 = vendor(layout, a, b, sa, sb, torch.float32)
                f32()
                fns.append(f32)
            except Exception:  # this torch build cannot ask the library for an fp32 result
                pass
            ts = timed(fns)
            t0, t1, t2 = ts[0], ts[1], (ts[2] if len(ts) > 2 else None)
            fl = 2.0 * M * N * K
            print("%-5s %-2s %5d %5d %5d | %7.1fus %6.0fTF | %7.1fus %6.0fTF | %s" % (
                name, LAY[layout], M, N, K, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6,
                "%7.1fus %6.0fTF" % (t2, fl / t2 / 1e6) if t2 else "       n/a"), flush=True)


if __name__ == "__main__":
    main()
