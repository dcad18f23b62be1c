"""GPU tool (feasibility probe): an fp32 contraction EMULATED on the bf16 matrix pipe.  Each fp32 operand is split
into three bfloat16 pieces (8 + 8 + 8 significand bits: a = a1 + a2 + a3 exactly, by truncation); the six products of order
<= 2^-16 (or all nine) are accumulated in fp32 by ONE bf16 GEMM whose K dimension is the concatenation of the piece pairs --
A' = [A1 A1 A2 A1 A2 A3 ...], B' = [B1; B2; B1; B3; B2; B1; ...] -- through the library's existing bf16 kernel
(tfk_gemm_bf16).  This layout stages every piece tile once per product (twice the LDS fill a purpose-built kernel needs), so
its time is an UPPER bound; its accuracy is what any such kernel delivers.  Beside it: the exact-fp32 MFMA kernel
(tfk_gemm_f32), both against float64.

    python tools/bf16x3_probe.py > profiles/rNN_bf16x3_probe.txt
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402

lib = _lib.load()


def split3(x):
    """fp32 tensor -> three bf16 tensors with x == p1 + p2 + p3 (truncation of the significand, 8 bits per piece)"""
    out = []
    r = x.clone()
    for _ in range(3):
        p = (r.view(torch.int32) & -65536).view(torch.float32)  # keep sign, exponent, 7 stored significand bits
        out.append(p.to(torch.bfloat16))
        r = r - p
    return out, r


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(M, N, K, terms):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(K, N, device="cuda", generator=g) / np.sqrt(K)
    (a1, a2, a3), ra = split3(A)
    (b1, b2, b3), rb = split3(B)
    pairs = [(a1, b1), (a1, b2), (a2, b1), (a1, b3), (a2, b2), (a3, b1)]
    if terms == 9:
        pairs += [(a2, b3), (a3, b2), (a3, b3)]
    # smallest products first: they are added to a small accumulator
    pairs = pairs[::-1]
    Ap = torch.cat([p[0] for p in pairs], dim=1).contiguous()
    Bp = torch.cat([p[1] for p in pairs], dim=0).contiguous()
    C = torch.zeros(M, N, device="cuda")
    C32 = torch.zeros(M, N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    Kp = Ap.shape[1]

    def emu():
        assert lib.tfk_gemm_bf16(st, 0, ctypes.c_void_p(Ap.data_ptr()), Kp, ctypes.c_void_p(Bp.data_ptr()), N,
                                 ctypes.c_void_p(C.data_ptr()), N, M, N, Kp, None, 0) == 0

    def f32():
        assert lib.tfk_gemm_f32(st, 0, ctypes.c_void_p(A.data_ptr()), K, ctypes.c_void_p(B.data_ptr()), N,
                                ctypes.c_void_p(C32.data_ptr()), N, M, N, K, None, 0, -1) == 0

    t_emu, t_f32 = timed(emu), timed(f32)
    ref = A.double() @ B.double()
    scale = (A.double().abs() @ B.double().abs())
    e_emu = ((C.double() - ref).abs() / scale).max().item()
    e_f32 = ((C32.double() - ref).abs() / scale).max().item()
    rms_emu = ((C.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    rms_f32 = ((C32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print("%5dx%5dx%5d  %d products: %6.1f us (%6.1f TF fp32-equivalent)  exact-fp32 MFMA: %6.1f us (%6.1f TF)   "
          "max err / sum|ab|: emulated %.2e  fp32 %.2e   rms rel: %.2e  %.2e   split residue %.1e"
          % (M, N, K, terms, t_emu, 2.0 * M * N * K / t_emu / 1e6, t_f32, 2.0 * M * N * K / t_f32 / 1e6, e_emu, e_f32, rms_emu,
             rms_f32, max(ra.abs().max().item(), rb.abs().max().item())))


if __name__ == "__main__":
    print("# fp32 contraction emulated with bf16 pieces through the EXISTING bf16 kernel (K-concatenated piece pairs) vs the fp32 MFMA kernel")
    for shape in ((1024, 2048, 2048), (8192, 2048, 2048), (2048, 4096, 4096)):
        for terms in (6, 9):
            run(*shape, terms)
