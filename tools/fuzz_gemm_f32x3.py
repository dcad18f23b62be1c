"""GPU tool: random shapes through the fp32-emulating contraction (gemm_bf16x3) against float64, weighted towards the shapes
that run two blocks per tile over half of K each (128 <= tiles of 128x128 < 200 in multiples of 8, NN / NT), with ragged edges
and K that leaves the halves uneven; every case launched three times (the split-K exchange must not depend on block order).

    python tools/fuzz_gemm_f32x3.py [cases] [seed]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tfkaldi_amd import _lib  # noqa: E402
from test_gpu_f32x3 import _planes, p4  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    worst, split = 0.0, 0
    for case in range(cases):
        layout = int(rng.integers(0, 3))
        if rng.integers(0, 4) > 0:  # a split-K candidate: tiles_m * tiles_n in {64, 72, ..., 192}
            while True:
                tm, tn = int(rng.integers(1, 25)), int(rng.integers(1, 25))
                if 64 <= tm * tn < 200 and tm * tn % 8 == 0:
                    break
            M = tm * 128 - int(rng.integers(0, 128)) * int(rng.integers(0, 2))
            N = tn * 128 - int(rng.integers(0, 128)) * int(rng.integers(0, 2))
            K = int(rng.integers(512, 2600))
        else:
            M, N, K = int(rng.integers(1, 700)), int(rng.integers(1, 700)), int(rng.integers(1, 1500))
        epi = int(rng.choice({0: [0, 1], 1: [0], 2: [0, 2]}[layout]))
        A = torch.randn(*((K, M) if layout == 2 else (M, K)), device="cuda", generator=g) * 2
        B = torch.randn(*((N, K) if layout == 1 else (K, N)), device="cuda", generator=g)
        Ad, Bd = A.double(), B.double()
        ref = (Ad.T if layout == 2 else Ad) @ (Bd.T if layout == 1 else Bd)
        sab = (Ad.abs().T if layout == 2 else Ad.abs()) @ (Bd.abs().T if layout == 1 else Bd.abs())
        wide = 32 * int(rng.integers(0, 2))  # leading dimensions one unit longer than needed, or tight
        Ap, lda = _planes(lib, torch, A, ((A.shape[1] + 31) & ~31) + wide)
        Bp, ldb = _planes(lib, torch, B, ((B.shape[1] + 31) & ~31) + wide)
        ldc = p4(N)
        C0 = torch.randn(M, ldc, device="cuda", generator=g)
        bias = torch.randn(N, device="cuda", generator=g)
        outs = []
        for _ in range(3):
            C = C0.clone()
            _lib.check(lib.tfk_gemm_bf16x3(st, layout, ctypes.c_void_p(Ap.data_ptr()), lda, ctypes.c_void_p(Bp.data_ptr()), ldb,
                                           ctypes.c_void_p(C.data_ptr()), ldc, M, N, K, ctypes.c_void_p(bias.data_ptr()), epi))
            outs.append(C)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "case %d: launches differ" % case
        want = ref + (bias.double() if epi & 1 else 0) + (C0[:, :N].double() if epi & 2 else 0)
        err = (outs[0][:, :N].double() - want).abs()
        ratio = float((err / (4e-7 * (sab + want.abs()) + 1e-6)).max())
        assert ratio <= 1, "case %d layout %d %dx%dx%d epi %d: %.2f x the bound" % (case, layout, M, N, K, epi, ratio)
        assert bool((outs[0][:, N:] == C0[:, N:]).all())
        worst = max(worst, ratio)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        tiles64 = ((M + 127) // 128) * ((N + 63) // 64)
        split += int(K >= 512 and ((layout != 2 and 64 <= tiles < 200 and tiles % 8 == 0) or
                                   (layout == 2 and tiles < 100 and 64 <= tiles64 < 200 and tiles64 % 8 == 0)))
    print("== gemm_bf16x3: %d cases OK (%d of them split-K shapes); worst error %.3f of the fp32 kernels' bound "
          "(4e-7 * sum|ab| + 1e-6)" % (cases, split, worst))


if __name__ == "__main__":
    main()
