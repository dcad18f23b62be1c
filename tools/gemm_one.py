"""GPU tool: run ONE gemm shape/config repeatedly (for rocprofv3 --pmc).  args: layout M N K cfg [iters]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib
lib = _lib.load()
layout, M, N, K, cfg = [int(x) for x in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
p4 = lambda n: (n + 3) & ~3
if layout == 0: a = torch.randn(M, p4(K), device="cuda"); b = torch.randn(K, p4(N), device="cuda")
elif layout == 1: a = torch.randn(M, p4(K), device="cuda"); b = torch.randn(N, p4(K), device="cuda")
else: a = torch.randn(K, p4(M), device="cuda"); b = torch.randn(K, p4(N), device="cuda")
c = torch.zeros(M, p4(N), device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 3):
    if it == 3: e0.record()
    assert lib.tfk_gemm_f32(st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                            ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0, cfg) == 0
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("layout %d %dx%dx%d cfg %d: %.1f us  %.1f TF" % (layout, M, N, K, cfg, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
