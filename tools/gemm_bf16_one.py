"""GPU tool: run ONE bf16 gemm shape repeatedly (timing; also the target of rocprofv3 --pmc).  args: layout M N K [iters]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib
lib = _lib.load()
layout, M, N, K = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
p8 = lambda n: (n + 7) & ~7
bf = dict(device="cuda", dtype=torch.bfloat16)
if layout == 0: a = torch.randn(M, p8(K), **bf); b = torch.randn(K, p8(N), **bf)
elif layout == 1: a = torch.randn(M, p8(K), **bf); b = torch.randn(N, p8(K), **bf)
else: a = torch.randn(K, p8(M), **bf); b = torch.randn(K, p8(N), **bf)
c = torch.zeros(M, (N + 3) & ~3, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 3):
    if it == 3: e0.record()
    assert lib.tfk_gemm_bf16(st, layout, ctypes.c_void_p(a.data_ptr()), a.shape[1], ctypes.c_void_p(b.data_ptr()), b.shape[1],
                             ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0) == 0
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("bf16 layout %d %dx%dx%d: %.1f us  %.1f TF" % (layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
