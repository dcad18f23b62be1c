"""GPU tool: run ONE fp32-emulating GEMM (gemm_bf16x3) shape repeatedly (timing; the target of rocprofv3 --pmc).  args: layout M N K [iters]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib, x3
lib = _lib.load()
layout, M, N, K = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
p32 = lambda n: (n + 31) & ~31
shape_a = (K, M) if layout == 2 else (M, K)
shape_b = (N, K) if layout == 1 else (K, N)
def planes(shape):  # (timing only: random bf16 values in the interleaved array's place)
    ld = p32(shape[1])
    return torch.randn(x3.elems(shape[0], ld), device="cuda").to(torch.bfloat16), ld
a, lda = planes(shape_a)
b, ldb = planes(shape_b)
c = torch.zeros(M, (N + 3) & ~3, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 3):
    if it == 3: e0.record()
    _lib.check(lib.tfk_gemm_bf16x3(st, layout, ctypes.c_void_p(a.data_ptr()), lda, ctypes.c_void_p(b.data_ptr()), ldb,
                                   ctypes.c_void_p(c.data_ptr()), c.shape[1], M, N, K, None, 0))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("bf16x3 layout %d %dx%dx%d: %.1f us  %.1f TF fp32-equivalent" % (layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
