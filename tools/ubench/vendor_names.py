"""GPU tool, measurement only: which vendor-library kernels torch.mm picks on the hot-path shapes (kernel names carry the
macro tile / MFMA shape).  Run under rocprofv3 --kernel-trace --stats."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
for dt in (torch.float32, torch.bfloat16):
    for (M, N, K) in ((1024, 2048, 2048), (2048, 4096, 4096)):
        a = torch.randn(M, K, device="cuda").to(dt)
        b = torch.randn(K, N, device="cuda").to(dt)
        bt = torch.randn(N, K, device="cuda").to(dt)
        at = torch.randn(K, M, device="cuda").to(dt)
        for _ in range(3):
            torch.mm(a, b)
            torch.mm(a, bt.t())
            torch.mm(at.t(), b)
torch.cuda.synchronize()
