import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
a = torch.randn(1024, 2048); b = torch.randn(2048, 2048)
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    (a @ b)
    t0 = time.perf_counter()
    for _ in range(10): (a @ b)
    dt = (time.perf_counter() - t0) / 10
    print("threads %3d: %.2f ms  %.1f GFLOP/s" % (th, dt * 1e3, 2 * 1024 * 2048 * 2048 / dt / 1e9))
