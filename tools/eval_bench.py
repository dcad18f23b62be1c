"""GPU tool: evaluation-mode forward + loss (tfk_eval_accumulate on HBM-resident frames, BASELINE cfg2's network) with
the fused evaluation epilogue (EPI_EVAL_ACT, one launch per hidden layer) and with the three-launch path it replaces
(TFK_FUSE_EVAL=0), each in its own process; plus tools/decode_bench.py's figures for both."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import ctypes
    import numpy as np
    import torch
    from tfkaldi_amd import _lib
    from tfkaldi_amd.engine import Engine
    F, L, H, O = 440, 6, 2048, 2000
    for dtype in ("float32", "bfloat16"):
        for T in (300, 1024, 8192):
            eng = Engine(_lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, max_frames=T, compute_dtype=dtype))
            eng.init_hidden_weights(np.random.default_rng(7))
            X = torch.randn(T, F, device="cuda")
            y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
            torch.cuda.synchronize()
            run = lambda: _lib.check(eng.lib.tfk_eval_accumulate(eng._h, ctypes.c_void_p(X.data_ptr()), F,
                                                                 ctypes.c_void_p(y.data_ptr()), T, _lib.DEVICE_PTRS))
            for _ in range(5):
                run()
            eng.synchronize()
            K = 50
            t0 = time.perf_counter()
            for _ in range(K):
                run()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / K
            eng.eval_finish()
            print("  eval forward+loss %-8s T=%5d: %8.1f us/pass  %9.0f frames/s" % (dtype, T, dt * 1e6, T / dt), flush=True)
            eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for flag in ("0", "1"):
            print("TFK_FUSE_EVAL=%s (%s)" % (flag, "GEMM + bn_stats_eval + act_forward per layer" if flag == "0"
                                             else "one launch per layer: EPI_BIAS | EPI_EVAL_ACT"), flush=True)
            env = dict(os.environ, TFK_FUSE_EVAL=flag)
            subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, check=True)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "decode_bench.py")], env=env, check=True)
