"""GPU tool: the backward pair of one layer (dA = dZ . W^T, dW = in^T . dZ) as two launches vs ONE dual launch
(csrc/gemm_bf16.hip: gemm_bf16_dual_kernel) on the per-GPU shapes of BASELINE cfg3 / cfg4.  The block geometry of the dual
launch is an environment switch read once per process (TFK_BF16_DUAL_CFG), so every geometry runs in a child process;
each child checks the results against torch on the same bf16 operands before timing.
    python tools/gemm_bf16_dual_bench.py > profiles/rNN_gemm_bf16_dual.txt"""
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("cfg3 hidden", 1024, 2048, 2048), ("cfg3 output", 1024, 2048, 4000), ("cfg4 hidden", 2048, 4096, 4096),
          ("cfg4 output", 2048, 4096, 8000), ("cfg2-size", 1024, 2048, 2000)]


def child():
    import torch
    from tfkaldi_amd import _lib
    lib = _lib.load()
    p8 = lambda n: (n + 7) & ~7
    res = {}
    for name, T, d_in, d_out in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(T + d_in + d_out)

        def mat(r, c):
            m = torch.zeros(r, p8(c), dtype=torch.bfloat16, device="cuda")
            m[:, :c] = torch.randn(r, c, generator=g, device="cuda").to(torch.bfloat16)
            return m
        dz, W, X = mat(T, d_out), mat(d_in, d_out), mat(T, d_in)
        dA = torch.zeros(T, (d_in + 3) & ~3, device="cuda")
        G = torch.zeros(d_in, (d_out + 3) & ~3, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())

        def single():
            assert lib.tfk_gemm_bf16(st, 1, p(dz), dz.shape[1], p(W), W.shape[1], p(dA), dA.shape[1], T, d_in, d_out, None, 0) == 0
            assert lib.tfk_gemm_bf16(st, 2, p(X), X.shape[1], p(dz), dz.shape[1], p(G), G.shape[1], d_in, d_out, T, None, 0) == 0

        def dual():
            rc = lib.tfk_gemm_bf16_dual(st, p(dz), dz.shape[1], p(W), W.shape[1], p(dA), dA.shape[1], T, d_in, d_out,
                                        p(X), X.shape[1], p(dz), dz.shape[1], p(G), G.shape[1], d_in, d_out, T, 0)
            assert rc == 0, lib.tfk_last_error()

        cfg = lib.tfk_gemm_bf16_dual_config(T, d_in, d_in, d_out)
        ref_a = (dz[:, :d_out].double() @ W[:, :d_out].double().t())
        ref_w = (X[:, :d_in].double().t() @ dz[:, :d_out].double())
        row = {"dual_cfg": cfg}
        for tag, fn in (("two_launches", single),) + ((("dual", dual),) if cfg else ()):
            dA.zero_(); G.zero_()
            fn()
            torch.cuda.synchronize()
            ea = ((dA[:, :d_in].double() - ref_a).abs().max() / ref_a.abs().max()).item()
            ew = ((G[:, :d_out].double() - ref_w).abs().max() / ref_w.abs().max()).item()
            assert ea < 1e-5 and ew < 1e-5, (name, tag, ea, ew)
            times = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn()
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / 20 * 1e3)
            row[tag + "_us"] = statistics.median(times)
        row["flops"] = 4.0 * T * d_in * d_out
        res[name] = row
    print("DUALBENCH " + json.dumps(res))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    table = {}
    for cfg in ("-1", "3", "4", "5", "8"):
        env = dict(os.environ, TFK_BF16_DUAL_CFG=cfg)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True,
                           timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("DUALBENCH ")]
        if not line:
            print("geometry %s FAILED\n%s" % (cfg, r.stderr[-2000:]))
            continue
        table[cfg] = json.loads(line[0][len("DUALBENCH "):])
    print("dA (NT) + dW (TN) of one layer, bf16 operands: two launches (each with its own heuristic tile) vs one dual launch; us and TFLOP/s of the pair")
    print("%-12s %5s %5s %5s | %14s | %s" % ("layer", "T", "d_in", "d_out", "two launches", "  ".join(
        "dual %-9s" % {"-1": "heuristic", "3": "128x64", "4": "128x128", "5": "256x128", "8": "256x128+256x256"}[c] for c in table)))
    for name, T, d_in, d_out in SHAPES:
        base = None
        cells = []
        for c in table:
            row = table[c][name]
            base = row["two_launches_us"] if base is None else min(base, row["two_launches_us"])
            cells.append("%6.1f %5.0fTF" % (row["dual_us"], row["flops"] / row["dual_us"] / 1e6) if "dual_us" in row else "     n/a     ")
        print("%-12s %5d %5d %5d | %6.1f %5.0fTF | %s" % (name, T, d_in, d_out, base, table[list(table)[0]][name]["flops"] / base / 1e6,
                                                        "  ".join(cells)))
    print(json.dumps(table))


if __name__ == "__main__":
    main()
