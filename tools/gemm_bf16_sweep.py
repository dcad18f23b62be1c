"""GPU tool: every tile configuration of the bf16 MFMA GEMM (tfkaldi_amd/csrc/gemm_bf16.hip) on the contractions of
BASELINE cfg3 / cfg4 per GPU: a correctness check against torch on the same bf16-rounded operands, then interleaved
timing rounds (all configurations inside one process, median of the rounds) in TFLOP/s (dense bf16 MFMA peak of
MI355X: 2500).  Usage: python tools/gemm_bf16_sweep.py [--quick] [out.json]"""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402

lib = _lib.load()
NAMES = ["r64x64", "r128x64", "r128x128", "d128x64s5", "d128x128s4", "d256x128s3", "d128x64s3", "d256x128w4", "d256x256k32pp"]
# TFK_SWEEP_CFGS="3,5": only these configurations
SEL = [int(x) for x in os.environ["TFK_SWEEP_CFGS"].split(",")] if os.environ.get("TFK_SWEEP_CFGS") else list(range(len(NAMES)))
LAY = ["NN", "NT", "TN"]


def shapes(T, F, H, O):
    return [("fwd0", 0, T, H, F), ("fwd", 0, T, H, H), ("fwdO", 0, T, O, H), ("dAO", 1, T, H, O), ("dA", 1, T, H, H),
            ("dWO", 2, H, O, T), ("dW", 2, H, H, T), ("dW0", 2, F, H, T)]


def p8(n):
    return (n + 7) & ~7


class Problem(object):
    def __init__(self, layout, M, N, K):
        g = torch.Generator(device="cuda").manual_seed(layout * 7919 + M + 3 * N + 5 * K)
        sa, sb = {0: ((M, K), (K, N)), 1: ((M, K), (N, K)), 2: ((K, M), (K, N))}[layout]
        self.a = torch.zeros(sa[0], p8(sa[1]), dtype=torch.bfloat16, device="cuda")
        self.b = torch.zeros(sb[0], p8(sb[1]), dtype=torch.bfloat16, device="cuda")
        self.a[:, :sa[1]] = torch.randn(sa, generator=g, device="cuda").to(torch.bfloat16)
        self.b[:, :sb[1]] = torch.randn(sb, generator=g, device="cuda").to(torch.bfloat16)
        self.c = torch.zeros(M, (N + 3) & ~3, device="cuda")
        self.layout, self.M, self.N, self.K = layout, M, N, K
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.args = (st, layout, ctypes.c_void_p(self.a.data_ptr()), self.a.shape[1], ctypes.c_void_p(self.b.data_ptr()),
                     self.b.shape[1], ctypes.c_void_p(self.c.data_ptr()), self.c.shape[1], M, N, K, None, 0)

    def run(self, cfg):
        lib.tfk_gemm_bf16_force_config(cfg)
        rc = lib.tfk_gemm_bf16(*self.args)
        assert rc == 0, lib.tfk_last_error()

    def reference(self):
        a, b = self.a.float(), self.b.float()
        sa = {0: (self.M, self.K), 1: (self.M, self.K), 2: (self.K, self.M)}[self.layout]
        sb = {0: (self.K, self.N), 1: (self.N, self.K), 2: (self.K, self.N)}[self.layout]
        a, b = a[:, :sa[1]].double(), b[:, :sb[1]].double()
        if self.layout == 0:
            return a @ b
        if self.layout == 1:
            return a @ b.t()
        return a.t() @ b

    def check(self, cfg):
        self.c.zero_()
        self.run(cfg)
        torch.cuda.synchronize()
        ref = self.reference()
        err = (self.c[:, :self.N].double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        return err / max(scale, 1e-30)

    def time_once(self, cfg, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.run(cfg)
        e0.record()
        for _ in range(iters):
            lib.tfk_gemm_bf16(*self.args)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters


def main():
    quick = "--quick" in sys.argv
    outfile = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = {}
    cfgs = SEL
    confs = {"cfg3/gpu": (1024, 440, 2048, 4000), "cfg4/gpu": (2048, 440, 4096, 8000)}
    # small ragged problems: every configuration must be right on edge tiles too
    for layout, M, N, K in [(0, 37, 29, 13), (1, 65, 63, 130), (2, 100, 250, 72), (0, 300, 200, 136), (2, 129, 257, 520)]:
        pr = Problem(layout, M, N, K)
        errs = [pr.check(c) for c in cfgs]
        print("check %s %dx%dx%d: " % (LAY[layout], M, N, K) + " ".join("%.1e" % e for e in errs), flush=True)
        assert max(errs) < 1e-5, "ragged-shape mismatch"
    for tag, (T, F, H, O) in confs.items():
        print("== %s  T=%d F=%d H=%d O=%d   (TFLOP/s, us; heuristic choice marked *)" % (tag, T, F, H, O))
        print("%-5s %-2s %5s %5s %5s | " % ("op", "ly", "M", "N", "K") + " ".join("%13s" % NAMES[c] for c in cfgs))
        for name, layout, M, N, K in shapes(T, F, H, O):
            pr = Problem(layout, M, N, K)
            errs = {c: pr.check(c) for c in cfgs}
            bad = [NAMES[c] for c in cfgs if errs[c] > 1e-5]
            lib.tfk_gemm_bf16_force_config(-1)
            pick = lib.tfk_gemm_bf16_config(M, N)
            rounds = 3 if quick else 7
            iters = 10 if quick else 20
            times = {c: [] for c in cfgs}
            for _ in range(rounds):
                for c in cfgs:
                    times[c].append(pr.time_once(c, iters))
            row = []
            for c in cfgs:
                ms = statistics.median(times[c])
                row.append((ms, 2.0 * M * N * K / ms / 1e9))
            out["%s/%s" % (tag, name)] = {"shape": [M, N, K], "layout": LAY[layout], "pick": pick,
                                         "configs": cfgs, "tflops": [r[1] for r in row], "us": [r[0] * 1e3 for r in row],
                                         "rel_err": [errs[c] for c in cfgs]}
            print("%-5s %-2s %5d %5d %5d | " % (name, LAY[layout], M, N, K) +
                  " ".join("%6.0f%s%5.0fus" % (tf, "*" if c == pick else " ", ms * 1e3)
                           for c, (ms, tf) in zip(cfgs, row)) + ("   WRONG: %s" % bad if bad else ""), flush=True)
    lib.tfk_gemm_bf16_force_config(-1)
    if outfile:
        json.dump(out, open(outfile[0], "w"), indent=1)


if __name__ == "__main__":
    main()
