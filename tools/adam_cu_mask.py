"""GPU tool (round 6): what the optimiser's traffic costs when only HALF of the CUs issue it.

Why: Adam inside the dW epilogue (verdict r5 item 6) would move the optimiser's 30 B per weight into the blocks of the
backward launch that own dW tiles -- at cfg2 128 of the launch's 256 blocks (the other 128 hold one dA tile each), i.e. half
of the chip's CUs, 16 per XCD.  The stand-alone `adam_kernel` streams from all 256.  This tool runs the cfg2 step on streams
created with hipExtStreamCreateWithCUMask and prints the optimiser's time per launch from the engine's HIP-event profile:
all CUs (control), 128 CUs, 64 CUs.  The bytes per second a half chip sustains on THIS access pattern are what a fused
epilogue could reach at best (its dword-per-lane accesses in MFMA register layout are narrower than adam_kernel's float4s).

usage: python tools/adam_cu_mask.py   (prints a table; one JSON line at the end)
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd import _lib  # noqa: E402
from tfkaldi_amd import engine as engine_mod  # noqa: E402


def masked_stream(hip, words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return s.value


def run(label, words, hip, T, F, L, H, O):
    cfg = _lib.make_config(F, L, H, O, nonlin="relu", batch_norm=True, keep_prob=1.0, max_frames=T, num_steps=1000,
                           compute_dtype="float32")
    real_stream = torch.cuda.Stream
    if words is not None:
        ptr = masked_stream(hip, words)
        torch.cuda.Stream = lambda device=None: torch.cuda.ExternalStream(ptr, device=device)
    try:
        eng = engine_mod.Engine(cfg, torch_state=True)
    finally:
        torch.cuda.Stream = real_stream
    eng.init_hidden_weights(np.random.default_rng(7))
    X = torch.randn(T, F, device="cuda")
    y = torch.randint(0, O, (T,), device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    for _ in range(5):
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        eng.apply()
    eng.synchronize()
    n = 20
    eng.profile_begin()
    for _ in range(n):
        eng.accumulate_device(X.data_ptr(), F, y.data_ptr(), T, last=True)
        eng.apply()
    rows = {s["name"]: s for s in eng.profile_end()}
    out = {"label": label}
    for k, s in rows.items():
        if "adam" in k.lower() or "gemm" in k.lower() or "backward" in k.lower() or "forward" in k.lower():
            out[k] = round(1e3 * s["total_ms"] / max(1, s["launches"]), 2)
            if "adam" in k.lower():
                out[k + "_GBps"] = round(s["bytes"] / s["total_ms"] / 1e6, 0)
    print(label, json.dumps(out))
    eng.close() if hasattr(eng, "close") else None
    return out


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    T, F, L, H, O = 1024, 440, 6, 2048, 2000
    full = [0xFFFFFFFF] * 8
    res = [run("all 256 CUs (plain stream)", None, hip, T, F, L, H, O),
           run("mask: 256 bits set", full, hip, T, F, L, H, O),
           run("mask: bits 0..127 (16 CUs on every XCD: tools/ubench/cu_mask_map.hip)", [0xFFFFFFFF] * 4 + [0] * 4, hip, T, F, L, H, O),
           run("mask: every other bit (selects all 256: the mask acts on CU pairs)", [0x55555555] * 8, hip, T, F, L, H, O),
           run("mask: bits 0..63 (8 CUs on every XCD)", [0xFFFFFFFF] * 2 + [0] * 6, hip, T, F, L, H, O)]
    print(json.dumps({"tool": "adam_cu_mask", "shape": [T, F, L, H, O], "rows": res}))


if __name__ == "__main__":
    main()
