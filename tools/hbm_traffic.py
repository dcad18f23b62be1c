"""Summarise the rocprofv3 --pmc passes of tools/hbm_counters.sh (FETCH_SIZE, WRITE_SIZE: separate counter-only runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes, plus the kernel trace that comes with each) over ONE bench.py configuration
into profiles/<round>_<cfg>_hbm_traffic.txt, and merge the per-kernel figures into profiles/hbm_traffic.json -- the copy
bench.py reads for `roofline.traffic` / `roofline.hbm`, stamped with the hash of the kernel sources it was measured on
(bench.py drops a stale record, tests/test_host_logic.py fails on one).

    python tools/hbm_traffic.py gpurun_out/<tag> profiles/r05 <cfg> <dtype> <steps the profiled command ran>

Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports exactly half the bytes of wide
(16 B/lane) coalesced reads (guide, "HBM" section); both counters are in KiB.  This is the L2 <-> fabric side, so
Infinity-Cache hits are included: an upper bound on what reached HBM.  GB/s = those bytes over the kernel's duration in the SAME
profiled run (rocprofv3 kernel trace of the FETCH pass; profiled runs clock ~3 % lower than un-profiled ones)."""
import collections
import csv
import glob
import json
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd.build import csrc_hash  # noqa: E402

LAYOUT = {("true", "false"): "nn(fwd affine)", ("true", "true"): "nt(dA)", ("false", "false"): "tn(dW)"}
SMALL = (("adam_kernel", "adam_apply"), ("bn_act_forward", "act_forward"), ("hb_stats", "hb_stats"), ("hb_apply", "hidden_backward"),
         ("softmax_xent", "softmax_xent"), ("loss_reduce", "loss_reduce"), ("colsum_partial", "colsum"), ("grad_final", "grad_final"),
         ("step_finish", "bn_ema_apply"), ("to_bf16_rows", "twin_of_input"), ("splice", "splice"), ("softmax_rows", "softmax_rows"))


def scope(kernel):
    """the engine's kernel-family label (bench.py: kernel_label) of a demangled kernel name; None: not a kernel of the step"""
    if "gemm_f32_dual_kernel" in kernel:
        return "gemm_f32_dual(dA+dW)"
    if "gemm_bf16x3_dual_kernel" in kernel:
        return "gemm_bf16x3_dual(dA+dW)"
    if "gemm_bf16_dual_kernel" in kernel:
        return "gemm_bf16_dual(dA+dW)"
    m = re.search(r"gemm_f32_kernel<.*?Tile<(?:\d+, ){6}(true|false), (true|false)[,>]", kernel)
    if m:
        return "gemm_f32_" + LAYOUT.get((m.group(1), m.group(2)), "?")
    m = re.search(r"gemm_bf16_(?:dma_)?kernel<(true|false), (true|false)((?:, \d+)*)>", kernel)
    if m:
        args = [int(x) for x in m.group(3).split(",") if x.strip()]
        x3 = len(args) >= 9 and args[8] == 3  # EPI, WM, WN, FM, FN, NS, BKT, SCHED, NPL, ...
        return ("gemm_bf16x3_" if x3 else "gemm_bf16_") + LAYOUT.get((m.group(1), m.group(2)), "?")
    for key, name in SMALL:
        if key in kernel:
            return name
    return None


def collect(src, sub, counter):
    acc, names, dur = collections.defaultdict(list), {}, collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (src, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                s = scope(r["Kernel_Name"])
                if s:
                    acc[s].append(float(r["Counter_Value"]))
                    names.setdefault(s, set()).add(r["Kernel_Name"])
                    if "End_Timestamp" in r and "Start_Timestamp" in r:
                        dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return acc, names, dur


def main():
    src, dst, cfg, dtype, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    fetch, names, dur = collect(src, "fetch", "FETCH_SIZE")
    write, _, _ = collect(src, "write", "WRITE_SIZE")
    rec = {}
    for s in fetch:
        if s not in write:
            continue
        f = sum(fetch[s]) / len(fetch[s])
        w = sum(write[s]) / len(write[s])
        us = sum(dur[s]) / len(dur[s]) if dur[s] else None
        rec[s] = {"fetch_size_kb": f, "write_size_kb": w, "bytes_per_launch": (2 * f + w) * 1024,
                  "launches_per_step": len(fetch[s]) / float(steps), "avg_launch_us_profiled": us,
                  "GBps": ((2 * f + w) * 1024 / (us * 1e-6) / 1e9) if us else None, "kernels": sorted(names[s])}
    step_bytes = sum(r["bytes_per_launch"] * r["launches_per_step"] for r in rec.values())
    step_us = sum(r["avg_launch_us_profiled"] * r["launches_per_step"] for r in rec.values() if r["avg_launch_us_profiled"])
    meta = {"csrc_sha16": csrc_hash(), "measured": time.strftime("%Y-%m-%d"), "source": os.path.basename(dst) + "_" + cfg,
            "config": cfg, "dtype": dtype, "steps_profiled": steps, "bytes_per_step": step_bytes,
            "kernel_us_per_step_profiled": step_us}
    path = os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json")
    book = json.load(open(path)) if os.path.exists(path) else {}
    if "_meta" in book:  # (round 1-4 layout: one flat cfg2 record)
        book = {}
    book["%s/%s" % (cfg, dtype)] = dict(rec, _meta=meta)
    json.dump(book, open(path, "w"), indent=1)
    with open("%s_%s_hbm_traffic.txt" % (dst, cfg), "w") as fid:
        fid.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate counter-only passes, tools/hbm_counters.sh) over\n"
                  "#   python bench.py --config %s --dtype %s --steps 10 --warmup 3   (1x MI355X; %d optimiser steps in the command)\n"
                  "# bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md): the L2 <-> fabric side,\n"
                  "# Infinity-Cache hits included.  GB/s = bytes / the kernel's duration in the same profiled run.\n\n" % (cfg, dtype, steps))
        fid.write("%-28s %9s %14s %14s %14s %10s %9s %8s\n" % ("kernel family", "per step", "FETCH_SIZE KiB", "WRITE_SIZE KiB",
                                                              "bytes/launch", "avg us", "GB/s", "of 8 TB/s"))
        for s, r in sorted(rec.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches_per_step"]):
            fid.write("%-28s %9.2f %14.1f %14.1f %14.0f %10.1f %9.0f %8.3f\n" % (
                s, r["launches_per_step"], r["fetch_size_kb"], r["write_size_kb"], r["bytes_per_launch"],
                r["avg_launch_us_profiled"] or 0, r["GBps"] or 0, (r["GBps"] or 0) / 8000.0))
        fid.write("\nper optimiser step: %.1f MB through the fabric in %.1f us of kernel time = %.0f GB/s = %.3f of 8 TB/s\n" % (
            step_bytes / 1e6, step_us, step_bytes / (step_us * 1e-6) / 1e9 if step_us else 0,
            step_bytes / (step_us * 1e-6) / 8e12 if step_us else 0))
    print(open("%s_%s_hbm_traffic.txt" % (dst, cfg)).read())


if __name__ == "__main__":
    main()
