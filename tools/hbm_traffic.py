"""Summarise the two rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE: separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<round>_hbm_traffic.{json,txt}.

    python tools/hbm_traffic.py gpurun_out/<tag> profiles/r02

Also writes profiles/hbm_traffic.json (the copy bench.py reads), stamped with the hash of the kernel sources it was
measured on: bench.py drops the figure when the sources have changed since.

Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports exactly half the bytes of
wide (16 B/lane) coalesced reads (guide, "HBM" section); both counters are in KiB.  This is the L2 <-> fabric
side, so Infinity-Cache hits are included: an upper bound on what reached HBM."""
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

src, dst = sys.argv[1], sys.argv[2]
LAYOUT = {("true", "false"): "gemm_f32_nn(fwd affine)", ("true", "true"): "gemm_f32_nt(dA)",
          ("false", "false"): "gemm_f32_tn(dW)"}


def scope(kernel):
    if "gemm_f32_dual_kernel" in kernel:
        return "gemm_f32_dual(dA+dW)"
    m = re.search(r"gemm_f32_kernel<.*?Tile<(?:\d+, ){6}(true|false), (true|false)[,>]", kernel)
    if m:
        return LAYOUT.get((m.group(1), m.group(2)))
    for key, name in (("adam_kernel", "adam_apply"), ("bn_act_forward", "act_forward"), ("hb_stats", "hb_stats"),
                      ("hb_apply", "hb_apply"), ("softmax_xent", "softmax_xent")):
        if key in kernel:
            return name
    return None


def collect(sub, counter):
    acc = collections.defaultdict(list)
    names = {}
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (src, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                s = scope(r["Kernel_Name"])
                if s:
                    acc[s].append(float(r["Counter_Value"]))
                    names.setdefault(s, set()).add(r["Kernel_Name"])
    return acc, names


fetch, names = collect("fetch", "FETCH_SIZE")
write, _ = collect("write", "WRITE_SIZE")
out = {}
for s in fetch:
    if s not in write:
        continue
    f = sum(fetch[s]) / len(fetch[s])
    w = sum(write[s]) / len(write[s])
    out[s] = {"fetch_size_kb": f, "write_size_kb": w, "bytes_per_launch": (2 * f + w) * 1024,
              "launches_sampled": len(fetch[s]), "kernels": sorted(names[s])}
out["_meta"] = {"csrc_sha16": csrc_hash(), "measured": time.strftime("%Y-%m-%d"), "source": os.path.basename(dst)}
json.dump(out, open(dst + "_hbm_traffic.json", "w"), indent=1)
json.dump(out, open(os.path.join(os.path.dirname(dst) or ".", "hbm_traffic.json"), "w"), indent=1)
del out["_meta"]
with open(dst + "_hbm_traffic.txt", "w") as fid:
    fid.write("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --steps 10 --warmup 3`\n")
    fid.write("bytes/launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 correction, MI355X_MICROARCH.md)\n\n")
    fid.write("%-28s %8s %14s %14s %16s\n" % ("scope", "launches", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "bytes/launch"))
    for s, r in sorted(out.items()):
        fid.write("%-28s %8d %14.1f %14.1f %16.0f\n" % (s, r["launches_sampled"], r["fetch_size_kb"],
                                                       r["write_size_kb"], r["bytes_per_launch"]))
print(open(dst + "_hbm_traffic.txt").read())
