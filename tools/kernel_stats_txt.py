"""Turn rocprofv3's <pid>_kernel_stats.csv (tools/profile_round.sh, trace pass) into the profiles/<round>_bench_kernel_stats
.csv / .txt pair.  usage: kernel_stats_txt.py gpurun_out/<tag> profiles/r01 ["note"]"""
import csv
import glob
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob("%s/trace/**/*kernel_stats.csv" % src, recursive=True)[0]
shutil.copy(f, dst + "_bench_kernel_stats.csv")
with open(dst + "_bench_kernel_stats.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 50 --warmup 5 "
              "--no-cpu-baseline   (MI355X%s)\n" % (", " + note if note else ""))
    out.write("# 5 warm-up + 50 timed (HBM-resident input) + 3 + 50 host-fed + 50 event-profiled steps of BASELINE cfg2; avg/min/max in "
              "microseconds (tfk::(anonymous namespace):: stripped)\n")
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("tfk::(anonymous namespace)::", "").replace("void tfk::", "")
        out.write("%-120s calls=%6d avg_us=%9.2f min_us=%9.2f max_us=%9.2f pct=%s\n" % (
            name[:120], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, r["Percentage"]))

# the dominant kernel's average launch, for bench.py to quote beside its own HIP-event figure (`roofline.avg_launch_us_rocprofv3`):
# stamped with the GEMM sources it was measured on, dropped by bench.py when stale
import json
import os
import sys as _sys
_sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfkaldi_amd.build import csrc_hash  # noqa: E402
rows = [r for r in csv.DictReader(open(f)) if "gemm_" in r["Name"]]
if rows:
    dom = max(rows, key=lambda r: float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]))
    ref_path = os.path.join(os.path.dirname(os.path.abspath(dst)), "kernel_stats_ref.json")
    book = json.load(open(ref_path)) if os.path.exists(ref_path) else {}
    key = _sys.argv[4] if len(_sys.argv) > 4 else "cfg2/float32"
    book[key] = {"kernel": dom["Name"].replace("tfk::(anonymous namespace)::", "").replace("void tfk::", "")[:100],
                 "avg_launch_us": float(dom["AverageNs"]) / 1e3, "calls": int(dom["Calls"]),
                 "source": os.path.basename(dst) + "_bench_kernel_stats.txt", "csrc_sha16": csrc_hash()}
    json.dump(book, open(ref_path, "w"), indent=1, sort_keys=True)
