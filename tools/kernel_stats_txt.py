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
