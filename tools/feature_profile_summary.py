"""Turn the rocprofv3 passes of tools/profile_features.sh (gpurun_out/<tag>) into profiles/<round>_features_{kernel_stats.csv,
kernel_stats.txt,pmc.txt,bench.jsonl}.  usage: python tools/feature_profile_summary.py gpurun_out/r02_features profiles/r02"""
import collections
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime)


f = newest(src + "/trace/**/*kernel_stats.csv")
shutil.copy(f, dst + "_features_kernel_stats.csv")
with open(dst + "_features_kernel_stats.txt", "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/feature_bench.py 512   (MI355X; tools/profile_features.sh)\n"
              "# 512 utterances (1.0 h of 16 kHz int16 audio, 354k frames) per launch; fbank40 / mfcc13+ddelta / ssc40+delta, 23 launches each,\n"
              "# + the host-to-host and prepare_data legs (256 wav files, 16 speakers); avg/min/max in microseconds\n")
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "")
        out.write("%-100s calls=%5d avg_us=%9.2f min_us=%9.2f max_us=%9.2f pct=%s\n" % (
            name[:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in ("pmc1", "pmc2"):
    f = newest("%s/%s/**/*counter_collection.csv" % (src, p))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = ("feat_frames_kernel" if "feat_frames" in k else "dynamic_kernel" if "dynamic_kernel" in k
               else "frame_meta_kernel" if "frame_meta" in k else "cmvn_stats_kernel" if "cmvn" in k else None)
        if fam:
            acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(dst + "_features_pmc.txt", "w") as out:
    out.write("# rocprofv3 --pmc (counter-only passes, tools/profile_features.sh) over `python tools/feature_bench.py 512`, 1x MI355X: mean per launch\n"
              "# (354k frames per feat_frames_kernel launch).  SQ_* cycle counters are quad-cycles summed over all waves; SQ_LDS_IDX_ACTIVE =\n"
              "# LDS-array cycles summed over the 256 CUs, SQ_LDS_BANK_CONFLICT the part of them spent on conflicts.\n")
    for fam in ("feat_frames_kernel", "frame_meta_kernel", "dynamic_kernel", "cmvn_stats_kernel"):
        if fam not in acc:
            continue
        c = {k: sum(v) / len(v) for k, v in acc[fam].items()}
        d = sum(dur[fam]) / len(dur[fam])
        out.write("\n%s   (mean duration under counters %.1f us)\n" % (fam, d))
        for k in sorted(c):
            out.write("  %-28s %14.4g\n" % (k, c[k]))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            out.write("  %-28s %14.3f\n" % ("lds_busy = LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz)", c["SQ_LDS_IDX_ACTIVE"] / (256 * d * 1e-6 * 2.4e9)))
            out.write("  %-28s %14.3f\n" % ("bank_conflict / LDS_IDX_ACTIVE", c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
                if k in c:
                    out.write("  %-28s %14.3f\n" % (k + " / WAVE_CYCLES", c[k] / c["SQ_WAVE_CYCLES"]))
shutil.copy(src + ".bench.jsonl", dst + "_features_bench.jsonl")
