"""summarise gpurun_out/<tag>/<case>/**/*_counter_collection.csv written by tools/pmc_x3_clock.sh: per case the mean launch
duration, the cycles an XCD counted during it (GRBM_GUI_ACTIVE / 8 XCDs) -> the clock, and the matrix pipe's busy cycles per SIMD"""
import collections, csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "x3clock"
root = "gpurun_out/%s" % tag
print("# case: launches, mean duration under the counter pass, GRBM cycles per launch and XCD, clock = cycles / duration, matrix pipe busy")
print("# cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 1024) and their share of the launch's cycles")
for case in sorted(os.listdir(root)):
    files = glob.glob("%s/%s/*/*_counter_collection.csv" % (root, case))
    if not files:
        continue
    agg = collections.defaultdict(list)
    dur = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not dur:
        continue
    d = sorted(dur.values())
    d = d[len(d) // 5:]  # (the first launches run on a cold clock)
    mean = lambda v: sum(v) / max(len(v), 1)
    us = mean(d)
    grbm = mean(agg.get("GRBM_GUI_ACTIVE", [0])) / 8  # (the counter is summed over the 8 XCDs)
    mfma = mean(agg.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / 1024
    sqb = mean(agg.get("SQ_BUSY_CYCLES", [0]))
    print("%-16s n=%3d  %7.1f us  %9.0f cycles  %.2f GHz  mfma busy %8.0f cycles/SIMD = %.2f of the launch  (SQ_BUSY_CYCLES %.3g)"
          % (case, len(dur), mean(list(dur.values())), grbm, grbm / (mean(list(dur.values())) * 1e3), mfma, mfma / max(grbm, 1), sqb))
if os.path.exists(root + "/unprofiled.txt"):
    print("# un-profiled, 30 launches back to back:")
    print(open(root + "/unprofiled.txt").read().rstrip())
