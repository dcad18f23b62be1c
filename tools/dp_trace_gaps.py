"""GPU-idle gaps of one training step from a rocprofv3 --kernel-trace csv of tools/dp_overhead.py <cfg> <mode>:
    python tools/dp_trace_gaps.py <trace dir> [min gap us]
prints the step's wall time, the summed kernel time and every gap above the threshold with the kernel that follows it."""
import csv
import glob
import re
import sys


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    rows = []
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("tfk::", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", n)[:44]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if "softmax_xent" in r[2]][20:60]
    n = len(idx) - 1
    seg = rows[idx[0]:idx[-1]]
    wall = (rows[idx[-1]][0] - rows[idx[0]][0]) / n / 1e3
    busy = sum(e - s for s, e, _ in seg) / n / 1e3
    print("%d steps: wall %.1f us/step, kernels %.1f us/step, idle %.1f us/step, %.1f launches/step" % (
        n, wall, busy, wall - busy, len(seg) / n))
    a, b = idx[5], idx[6]
    t0, prev = rows[a][0], None
    for s, e, k in rows[a:b + 1]:
        gap = (s - prev) / 1e3 if prev else 0
        if gap > floor or "adam" in k:
            print("  +%7.1f  gap %6.1f  before %s (%.1f us)" % ((s - t0) / 1e3, gap, k, (e - s) / 1e3))
        prev = e


if __name__ == "__main__":
    main()
