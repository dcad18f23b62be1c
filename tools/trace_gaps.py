"""Idle time between kernels on the engine stream, from a rocprofv3 --kernel-trace csv.

    python tools/trace_gaps.py gpurun_out/<tag>/trace

Takes the steady-state launches (between the first and last adam_kernel), and reports, per optimiser step,
the summed kernel execution time, the wall time and the gap after each kernel kind (start of next - end of this)."""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
lo, hi = adam[5], adam[50]  # warm steps of the un-instrumented timed region
steps = 45
seg = rows[lo + 1:hi + 1]
busy = sum(e - s for s, e, _ in seg)
wall = seg[-1][1] - rows[lo][1]
print("steps %d: wall %.1f us/step, kernels busy %.1f us/step, idle %.1f us/step (%.1f %%), %d launches/step" %
      (steps, wall / steps / 1e3, busy / steps / 1e3, (wall - busy) / steps / 1e3, 100.0 * (wall - busy) / wall,
       len(seg) // steps))
gap = collections.defaultdict(list)
prev_end = rows[lo][1]
prev_name = rows[lo][2]
for s, e, n in seg:
    gap[prev_name.split("(")[0][:60]].append(s - prev_end)
    prev_end, prev_name = e, n
print("%-62s %6s %9s %9s" % ("gap AFTER kernel", "n", "avg ns", "us/step"))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s %6d %9.0f %9.2f" % (k, len(v), sum(v) / len(v), sum(v) / steps / 1e3))
