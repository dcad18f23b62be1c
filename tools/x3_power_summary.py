"""summarise gpurun_out/<tag>/*.smi.txt (rocm-smi samples taken by tools/x3_power_sample.sh while a kernel ran back to back)"""
import glob, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "x3power"
root = "gpurun_out/%s" % tag
idle = open(root + "/idle.txt").read() if os.path.exists(root + "/idle.txt") else ""
for line in idle.splitlines():
    if re.search(r"Max Graphics Package Power|sclk|Package Power", line):
        print("# idle: " + line.strip())
print("# case: samples, socket power mean / max (W), shader clock mean / min (MHz), the run's own line")
for f in sorted(glob.glob(root + "/*.smi.txt")):
    name = os.path.basename(f)[:-8]
    pw, ck = [], []
    for line in open(f):
        m = re.search(r"Package Power \(W\):\s*([0-9.]+)", line)
        if m:
            pw.append(float(m.group(1)))
        m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", line)
        if m:
            ck.append(float(m.group(1)))
    run = open(root + "/%s.run.txt" % name).read().strip().splitlines()
    if pw:
        print("%-16s n=%2d  power %6.0f / %6.0f W   sclk %5.0f / %5.0f MHz   %s" % (
            name, len(pw), sum(pw) / len(pw), max(pw), sum(ck) / max(len(ck), 1), min(ck) if ck else 0, run[-1] if run else ""))
    else:
        print("%-16s no samples parsed: %s" % (name, open(f).read()[:200].replace("\n", " | ")))
