#!/bin/bash
# GPU tool (round 6): the loss hand-over by an event record + synchronise (TFK_LOSS_EVENT=1, rounds 1-5) against the sequence word in
# mapped memory the host polls (default): un-profiled bench steps interleaved three times, and the gap in front of adam_kernel from a
# kernel trace of each.  usage: bash tools/loss_seq_ablate.sh [tag]; summary -> gpurun_out/<tag>/summary.txt
tag=${1:-loss_seq}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-other-arithmetic --no-eval --no-f64-trace"
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  TFK_LOSS_EVENT=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/ev$v -- $B --no-api-fed --steps 60 --warmup 5 > $out/ev$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
unset TFK_BENCH_PREWARM_MS
for rep in 1 2 3; do
  for v in 1 0; do
    TFK_LOSS_EVENT=$v timeout 300 $B --steps 100 --warmup 10 > $out/ev$v.bench$rep.json 2> $out/ev$v.bench$rep.err
  done
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, json, sys
out = sys.argv[1]
print("# TFK_LOSS_EVENT: idle stream in front of adam_kernel and per step (kernel trace, steps 10..45 of the timed region) | un-profiled ms/step x3 | api_fed / host_fed frames/s of the last repetition | loss")
for v in (1, 0):
    gap_adam = gap_all = None
    f = glob.glob("%s/ev%d/**/*kernel_trace.csv" % (out, v), recursive=True)
    if f:
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
        adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
        ga, gt, n = 0.0, 0.0, 0
        for k in range(10, 45):
            lo, hi = adam[k], adam[k + 1]
            prev = int(rows[lo]["End_Timestamp"])
            for r in rows[lo + 1:hi + 1]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                g = (s - prev) / 1e3
                gt += g
                if "adam_kernel" in r["Kernel_Name"]:
                    ga += g
                prev = e
            n += 1
        gap_adam, gap_all = ga / n, gt / n
    ms, extra = [], None
    for rep in (1, 2, 3):
        try:
            l = json.loads(open("%s/ev%d.bench%d.json" % (out, v, rep)).read().strip().splitlines()[-1])
            ms.append("%.4f" % l["ms_per_step"])
            extra = (round(l.get("api_fed_value") or 0), round(l.get("host_fed_value") or 0), l.get("loss_first_last"))
        except Exception:
            ms.append("-")
    print("%s  gap before adam %s us, all gaps %s us/step | %s | %s" % (
        "event (TFK_LOSS_EVENT=1)" if v else "sequence word (default)",
        "%.2f" % gap_adam if gap_adam is not None else "-", "%.2f" % gap_all if gap_all is not None else "-", " ".join(ms), extra))
PY
