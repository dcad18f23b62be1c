#!/bin/bash
# GPU tool (round 6): TFK_X3_KSKEW = 0 .. 4 on the cfg2 step -- ring tiles (32 k) moved from the first block's share of a split-K
# forward contraction to the second's, so that the first reaches the hand-over early and the second does not wait for it
# each variant as (1) rocprofv3 kernel statistics of the bench command and (2) the un-profiled step time, interleaved three times
# a value may be "skew:prefetch" (TFK_X3_PREFETCH: the adder requests its partner's partial sums under its last ring tile)
# usage: bash tools/kskew_ablate.sh [tag] [values]; summary -> gpurun_out/<tag>/summary.txt
tag=${1:-kskew}
vals=${2:-"0 1 2 3 4"}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval --no-f64-trace"
cd /tmp && export TMPDIR=/tmp
for v in $vals; do
  export TFK_X3_KSKEW=${v%%:*}; case $v in *:*) export TFK_X3_PREFETCH=${v##*:};; *) unset TFK_X3_PREFETCH;; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k$v -- $B --steps 60 --warmup 5 > $out/k$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
unset TFK_BENCH_PREWARM_MS
for rep in 1 2 3; do
  for v in $vals; do
    export TFK_X3_KSKEW=${v%%:*}; case $v in *:*) export TFK_X3_PREFETCH=${v##*:};; *) unset TFK_X3_PREFETCH;; esac
    timeout 200 $B --steps 100 --warmup 10 > $out/k$v.bench$rep.json 2> $out/k$v.bench$rep.err
  done
done
python - $out $vals <<'PY' | tee $out/summary.txt
import csv, glob, json, sys
out = sys.argv[1]
vals = sys.argv[2:]
keys = (("fwd split-K pair", "gemm_bf16_dma_kernel<true, false, 9, 2, 4"), ("dual", "gemm_bf16x3_dual_kernel"),
        ("bn_act_forward", "bn_act_forward_kernel"), ("adam", "adam_kernel"))
print("# TFK_X3_KSKEW: avg us per launch by rocprofv3 (calls) | un-profiled ms/step, three interleaved repetitions | loss after the timed steps")
for v in vals:
    f = glob.glob("%s/k%s/**/*kernel_stats.csv" % (out, v), recursive=True)
    row = []
    if f:
        rows = list(csv.DictReader(open(f[0])))
        for label, a in keys:
            hit = [r for r in rows if a in r["Name"]]
            hit.sort(key=lambda r: -int(r["Calls"]))
            row.append("%s %.2f (%s)" % (label, float(hit[0]["AverageNs"]) / 1e3, hit[0]["Calls"]) if hit else "%s -" % label)
    ms, loss = [], None
    for rep in (1, 2, 3):
        try:
            l = json.loads(open("%s/k%s.bench%d.json" % (out, v, rep)).read().strip().splitlines()[-1])
            ms.append("%.4f" % l["ms_per_step"])
            loss = l.get("loss_first_last")
        except Exception as exc:
            ms.append("-")
    print("skew %-4s %s | %s | %s" % (v, "; ".join(row), " ".join(ms), loss))
PY
