#!/bin/bash
# GPU tool (round 6): TFK_ADAM_NT = 1 (streaming g / m / v) against 2 (also the parameters and the twins the optimiser writes) on the cfg2 step
# each variant as (1) rocprofv3 kernel statistics of the bench command and (2) the un-profiled step time, interleaved three times
# usage: bash tools/bn_nt_ablate.sh [tag]; summary -> gpurun_out/<tag>/summary.txt
tag=${1:-bn_nt}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval --no-f64-trace"
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  TFK_ADAM_NT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/nt$v -- $B --steps 60 --warmup 5 > $out/nt$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
unset TFK_BENCH_PREWARM_MS
for rep in 1 2 3; do
  for v in 1 2; do
    TFK_ADAM_NT=$v timeout 200 $B --steps 100 --warmup 10 > $out/nt$v.bench$rep.json 2> $out/nt$v.bench$rep.err
  done
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, json, sys
out = sys.argv[1]
keys = (("bn_act_forward", "bn_act_forward_kernel", ""), ("hb_apply", "hb_apply_kernel", ""), ("fwd split-K pair", "gemm_bf16_dma_kernel<true, false, 9, 2, 4", ""),
        ("dual", "gemm_bf16x3_dual_kernel", ""), ("adam", "adam_kernel", ""))
print("# TFK_ADAM_NT variant: avg us per launch by rocprofv3 (calls) | un-profiled ms/step, three interleaved repetitions")
for v in (1, 2):
    f = glob.glob("%s/nt%d/**/*kernel_stats.csv" % (out, v), recursive=True)
    row = []
    if f:
        rows = list(csv.DictReader(open(f[0])))
        for label, a, b in keys:
            hit = [r for r in rows if a in r["Name"] and b in r["Name"]]
            hit.sort(key=lambda r: -int(r["Calls"]))
            row.append("%s %.2f (%s)" % (label, float(hit[0]["AverageNs"]) / 1e3, hit[0]["Calls"]) if hit else "%s -" % label)
    ms = []
    for rep in (1, 2, 3):
        try:
            ms.append("%.4f" % json.load(open("%s/nt%d.bench%d.json" % (out, v, rep)))["ms_per_step"])
        except Exception:
            ms.append("?")
    print("adam_nt=%d  %s | %s" % (v, "; ".join(row), " ".join(ms)))
PY
