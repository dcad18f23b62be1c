#!/bin/bash
# GPU tool (round 6): A/B of two switches on the cfg2 step, each variant as (1) rocprofv3 kernel statistics of the bench command and
# (2) the un-profiled step time, interleaved twice so that box drift shows:
#   TFK_X3_HANDOVER = mem | l2     split-K hand-over of the forward contractions through memory / through the shared L2
#   TFK_CT_ROWS     = 16 | 32 | 64 rows per block of the column-tiled BN / activation kernels (32 = one batch of loads per thread)
# usage: bash tools/r6_ablate.sh [tag]; summary -> gpurun_out/<tag>/summary.txt
tag=${1:-r6abl}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval --no-f64-trace"
cd /tmp && export TMPDIR=/tmp
variant() {  # name env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- $B --steps 60 --warmup 5 > $out/$name.log 2>&1
}
variant mem TFK_X3_HANDOVER=mem
variant l2 TFK_X3_HANDOVER=l2
variant rows16 TFK_CT_ROWS=16
variant rows64 TFK_CT_ROWS=64
cd $GRAFT_REPO_ROOT
unset TFK_BENCH_PREWARM_MS
for rep in 1 2; do
  for v in "mem TFK_X3_HANDOVER=mem" "l2 TFK_X3_HANDOVER=l2" "rows16 TFK_CT_ROWS=16" "rows64 TFK_CT_ROWS=64"; do
    set -- $v
    env $2 timeout 200 $B --steps 100 --warmup 10 > $out/$1.bench$rep.json 2> $out/$1.bench$rep.err
  done
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, json, sys
out = sys.argv[1]
keys = (("fwd split-K pair", "gemm_bf16_dma_kernel<true, false", ", 3, 2"), ("dual", "gemm_bf16x3_dual_kernel", ""),
        ("bn_act_forward", "bn_act_forward_kernel", ""), ("hb_apply", "hb_apply_kernel", ""), ("adam", "adam_kernel", ""))
print("# variant: avg us per launch by rocprofv3 (calls) | un-profiled ms/step, two interleaved repetitions")
for name in ("mem", "l2", "rows16", "rows64"):
    f = glob.glob("%s/%s/**/*kernel_stats.csv" % (out, name), recursive=True)
    row = []
    if f:
        rows = list(csv.DictReader(open(f[0])))
        for label, a, b in keys:
            hit = [r for r in rows if a in r["Name"] and b in r["Name"]]
            hit.sort(key=lambda r: -int(r["Calls"]))
            row.append("%s %.2f (%s)" % (label, float(hit[0]["AverageNs"]) / 1e3, hit[0]["Calls"]) if hit else "%s -" % label)
    ms = []
    for rep in (1, 2):
        try:
            ms.append("%.4f" % json.load(open("%s/%s.bench%d.json" % (out, name, rep)))["ms_per_step"])
        except Exception as e:
            ms.append("?")
    print("%-7s %s | %s" % (name, "; ".join(row), " ".join(ms)))
PY
