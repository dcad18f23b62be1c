#!/bin/bash
# GPU tool (round 6): TFK_SLOT_EVENT=1 (an event record behind every micro-batch that read a host-fed input slot, rounds 1-5) against the
# default (none behind the LAST micro-batch of a step: the slot is free once the host has seen the step's loss) -- the bench command with
# its host-fed and Nnet.train legs, three interleaved repetitions.  usage: bash tools/slot_event_ablate.sh [tag]
tag=${1:-slot_event}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-other-arithmetic --no-eval --no-f64-trace"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 1 0; do
    TFK_SLOT_EVENT=$v timeout 300 $B --steps 100 --warmup 10 > $out/sl$v.bench$rep.json 2> $out/sl$v.bench$rep.err
  done
done
python - $out <<'PY' | tee $out/summary.txt
import json, sys
out = sys.argv[1]
print("# TFK_SLOT_EVENT: value | host_fed_value | api_fed_value | api_fed_value_recipe (frames/s), three interleaved repetitions")
for v in (1, 0):
    rows = []
    for rep in (1, 2, 3):
        try:
            l = json.loads(open("%s/sl%d.bench%d.json" % (out, v, rep)).read().strip().splitlines()[-1])
            rows.append("%d | %d | %d | %d" % (l["value"], l.get("host_fed_value") or 0, l.get("api_fed_value") or 0, l.get("api_fed_value_recipe") or 0))
        except Exception as exc:
            rows.append("- (%s)" % exc)
    print("%s   %s" % ("event behind every micro-batch (TFK_SLOT_EVENT=1)" if v else "none behind the last one (default)            ", "   ;   ".join(rows)))
PY
