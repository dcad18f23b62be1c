#!/bin/bash
# GPU tool: the cfg2 part of tools/profile_round.sh once more on the round's FINAL library (streaming stores in the BN / activation
# passes, the exchange's 32 MiB spans): kernel trace + stats, two SQ counter passes, FETCH_SIZE / WRITE_SIZE passes of cfg2 in the
# default arithmetic, then the un-profiled bench lines of cfg2 / cfg3 / cfg4.  usage: bash tools/profile_final_round6.sh [tag]
tag=${1:-r06f}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 50 --warmup 5 > $out.trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- $B --steps 10 --warmup 3 > $out.pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- $B --steps 10 --warmup 3 > $out.pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/hbm_counters.sh ${tag}_hbm_cfg2 cfg2 float32 profiles/r06 > $out.hbm_cfg2.log 2>&1
unset TFK_BENCH_PREWARM_MS TFK_BENCH_SUSTAIN_S
timeout 400 python bench.py --steps 100 --warmup 10 > $out.bench.json 2> $out.bench.err
timeout 300 python bench.py --config cfg3 --steps 100 --warmup 10 --no-cpu-baseline > $out.bench_cfg3.json 2> $out.bench_cfg3.err
timeout 400 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out.bench_cfg4.json 2> $out.bench_cfg4.err
timeout 200 python bench.py > $out.bench_default.json 2> $out.bench_default.err
mkdir -p $out/lines; cp $out.bench*.json $out/lines/ 2>/dev/null; cp profiles/r06_cfg2_hbm_traffic.txt profiles/hbm_traffic.json $out/lines/ 2>/dev/null
for f in bench bench_cfg3 bench_cfg4 bench_default; do python - <<P
import json
try:
    l = json.loads(open("$out.$f.json").read().strip().splitlines()[-1])
    print("$f", round(l["value"]), round(l["ms_per_step"], 4), l["roofline"]["frac"], l["roofline"].get("traffic"))
except Exception as e:
    print("$f", "no line", e)
P
done
