#!/bin/bash
# GPU tool: HBM-side counters of one bench.py configuration.  Two counter-only rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do
# not fit one pass, MI355X_MICROARCH.md; --kernel-trace only, no other trace domain), summarised by tools/hbm_traffic.py.
# usage: bash tools/hbm_counters.sh <tag> <cfg2|cfg3|cfg4> <dtype> [profiles prefix, default profiles/r06]
tag=$1; cfg=$2; dtype=$3; pre=${4:-profiles/r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TFK_BENCH_PREWARM_MS=0   # (the clock pre-warm GEMMs would sit in the counters)
export TFK_BENCH_SUSTAIN_S=0    # (and so would the sustained leg's thousands of steps)
B="python $GRAFT_REPO_ROOT/bench.py --config $cfg --dtype $dtype --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out.fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out.write.log 2>&1
cd $GRAFT_REPO_ROOT
# bench.py ran 3 warm-up + 10 timed + 3 + 10 host-fed + 10 event-profiled steps
python tools/hbm_traffic.py gpurun_out/$tag $pre $cfg $dtype 36
