#!/bin/bash
# GPU tool: everything profiles/<round>_* is made from.  usage: bash tools/profile_round.sh <tag>   (output: gpurun_out/<tag>*)
# rocprofv3 runs: kernel trace + stats; FETCH_SIZE / WRITE_SIZE in separate counter-only passes (MI355X_MICROARCH.md); two SQ
# counter passes; the same trace for the mixed-precision bench; then the per-configuration step lines.
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
# (the clock pre-warm GEMMs of bench.py are switched off under the profiler: they would fill the kernel statistics)
export TFK_BENCH_PREWARM_MS=0
# (the decode and Nnet.train legs of the bench line are measured un-profiled below: they would fill the kernel statistics)
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 50 --warmup 5 > $out.trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 10 --warmup 3 > $out.fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 10 --warmup 3 > $out.write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- $B --steps 10 --warmup 3 > $out.pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- $B --steps 10 --warmup 3 > $out.pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc3 -- $B --steps 10 --warmup 3 > $out.pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bf16 -- $B --steps 50 --warmup 5 --dtype bfloat16 > $out.trace_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
for c in cfg2 cfg3 cfg4; do timeout 200 python tools/step_line.py $c $out.step_$c.json > /dev/null 2>&1; done
unset TFK_BENCH_PREWARM_MS
timeout 300 python bench.py --steps 100 --warmup 10 > $out.bench.json 2> $out.bench.err
timeout 300 python bench.py --steps 100 --warmup 10 --dtype bfloat16 --no-cpu-baseline > $out.bench_bf16.json 2> $out.bench_bf16.err
timeout 300 python bench.py --config cfg3 --steps 100 --warmup 10 > $out.bench_cfg3.json 2> $out.bench_cfg3.err
timeout 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out.bench_cfg4.json 2> $out.bench_cfg4.err
ls $out* | head -40
