#!/bin/bash
# GPU tool: everything profiles/<round>_* is made from.  usage: bash tools/profile_round.sh <tag>   (output: gpurun_out/<tag>*, and
# the summaries tools/hbm_traffic.py writes straight into profiles/<tag>_*).  rocprofv3 runs: kernel trace + stats of the default
# bench command; FETCH_SIZE / WRITE_SIZE in separate counter-only passes per configuration (tools/hbm_counters.sh,
# MI355X_MICROARCH.md); two SQ counter passes; then the un-profiled bench lines of every configuration and the loss traces.
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
# (the clock pre-warm GEMMs of bench.py are switched off under the profiler: they would fill the kernel statistics; so are the
#  decode / eval / Nnet.train / other-arithmetic legs, measured un-profiled below)
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0   # (neither the pre-warm GEMMs nor the sustained leg's thousands of steps)
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 50 --warmup 5 > $out.trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- $B --steps 10 --warmup 3 > $out.pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- $B --steps 10 --warmup 3 > $out.pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/hbm_counters.sh ${tag}_hbm_cfg2 cfg2 float32 profiles/$tag > $out.hbm_cfg2.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg2m cfg2 float32_mfma profiles/${tag}_mfma > $out.hbm_cfg2m.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg2b cfg2 bfloat16 profiles/${tag}_bf16 > $out.hbm_cfg2b.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg3 cfg3 bfloat16 profiles/$tag > $out.hbm_cfg3.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg4 cfg4 bfloat16 profiles/$tag > $out.hbm_cfg4.log 2>&1
for c in cfg2 cfg2x3 cfg3 cfg4; do timeout 200 python tools/step_line.py $c $out.step_$c.json > /dev/null 2>&1; done
bash tools/step_kernel_stats.sh cfg3 $out.cfg3_kernel_stats.txt > /dev/null 2>&1
bash tools/step_kernel_stats.sh cfg4 $out.cfg4_kernel_stats.txt > /dev/null 2>&1
unset TFK_BENCH_PREWARM_MS TFK_BENCH_SUSTAIN_S
timeout 400 python bench.py --steps 100 --warmup 10 > $out.bench.json 2> $out.bench.err
timeout 400 python bench.py --steps 100 --warmup 10 --dtype float32_mfma --no-cpu-baseline --no-api-fed > $out.bench_mfma.json 2> $out.bench_mfma.err
timeout 300 python bench.py --steps 100 --warmup 10 --dtype bfloat16 --no-cpu-baseline > $out.bench_bf16.json 2> $out.bench_bf16.err
timeout 300 python bench.py --config cfg3 --steps 100 --warmup 10 > $out.bench_cfg3.json 2> $out.bench_cfg3.err
timeout 400 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out.bench_cfg4.json 2> $out.bench_cfg4.err
timeout 300 python tools/loss_trace_f64.py 20 > $out.loss_trace_f64.json 2> $out.loss_trace.err
ls $out* | head -60
