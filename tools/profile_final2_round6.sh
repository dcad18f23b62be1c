#!/bin/bash
# GPU tool: every record that is stamped with the kernel sources, once more on the round's LAST library (uneven split-K halves):
# kernel trace + stats and two SQ counter passes of the cfg2 bench command, FETCH_SIZE / WRITE_SIZE passes of EVERY configuration
# and arithmetic bench.py quotes traffic for, summarised on the box (tools/kernel_stats_txt.py, pmc_step_summary.py,
# hbm_traffic.py write into profiles/; copied to gpurun_out/<tag>/profiles_out, which is what travels back), then the un-profiled
# bench lines.  usage: bash tools/profile_final2_round6.sh [tag]
tag=${1:-r06g}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out/profiles_out
cd $GRAFT_REPO_ROOT
export TFK_BENCH_PREWARM_MS=0 TFK_BENCH_SUSTAIN_S=0
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-decode --no-api-fed --no-other-arithmetic --no-eval"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 50 --warmup 5 > $out.trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- $B --steps 10 --warmup 3 > $out.pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- $B --steps 10 --warmup 3 > $out.pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/hbm_counters.sh ${tag}_hbm_cfg2 cfg2 float32 profiles/r06 > $out.hbm_cfg2.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg2m cfg2 float32_mfma profiles/r06_mfma > $out.hbm_cfg2m.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg2b cfg2 bfloat16 profiles/r06_bf16 > $out.hbm_cfg2b.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg3 cfg3 bfloat16 profiles/r06 > $out.hbm_cfg3.log 2>&1
bash tools/hbm_counters.sh ${tag}_hbm_cfg4 cfg4 bfloat16 profiles/r06 > $out.hbm_cfg4.log 2>&1
python tools/kernel_stats_txt.py gpurun_out/$tag profiles/r06 "round 6, last library: uneven split-K halves" > $out.kstats.log 2>&1
python tools/pmc_step_summary.py gpurun_out/$tag profiles/r06 > $out.pmcsum.log 2>&1
cp profiles/hbm_traffic.json profiles/kernel_stats_ref.json profiles/r06_bench_kernel_stats.csv profiles/r06_bench_kernel_stats.txt profiles/r06_bench_pmc.txt \
   profiles/r06_cfg2_hbm_traffic.txt profiles/r06_mfma_cfg2_hbm_traffic.txt profiles/r06_bf16_cfg2_hbm_traffic.txt profiles/r06_cfg3_hbm_traffic.txt \
   profiles/r06_cfg4_hbm_traffic.txt $out/profiles_out/ 2> $out.cp.log
unset TFK_BENCH_PREWARM_MS TFK_BENCH_SUSTAIN_S
timeout 400 python bench.py --steps 100 --warmup 10 > $out/profiles_out/r06_bench_last.json 2> $out.bench.err
timeout 300 python bench.py --config cfg3 --steps 100 --warmup 10 --no-cpu-baseline > $out/profiles_out/r06_bench_cfg3_last.json 2> $out.bench_cfg3.err
timeout 400 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out/profiles_out/r06_bench_cfg4_last.json 2> $out.bench_cfg4.err
timeout 300 python bench.py > $out/profiles_out/r06_bench_driver_cmd_last.json 2> $out.bench_default.err
for f in r06_bench_last r06_bench_cfg3_last r06_bench_cfg4_last r06_bench_driver_cmd_last; do python - <<P
import json
try:
    l = json.loads(open("$out/profiles_out/$f.json").read().strip().splitlines()[-1])
    print("$f", round(l["value"]), round(l["ms_per_step"], 4), l["roofline"]["frac"], l["roofline"].get("traffic"), l["roofline"].get("avg_launch_us_rocprofv3"))
except Exception as e:
    print("$f", "no line", e)
P
done
grep -E "dual_kernel|dma_kernel<true, false, 9|adam_kernel|hb_apply|bn_act_forward" profiles/r06_bench_kernel_stats.txt | cut -c1-60,118-200
