#!/bin/bash
# GPU tool: rocprofv3 kernel statistics of the feature computation (tools/feature_bench.py).  usage: bash tools/profile_features.sh <tag>
tag=${1:-r02_features}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $GRAFT_REPO_ROOT/tools/feature_bench.py 512 > $out.trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc1 -- python $GRAFT_REPO_ROOT/tools/feature_bench.py 512 > $out.pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- python $GRAFT_REPO_ROOT/tools/feature_bench.py 512 > $out.pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/feature_bench.py 1024 $out.bench.jsonl > /dev/null 2>&1
find $out -name "*kernel_stats.csv" | head; ls $out*
