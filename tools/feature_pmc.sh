#!/bin/bash
# GPU tool: LDS-array cycles / bank conflicts of feat_frames_kernel per frame.  usage: bash tools/feature_pmc.sh
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/featpmc
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/featpmc -- python $GRAFT_REPO_ROOT/tools/feature_bench.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<EOF
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/featpmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "feat_frames" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
print(m, "per frame (354k frames/launch): lds cycles %.0f, conflicts %.0f" % (m["SQ_LDS_IDX_ACTIVE"]/354e3, m["SQ_LDS_BANK_CONFLICT"]/354e3))
EOF
