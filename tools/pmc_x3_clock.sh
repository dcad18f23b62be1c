#!/bin/bash
# GPU tool: at what CLOCK and with what matrix-pipe occupancy does the fp32-emulating contraction run?  rocprofv3 counter pass
# (GRBM_GUI_ACTIVE = cycles at the actual clock, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES; counters only + kernel trace) over
# tools/bin/x3abl0 (the kernel) and x3abl6 (its MFMAs alone, constant register operands) for operands of random significands
# and of zeros.  usage: bash tools/pmc_x3_clock.sh <tag>; summary: python tools/pmc_x3_clock.py <tag>
tag=${1:-x3clock}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name binary data args...
  local name=$1 bin=$2 data=$3; shift 3
  TFK_ABL_DATA=$data timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $out/$name -- $GRAFT_REPO_ROOT/tools/bin/$bin "$@" > $out/$name.log 2>&1
}
for data in random zero; do
  run dual_$data x3abl0 $data 3 1024 2048 2048
  run fwd8192_$data x3abl0 $data 0 8192 2048 2048
  run fwd1024_$data x3abl0 $data 0 1024 2048 2048
done
run dual_mfma x3abl6 random 3 1024 2048 2048
run fwd8192_mfma x3abl6 random 0 8192 2048 2048
# un-profiled, back to back (30 launches): the durations the counters have to be read against
for data in random zero; do
  echo "data $data:"
  TFK_ABL_DATA=$data $GRAFT_REPO_ROOT/tools/bin/x3abl0 3 1024 2048 2048
  TFK_ABL_DATA=$data $GRAFT_REPO_ROOT/tools/bin/x3abl0 0 8192 2048 2048
done > $out/unprofiled.txt 2>&1
echo "MFMAs alone (constant operands):" >> $out/unprofiled.txt
$GRAFT_REPO_ROOT/tools/bin/x3abl6 3 1024 2048 2048 >> $out/unprofiled.txt 2>&1
$GRAFT_REPO_ROOT/tools/bin/x3abl6 0 8192 2048 2048 >> $out/unprofiled.txt 2>&1
cd $GRAFT_REPO_ROOT && python tools/pmc_x3_clock.py $tag | tee $out/summary.txt
