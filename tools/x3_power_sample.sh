#!/bin/bash
# GPU tool: socket power and shader clock WHILE the fp32-emulating contraction runs back to back for a few seconds
# (tools/bin/x3abl0 / x3abl6 with TFK_ABL_ITERS), sampled with rocm-smi every 0.2 s.  usage: bash tools/x3_power_sample.sh [tag]
tag=${1:-x3power}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showclocks --showpower > $out/idle.txt 2>&1
sample() {  # name iters binary data args...
  local name=$1 iters=$2 bin=$3 data=$4; shift 4
  ( TFK_ABL_DATA=$data TFK_ABL_ITERS=$iters tools/bin/$bin "$@" > $out/$name.run.txt 2>&1 ) &
  local pid=$!
  sleep 0.7
  : > $out/$name.smi.txt
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" >> $out/$name.smi.txt
    sleep 0.2
  done
  wait $pid
}
sample fwd8192_random 9000 x3abl0 random 0 8192 2048 2048
sample fwd8192_zero 12000 x3abl0 zero 0 8192 2048 2048
sample fwd8192_mfma 12000 x3abl6 random 0 8192 2048 2048
sample dual_random 30000 x3abl0 random 3 1024 2048 2048
sample dual_zero 40000 x3abl0 zero 3 1024 2048 2048
python tools/x3_power_summary.py $tag | tee $out/summary.txt
