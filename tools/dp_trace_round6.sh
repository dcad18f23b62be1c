out=$GRAFT_REPO_ROOT/gpurun_out/r06_dptrace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in plain allreduce sharded sharded+planes; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/$mode -- python $GRAFT_REPO_ROOT/tools/dp_overhead.py cfg2 $mode > $out/$mode.log 2>&1
  echo "== cfg2 $mode"; grep DPOVERHEAD $out/$mode.log | cut -c1-110; python $GRAFT_REPO_ROOT/tools/dp_trace_gaps.py $out/$mode 3.0
done
