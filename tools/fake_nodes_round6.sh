# real RCCL at world > 1 on the ONE GPU of a gpurun box (TFK_FAKE_NODES: every rank claims its own host): probe + the tests that use it
out=$GRAFT_REPO_ROOT/gpurun_out/r06_fake_nodes; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 240 python tools/rccl_fake_nodes_probe.py 8 > $out/probe8.log 2>&1; echo "probe8 rc=$?"; grep FAKE_NODES_OK $out/probe8.log
timeout 3000 python -m pytest tests/test_gpu_rccl_single_rank.py tests/test_gpu_dp_two_ranks.py -q -m gpu -k "${TESTS:-real_rccl or cfg2_size}" -p no:cacheprovider -rA --durations=15 > $out/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|PASSED|FAILED|^E " $out/tests.log | cut -c1-250 | tail -60
