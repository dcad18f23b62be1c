# real RCCL at world > 1 on the ONE GPU of a gpurun box (TFK_FAKE_NODES: every rank claims its own host): probe, tests, bench lines
out=$GRAFT_REPO_ROOT/gpurun_out/r06_fake_nodes; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 240 python tools/rccl_fake_nodes_probe.py 8 > $out/probe8.log 2>&1; echo "probe8 rc=$?"; grep FAKE_NODES_OK $out/probe8.log
timeout 2400 python -m pytest tests/test_gpu_rccl_single_rank.py -q -m gpu -k "${TESTS:-real_rccl_ranks}" -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/tests.log
for n in ${NS:-2 8}; do
  TFK_FAKE_NODES=1 TFK_BENCH_SUSTAIN_S=1 timeout 900 python bench.py --gpus $n --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_n$n.json 2> $out/bench_n$n.err; echo "bench n=$n rc=$?"
  python - <<P
import json
try:
    l = json.loads(open("$out/bench_n$n.json").read().strip().splitlines()[-1])
    print({k: l.get(k) for k in ("n_gpus", "value", "ms_per_step", "rccl_ranks", "dist_backend", "exchange", "incomplete", "exchange_algorithm", "api_fed_value", "api_fed_error")})
    print("exchange_ab:", json.dumps(l.get("exchange_ab"))[:800])
except Exception as e:
    print("no line:", e)
P
  grep -v "hostname of the client\|amdgpu.ids" $out/bench_n$n.err | tail -8
done
