# the whole GPU suite as the driver runs it, then a clean `bench.py --gpus 8` line over eight real RCCL ranks on this one GPU
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out/r06_fake_nodes
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x --durations=12 ) > $out/r06_gpusuite2.log 2>&1; echo "suite rc=$?"; tail -25 $out/r06_gpusuite2.log | cut -c1-200
TFK_FAKE_NODES=1 TFK_BENCH_SUSTAIN_S=1 TFK_BENCH_DIAG_BUDGET_S=900 timeout 1000 python bench.py --gpus 8 --steps 4 --warmup 2 > $out/r06_fake_nodes/bench_n8.json 2> $out/r06_fake_nodes/bench_n8.err; echo "bench n=8 rc=$?"
python - <<P
import json
l = json.loads(open("$out/r06_fake_nodes/bench_n8.json").read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("n_gpus", "value", "ms_per_step", "rccl_ranks", "incomplete", "api_fed_value", "api_fed_error")})
P
