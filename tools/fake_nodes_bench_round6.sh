# `bench.py --gpus N` over N REAL RCCL ranks on the one GPU of a gpurun box (TFK_FAKE_NODES): the lines kept under profiles/
out=$GRAFT_REPO_ROOT/gpurun_out/r06_fake_nodes; mkdir -p $out
cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
  TFK_FAKE_NODES=1 TFK_BENCH_SUSTAIN_S=1 TFK_BENCH_DIAG_BUDGET_S=900 timeout 1000 python bench.py --gpus $n --steps 4 --warmup 2 > $out/bench_n$n.json 2> $out/bench_n$n.err; echo "bench n=$n rc=$?"
  python - <<P
import json
l = json.loads(open("$out/bench_n$n.json").read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("n_gpus", "value", "ms_per_step", "rccl_ranks", "incomplete", "api_fed_value", "api_fed_error", "collective_spans_last_step")})
print(l["exchange_model"]["timeline"][str($n)]["best_span_MiB"])
P
done
