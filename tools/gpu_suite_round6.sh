# the whole GPU suite as the driver runs it, then the driver's default bench command on the same library
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 ) > $out/r06_gpusuite_final.log 2>&1; echo "suite rc=$?"; tail -22 $out/r06_gpusuite_final.log | cut -c1-200
timeout 600 python bench.py > $out/r06_bench_final2.json 2> $out/r06_bench_final2.err; echo "bench rc=$?"
python - <<P
import json
l = json.loads(open("$out/r06_bench_final2.json").read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("value", "ms_per_step", "lib_build_id", "api_fed_value")})
print(l["roofline"]["frac"], l["roofline"].get("avg_launch_us"), l["sustained"]["value"], l["sustained"].get("socket_power_w_mean"))
print({n: l["exchange_model"]["timeline"][n]["best_span_MiB"] for n in ("2", "4", "8")})
P
