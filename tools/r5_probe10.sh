#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r5p10
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_loss_trace.py tests/test_gpu_stacked.py tests/test_gpu_packed_feed.py -q -m gpu -k "adversarial or non_finite or timeout or loss or stacked_evaluation or packed_feed" -s > $out/pytest.log 2>&1; echo rc $? >> $out/pytest.log
grep -v "^$" $out/pytest.log | grep -i "wide exponents\|cancellation\|passed\|failed\|rc \|Error\|assert\|^ *[0-9]* *[0-9.e+-]* *[0-9.e+-]* *[0-9.e+-]*$" | tail -60
timeout 600 python bench.py --steps 100 --warmup 10 > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -3 $out/bench.err
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5p10/bench.json"))
for k in ("value","ms_per_step","dtype","api_fed_value","api_fed_value_recipe","api_fed_value_recipe_with_validation","validation_cost","host_fed_value","loss_trace_f64_max_rel_diff","posterior_max_err"): print(k, d.get(k))
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("traffic_source","peak_note","hbm")})
print("exact", {k:d["exact_fp32"][k] for k in ("value","ms_per_step","loss_trace_f64_max_rel_diff")}, d["exact_fp32"]["roofline"])
print("eval", d.get("eval")); print("decode", d["decode"]["value"], d["decode"]["passes"])
print("cpu", d.get("cpu_baseline",{}).get("value"))
print("xm8", {k:v for k,v in d["exchange_model"]["per_world"]["8"].items() if k!="spans"})
print(d["kernel_ms_per_step"])
PY
