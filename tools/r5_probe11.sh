#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r5p11
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_native_exchange.py tests/test_gpu_rccl_single_rank.py -q -m gpu -k "adversarial or non_finite or bf16_wire or f32x3 or serial_gradient_sum or single_rank_rccl" > $out/pytest.log 2>&1; echo rc $? >> $out/pytest.log
tail -25 $out/pytest.log
bash tools/profile_round.sh r05 > $out/profile.log 2>&1
tail -5 $out/profile.log
for f in gpurun_out/r05.bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), (d["roofline"].get("hbm") or {}).get("GBps_over_the_step"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
