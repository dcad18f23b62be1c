#!/bin/bash
# GPU tool (round 5, probe 8): tiled layout end to end -- correctness, then the cfg2 step in each block form
out=$GRAFT_REPO_ROOT/gpurun_out/r5p8
mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in 44 8 4; do
TFK_BF16X3_WAVES=$w timeout 600 python -m pytest tests/test_gpu_f32x3.py -x -q -k "contraction or split_k or transpose" > $out/pytest_gemm_w$w.log 2>&1
echo "pytest_gemm waves $w rc $?" >> $out/summary.txt
done
TFK_TEST_DTYPE=float32x3 timeout 900 python -m pytest tests/test_gpu_engine_parity.py tests/test_gpu_stacked.py -x -q -m gpu -k "not optimiser_on_its_own and not bf16" > $out/pytest_engine.log 2>&1
echo "pytest_engine rc $?" >> $out/summary.txt
for rep in 1 2; do
for w in 44 8 4; do
TFK_BF16X3_WAVES=$w timeout 200 python tools/step_line.py cfg2x3 $out/step_cfg2x3_w${w}_$rep.json > $out/step_cfg2x3_w${w}_$rep.log 2>&1
done
done
cat $out/summary.txt; for f in $out/pytest_*.log; do tail -n 3 $f; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5p8/step_*.json")):
    d=json.load(open(f)); k=d["kernel_ms_per_step"]
    print(os.path.basename(f), "%.4f ms"%d["ms_per_step"], "dual %.1f us"%d["roofline"]["avg_launch_us"], "fwd %.3f"%k["gemm_f32_nn(fwd affine)"], "dual %.3f"%k["gemm_f32_dual(dA+dW)"], "act %.3f hb %.3f adam %.3f"%(k["act_forward"],k["hidden_backward"],k["adam_apply"]))
PY
