#!/bin/bash
# GPU tool (round 5, probe 1): the plane-interleaved x3 kernel -- correctness first, then A/B timing against round 4's layout.
out=$GRAFT_REPO_ROOT/gpurun_out/r5p1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_f32x3.py -x -q -k "contraction or split_k or transpose" > $out/pytest_gemm.log 2>&1
echo "pytest_gemm rc $?" >> $out/summary.txt
TFK_TEST_DTYPE=float32x3 timeout 900 python -m pytest tests/test_gpu_engine_parity.py tests/test_gpu_stacked.py -x -q -m gpu -k "not optimiser_on_its_own and not bf16" > $out/pytest_engine.log 2>&1
echo "pytest_engine rc $?" >> $out/summary.txt
{
for rep in 1 2; do
for abl in 0 1 6; do
  for sh in "0 1024 2048 2048" "1 1024 2048 2048" "2 2048 2048 1024" "2 440 2048 1024" "0 1024 2048 440" "0 1024 2000 2048"; do
    echo -n "old "; timeout 60 tools/bin/x3old$abl $sh
    echo -n "new "; timeout 60 tools/bin/x3abl$abl $sh
  done
  echo -n "new "; timeout 60 tools/bin/x3abl$abl 3 1024 2048 2048
  echo -n "new "; timeout 60 tools/bin/x3abl$abl 3 1024 2048 2000
  echo -n "new "; timeout 60 tools/bin/x3abl$abl 3 8192 2048 2048
done
done
} > $out/ablate.txt 2>&1
timeout 200 python tools/step_line.py cfg2x3 $out/step_cfg2x3.json > $out/step_cfg2x3.log 2>&1
timeout 200 python tools/step_line.py cfg2 $out/step_cfg2.json > $out/step_cfg2.log 2>&1
tail -3 $out/pytest_gemm.log $out/pytest_engine.log; cat $out/summary.txt; tail -30 $out/ablate.txt; tail -2 $out/step_cfg2x3.log
