#!/bin/bash
# GPU tool (round 5, probe 4): 4 vs 8 waves per 128x128 block of the fp32-emulating contraction
out=$GRAFT_REPO_ROOT/gpurun_out/r5p4
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_f32x3.py -x -q -k "contraction or split_k or transpose" > $out/pytest_gemm.log 2>&1
echo "pytest_gemm rc $?" > $out/summary.txt
{
for w in 4 8; do
export TFK_BF16X3_WAVES=$w
for abl in 0 1 6; do
  for sh in "0 1024 2048 2048" "1 1024 2048 2048" "2 2048 2048 1024" "0 1024 2000 2048" "3 1024 2048 2048" "3 1024 2048 2000" "3 8192 2048 2048" "0 8192 2048 2048"; do
    echo -n "waves $w "; timeout 60 tools/bin/x3abl$abl $sh
  done
done
done
} > $out/ablate.txt 2>&1
for w in 4 8; do
TFK_BF16X3_WAVES=$w timeout 200 python tools/step_line.py cfg2x3 $out/step_cfg2x3_w$w.json > $out/step_cfg2x3_w$w.log 2>&1
done
cat $out/summary.txt; tail -n 4 $out/pytest_gemm.log; cat $out/ablate.txt; for w in 4 8; do tail -n 1 $out/step_cfg2x3_w$w.log | cut -c1-900; done
