#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r5p3
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/pmc_gemm_f32x3.sh r5nn 0 1024 2048 2048
bash tools/pmc_gemm_f32x3.sh r5nt 1 1024 2048 2048
bash tools/pmc_gemm_f32x3.sh r5tn 2 2048 2048 1024
cd $GRAFT_REPO_ROOT
for t in r5nn r5nt r5tn; do echo "== $t"; python tools/pmc_summary.py $t; done > $out/pmc.txt 2>&1
cat $out/pmc.txt
