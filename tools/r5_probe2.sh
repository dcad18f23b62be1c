#!/bin/bash
# GPU tool (round 5, probe 2): where did the interleaved layout lose?  reads alone / fill alone, old vs new, and LDS conflict counters.
out=$GRAFT_REPO_ROOT/gpurun_out/r5p2
mkdir -p $out
cd $GRAFT_REPO_ROOT
{
for abl in 3 5; do
  for sh in "0 1024 2048 2048" "1 1024 2048 2048" "2 2048 2048 1024"; do
    echo -n "old "; timeout 60 tools/bin/x3old$abl $sh
    echo -n "new "; timeout 60 tools/bin/x3abl$abl $sh
  done
done
} > $out/ablate.txt 2>&1
bash tools/pmc_gemm_f32x3.sh r5nn 0 1024 2048 2048
bash tools/pmc_gemm_f32x3.sh r5nt 1 1024 2048 2048
bash tools/pmc_gemm_f32x3.sh r5tn 2 2048 2048 1024
cd $GRAFT_REPO_ROOT
for t in r5nn r5nt r5tn; do echo "== $t"; python tools/pmc_summary.py $t; done > $out/pmc.txt 2>&1
cat $out/ablate.txt $out/pmc.txt
