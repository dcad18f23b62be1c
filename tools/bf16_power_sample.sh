#!/bin/bash
# GPU tool: socket power and shader clock while the bf16 GEMMs of BASELINE cfg4's per-GPU step run back to back for a few seconds --
# this library's kernel (tools/bin/abl0, heuristic block geometry) and the vendor library's (torch.mm), random operands and zeros.
# usage: bash tools/bf16_power_sample.sh [tag]; summary by tools/x3_power_summary.py
tag=${1:-bf16power}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showclocks --showpower > $out/idle.txt 2>&1
watch_pid() {  # name pid
  sleep 0.7
  : > $out/$1.smi.txt
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" >> $out/$1.smi.txt
    sleep 0.2
  done
  wait $2
}
for data in random zero; do
  ( TFK_ABL_DATA=$data TFK_ABL_ITERS=60000 tools/bin/abl0 0 2048 4096 4096 -1 > $out/own_nn_$data.run.txt 2>&1 ) & watch_pid own_nn_$data $!
  ( TFK_ABL_DATA=$data TFK_ABL_ITERS=60000 tools/bin/abl0 2 4096 4096 2048 -1 > $out/own_tn_$data.run.txt 2>&1 ) & watch_pid own_tn_$data $!
  ( python - $data > $out/vendor_nn_$data.run.txt 2>&1 <<'PY'
import sys, time, torch
M, N, K = 2048, 4096, 4096
z = sys.argv[1] == "zero"
a = (torch.zeros if z else torch.randn)(M, K, device="cuda", dtype=torch.bfloat16)
b = (torch.zeros if z else torch.randn)(K, N, device="cuda", dtype=torch.bfloat16)
for _ in range(20): torch.mm(a, b)
torch.cuda.synchronize(); t = time.perf_counter(); n = 0
while time.perf_counter() - t < 3.5:
    for _ in range(500): torch.mm(a, b)
    torch.cuda.synchronize(); n += 500
dt = (time.perf_counter() - t) / n
print("torch.mm bf16 %dx%dx%d: %7.1f us  %7.1f TF" % (M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12))
PY
  ) & watch_pid vendor_nn_$data $!
done
python tools/x3_power_summary.py $tag | tee $out/summary.txt
