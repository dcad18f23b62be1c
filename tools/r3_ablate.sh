#!/bin/bash
# GPU tool (round 3): the step of each BASELINE configuration as one rank sees it with the round's two structural changes
# switched on and off -- the bf16 dual (dA + dW) launch (TFK_BF16_DUAL_CFG: 0 off, -1 heuristic, 3 / 4 / 5 forced geometry)
# and the per-layer optimiser on its own stream (TFK_ADAM_OVERLAP).  usage: bash tools/r3_ablate.sh <outdir>
out=${1:-gpurun_out/r3_ablate}
mkdir -p $out
for c in cfg3 cfg4; do
  for dual in 0 -1; do
    for ov in 0 1; do
      TFK_BF16_DUAL_CFG=$dual TFK_ADAM_OVERLAP=$ov timeout 200 python tools/step_line.py $c $out/${c}_dual${dual}_ov${ov}.json > /dev/null 2> $out/${c}_dual${dual}_ov${ov}.err
    done
  done
done
for g in 3 4 5; do
  TFK_BF16_DUAL_CFG=$g TFK_ADAM_OVERLAP=1 timeout 200 python tools/step_line.py cfg3 $out/cfg3_dual${g}_ov1.json > /dev/null 2>&1
done
TFK_BF16_DUAL_CFG=4 TFK_ADAM_OVERLAP=1 timeout 200 python tools/step_line.py cfg4 $out/cfg4_dual4_ov1.json > /dev/null 2>&1
for ov in 0 1; do
  TFK_ADAM_OVERLAP=$ov timeout 200 python tools/step_line.py cfg2 $out/cfg2_ov${ov}.json > /dev/null 2> $out/cfg2_ov${ov}.err
done
python - <<PY
import glob, json, os
rows = []
for f in sorted(glob.glob("$out/*.json")):
    d = json.loads(open(f).read().splitlines()[0])
    k = d["kernel_ms_per_step"]
    rows.append((os.path.basename(f)[:-5], d["ms_per_step"], d["step_frac_of_mfma_peak"], d["roofline"]["kernel"], d["roofline"]["frac"],
                 sum(v for n, v in k.items() if n.startswith("gemm")), k.get("adam_apply", 0),
                 sum(v for n, v in k.items() if not n.startswith("gemm") and n != "adam_apply")))
print("%-20s %8s %9s %-26s %6s %9s %8s %8s" % ("run", "ms/step", "step frac", "dominant kernel", "frac", "GEMMs ms", "adam ms", "small ms"))
for r in rows:
    print("%-20s %8.3f %9.3f %-26s %6.3f %9.3f %8.3f %8.3f" % r)
PY
