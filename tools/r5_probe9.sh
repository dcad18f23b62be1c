#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r5p9
mkdir -p $out
cd $GRAFT_REPO_ROOT
for ov in 0 1; do for rep in 1 2; do
TFK_ADAM_OVERLAP=$ov timeout 200 python tools/step_line.py cfg2x3 $out/step_ov${ov}_$rep.json > $out/step_ov${ov}_$rep.log 2>&1
done; done
TFK_ADAM_OVERLAP=1 timeout 200 python tools/step_line.py cfg3 $out/step_cfg3_ov1.json > /dev/null 2>&1
TFK_ADAM_OVERLAP=0 timeout 200 python tools/step_line.py cfg3 $out/step_cfg3_ov0.json > /dev/null 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5p9/step_*.json")):
    d=json.load(open(f)); print(os.path.basename(f), "%.4f ms"%d["ms_per_step"])
PY
timeout 1500 python -m pytest tests -q -m gpu --timeout 1200 > $out/pytest.log 2>&1; echo rc $? >> $out/pytest.log; tail -60 $out/pytest.log
