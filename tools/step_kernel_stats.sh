#!/bin/bash
# GPU tool: rocprofv3 kernel statistics of one rank's step at a BASELINE configuration (tools/step_line.py cfg3 | cfg4).
# usage: bash tools/step_kernel_stats.sh <cfg> <out.txt>
cfg=$1; outf=$2
d=/tmp/sks_$cfg
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/step_line.py $cfg > /tmp/sks_$cfg.log 2>&1 < /dev/null
f=$(find $d -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT
if [ -z "$f" ]; then echo "no kernel_stats.csv (see /tmp/sks_$cfg.log)"; tail -5 /tmp/sks_$cfg.log; exit 1; fi
python - "$f" "$cfg" > "$outf" <<'PY'
import csv, sys
f, cfg = sys.argv[1], sys.argv[2]
print("# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/step_line.py %s   (MI355X; one rank's step; avg/min/max in microseconds)" % cfg)
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("tfk::(anonymous namespace)::", "").replace("void tfk::", "")
    print("%-110s calls=%6d avg_us=%9.2f min_us=%9.2f max_us=%9.2f pct=%s" % (name[:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
head -12 "$outf"
