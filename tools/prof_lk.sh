#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${1:-lk}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/time_lk.py > $OUT/trace.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vstab::" in r["Name"]:
            print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f}")
PY
grep -E "features|pyr_lk" $OUT/trace.log
