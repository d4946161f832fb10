#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${1:-ic}
mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/tools/quick_warp_time.py > $OUT/pmc.log 2>&1 || true
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_warp_tiled" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:24s} {sum(v)/len(v):.4g}")
PY
tail -2 $OUT/pmc.log
