#!/bin/bash
# usage (GPU box): tools/prof_p010.sh   -- rocprofv3 kernel stats + SQ counters of vstab_warp_p010 alone at 4K (tools/quick_p010_time.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_p010
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/quick_p010_time.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/quick_p010_time.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/quick_p010_time.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/quick_p010_time.py > $OUT/pmc_write.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_warp" in r["Name"]:
            lines.append(f"stats: {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):10.1f} min={r['MinNs']} max={r['MaxNs']}")
for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    pm = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_warp" in r["Kernel_Name"]:
                pm[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(pm.items()):
        lines.append(f"pmc[{d}]: {k[0]:46s} {k[1]:22s} mean_per_launch={sum(v)/len(v):.5g}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
