#!/bin/bash
# usage (GPU box): tools/prof_detect.sh   -- per-kernel times + SQ counters of the fused and the two-pass corner detector (4K and 1080p)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_detect
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/detect_time.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/detect_time.py > $OUT/pmc_sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "vstab::" in r["Name"]:
            lines.append(f"stats: {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):10.1f} min={r['MinNs']} max={r['MaxNs']}")
# per-size split of the kernel trace (grid size tells 4K from 1080p)
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "vstab::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    lines.append(f"trace: {k[0]:32s} grid={k[1]}x{k[2]:6s} n={len(v):3d} median_us={v[len(v)//2]:8.1f} min_us={v[0]:8.1f}")
pm = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "vstab::" in r["Kernel_Name"]:
            pm[(r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(pm.items()):
    lines.append(f"pmc: {k[0]:32s} grid={k[1]:9s} {k[2]:22s} mean={sum(v)/len(v):.5g}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
