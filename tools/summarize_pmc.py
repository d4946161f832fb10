"""Summarise rocprofv3 csv output directories (kernel-trace stats + pmc passes) for one kernel."""
import csv, glob, os, sys, collections
out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_warp"
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r.get("Name", ""):
            lines.append(f"stats: {r['Name'][:60]} calls={r['Calls']} avg_ns={r['AverageNs']} min={r['MinNs']} max={r['MaxNs']}")
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    lines.append(f"pmc: {k:24s} mean_per_dispatch={sum(v)/len(v):.4g}  n={len(v)}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
