"""Summarise tools/prof_bench.sh output: per-kernel stats, PMC means per launch, calibrated HBM bytes."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
WARP = os.environ.get("WARP_KERNEL", "k_warp_fused")   # k_warp_planar for the plane-wise warp (its staging is LDS-DMA: 16 B per lane)
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "vstab::" in r["Name"]:
            lines.append(f"stats: {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):10.1f} min={r['MinNs']} max={r['MaxNs']} pct={r['Percentage']}")
def pmc(dirname):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "vstab::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
res = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "cal_fetch", "cal_write"):
    for (k, c), v in sorted(pmc(d).items()):
        lines.append(f"pmc[{d}]: {k:42s} {c:12s} mean_per_launch={v:.5g} (KB)")
        res[(d, k, c)] = v
def find(d, frag, c):
    for (dd, k, cc), v in res.items():
        if dd == d and frag in k and cc == c:
            return v
    return None
known = 3840 * 2160 * 1.5
cal16 = find("cal_fetch", ", 4u>", "FETCH_SIZE") or find("cal_fetch", "uint4", "FETCH_SIZE")
cal4 = find("cal_fetch", "k_pack_nv12<unsigned int>", "FETCH_SIZE")
cal8 = find("cal_fetch", ", 2u>", "FETCH_SIZE") or find("cal_fetch", "uint2", "FETCH_SIZE")
calw16 = find("cal_write", ", 4u>", "WRITE_SIZE") or find("cal_write", "uint4", "WRITE_SIZE")
wf, ww = find("pmc_fetch", WARP, "FETCH_SIZE"), find("pmc_write", WARP, "WRITE_SIZE")
summary = {"known_copy_bytes": known, "round": os.environ.get("ROUND", "r05"), "map_precision": os.environ.get("MAP_PRECISION", "opencl"), "kernel": WARP}
if cal16: summary["fetch_factor_16B_per_lane"] = known / (cal16 * 1024)
if cal4: summary["fetch_factor_4B_per_lane"] = known / (cal4 * 1024)
if cal8: summary["fetch_factor_8B_per_lane"] = known / (cal8 * 1024)
if calw16: summary["write_factor_16B_per_lane"] = known / (calw16 * 1024)
if wf and ww:
    # the BGR kernel stages with 8 B/lane loads, the plane-wise kernel with 16 B/lane LDS-DMA loads
    ff = (summary.get("fetch_factor_16B_per_lane", 1.0) if WARP == "k_warp_planar" else summary.get("fetch_factor_8B_per_lane", summary.get("fetch_factor_4B_per_lane", 1.0)))
    summary["warp_fetch_bytes_raw"] = wf * 1024
    summary["warp_write_bytes_raw"] = ww * 1024
    summary["warp_fetch_bytes_corrected"] = wf * 1024 * ff
    summary["hbm_bytes_per_launch"] = wf * 1024 * ff + ww * 1024
    summary["algorithmic_bytes_per_launch"] = 3840 * 2160 * 1.5 + (3524 * 1999 + 2 * 1762 * 1000 if WARP == "k_warp_planar" else 3524 * 1999 * 3)
# SQ counters of the warp kernel in the pipeline run -> VALU busy fraction (quad-cycle counters; 1024 SIMDs; shader clock
# under this load 2.1 GHz, measured in-kernel with s_memtime / s_memrealtime: tools/wg_timeline.py)
sq = {c: find("pmc_sq", WARP, c) for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if WARP in r["Name"]:
            summary["rocprof_avg_launch_us"] = round(float(r["AverageNs"]) / 1e3, 2)
if sq["SQ_ACTIVE_INST_VALU"] and "rocprof_avg_launch_us" in summary:
    clk = float(os.environ.get("SHADER_CLOCK_GHZ", "2.1")) * 1e9   # ASSUMED unless given: the clock tools/wg_timeline.py measured in-kernel (r02: 2.10 GHz)
    summary["shader_clock_ghz_assumed"] = clk / 1e9
    summary["valu_busy"] = round(sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * summary["rocprof_avg_launch_us"] * 1e-6 * clk), 3)
    summary["sq"] = {k: v for k, v in sq.items() if v is not None}
lines.append("traffic: " + json.dumps(summary))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("\n".join(lines))
