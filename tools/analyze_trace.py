"""Timeline analysis of a rocprofv3 kernel trace of bench.py (development helper).
usage: python tools/analyze_trace.py gpurun_out/prof_<tag>"""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "vstab::" not in n:
        continue
    short = n.split("vstab::")[1].split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Queue_Id"])))
rows.sort()
t0 = rows[0][0]
byk = collections.defaultdict(list)
for s, e, k, q in rows:
    byk[k].append((s - t0, e - t0))
# steady-state window: 200 warp launches in the middle of the run (the end of a run is the drain of the look-ahead queue)
w = byk["k_warp_fused"] if "k_warp_fused" in byk else byk["k_warp_tiled"]
mid = len(w) // 2
lo, hi = w[mid - 100][0], w[mid + 100][1]
print("window %.1f us, %d warps -> period %.2f us" % ((hi - lo) / 1e3, 200, (w[mid + 100][0] - w[mid - 100][0]) / 200 / 1e3))
for k, v in byk.items():
    vv = [(s, e) for s, e in v if s >= lo and e <= hi]
    if not vv:
        continue
    d = np.array([e - s for s, e in vv]) / 1e3
    busy = d.sum() / ((hi - lo) / 1e3)
    starts = np.array([s for s, e in vv])
    gaps = (starts[1:] - np.array([e for s, e in vv])[:-1]) / 1e3
    print(f"{k:22s} n={len(vv):4d} avg={d.mean():7.2f} us  busy={busy*100:5.1f}%  gap_to_next(avg/med/min)={gaps.mean():7.2f}/{np.median(gaps):7.2f}/{gaps.min():7.2f}")
# union busy time of all kernels
ev = sorted((s, e) for k, v in byk.items() for s, e in v if s >= lo and e <= hi)
tot, cur_s, cur_e = 0, *ev[0]
for s, e in ev[1:]:
    if s > cur_e:
        tot += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
tot += cur_e - cur_s
print("GPU busy (any kernel) %.1f%% of the window" % (tot / (hi - lo) * 100))
# print a sample of the timeline
print("sample timeline (us):")
base = w[mid][0]
for s, e, k, q in rows:
    s -= t0; e -= t0
    if s >= base and s < base + 200000:
        print(f"  {(s-base)/1e3:8.1f} -> {(e-base)/1e3:8.1f}  {k}")
