"""Development helper: the prefetch stream in a rocprofv3 kernel trace of bench.py -- per frame: k_pyr_down, gap, k_pyr_down_x2, gap to the next frame's k_pyr_down.
usage: python tools/pyr_stream_gaps.py gpurun_out/prof_<tag>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_pyr_down" in n or "k_pyr_all" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "x2" if "x2" in n else ("all" if "k_pyr_all" in n else "l1")))
rows.sort()
rows = rows[len(rows) // 4: 3 * len(rows) // 4]   # steady state
d1, g1, d2, g2, per = [], [], [], [], []
for i in range(len(rows) - 2):
    a, b, c = rows[i], rows[i + 1], rows[i + 2]
    if a[2] == "l1" and b[2] == "x2" and c[2] == "l1":
        d1.append(a[1] - a[0]); g1.append(b[0] - a[1]); d2.append(b[1] - b[0]); g2.append(c[0] - b[1]); per.append(c[0] - a[0])
    if a[2] == "all" and b[2] == "all":
        d1.append(a[1] - a[0]); g2.append(b[0] - a[1]); per.append(b[0] - a[0])
us = lambda v: (np.mean(v) / 1e3, np.median(v) / 1e3) if len(v) else (0, 0)
print("frames %d: first kernel %.2f/%.2f us (mean/median), gap %.2f/%.2f, second kernel %.2f/%.2f, gap to the next frame's first kernel %.2f/%.2f, start-to-start %.2f/%.2f" %
      ((len(per),) + us(d1) + us(g1) + us(d2) + us(g2) + us(per)))
