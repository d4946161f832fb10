"""Development helper: which kernels beside it make the fused warp slower?  From a rocprofv3 kernel trace of bench.py: every warp launch's duration against
the fraction of it during which each of the other kernels was running (least squares).  usage: python tools/warp_overlap_regression.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "vstab::" not in n: continue
    k = n.split("vstab::")[1].split("(")[0].split("<")[0]
    ev.setdefault(k, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k in ev: ev[k].sort()
warps = ev["k_warp_fused"]
warps = warps[len(warps) // 4: 3 * len(warps) // 4]
others = [k for k in ev if k != "k_warp_fused"]
def overlap(iv, s, e):
    a = np.array(iv); lo = np.maximum(a[:, 0], s); hi = np.minimum(a[:, 1], e)
    return np.clip(hi - lo, 0, None).sum()
X, y = [], []
for s, e in warps:
    X.append([overlap(ev[k], s, e) / (e - s) for k in others] + [1.0]); y.append((e - s) / 1e3)
X, y = np.array(X), np.array(y)
coef, *_ = np.linalg.lstsq(X, y, rcond=None)
print(f"{len(y)} warp launches, mean {y.mean():.2f} us, median {np.median(y):.2f}, min {y.min():.2f}")
print(f"  alone (intercept): {coef[-1]:.2f} us")
for i, k in enumerate(others):
    print(f"  {k:18s}: running beside the warp {X[:, i].mean() * 100:5.1f} % of its time; a warp that has it beside it ALL the time is {coef[i]:+.2f} us longer -> {coef[i] * X[:, i].mean():+.2f} us on average")
print(f"  residual rms {np.sqrt(np.mean((X @ coef - y) ** 2)):.2f} us")
# quartiles of warp duration by tracker overlap
i = others.index("k_lk_track")
for lo, hi in ((0, 0.05), (0.05, 0.5), (0.5, 0.95), (0.95, 1.01)):
    m = (X[:, i] >= lo) & (X[:, i] < hi)
    if m.any(): print(f"  tracker beside the warp {lo * 100:.0f}-{min(hi, 1) * 100:.0f} % of the time: {m.sum()} launches, mean {y[m].mean():.2f} us")
