"""Development helper (tools/dev/libvstab_dev.so): per-feature wall-clock stamps of the tracker launches of a bench-shaped 4K
pipeline run -> where a feature's time goes (start point, staging, per level: derivatives + patch matrix, iterations)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import numpy as np, torch
import devlib
vs = devlib.load()
sys.modules["video-annotator_amd"] = vs
import bench
dev = torch.device("cuda:0")
w, h = 3840, 2160
K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
clip, _ = bench.shaky_ring(torch, dev, w, h, K, 64, seed=0)
alone = bool(int(os.environ.get("LK_ALONE", "0")))   # 1: pull without the warp's competition is not possible -- instead VSTAB tracking only
stab = vs.Stabilizer(clip, total=5000, preset=4, smooth_radius=30, seed=1234)
outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
for i in range(600): assert stab.pull_into(outs[i % 8])
buf = torch.zeros((64, 256, 32), dtype=torch.int64, device=dev)
vs._L.vstab_dev_set_lk_timing.argtypes = [ctypes.c_void_p]
vs._L.vstab_dev_set_lk_timing(ctypes.c_void_p(buf.data_ptr()))
for i in range(48): assert stab.pull_into(outs[i % 8])
torch.cuda.synchronize()
vs._L.vstab_dev_set_lk_timing(ctypes.c_void_p(0))
for i in range(100): assert stab.pull_into(outs[i % 8])
t = buf.cpu().numpy().astype(np.float64)
live = (t[:, :, 15] != 0) & (t[:, :, 2] != 0)
print("launches with stamps:", int(live.any(1).sum()), " features per launch:", int(live.sum(1).max()))
us = lambda a: a / 100.0
rows = t[live]
tot = us(rows[:, 15] - rows[:, 0])
print(f"feature time us: median {np.median(tot):.1f} p90 {np.percentile(tot, 90):.1f} max {tot.max():.1f}")
print(f"  start point (chain record / host point): median {np.median(us(rows[:, 1] - rows[:, 0])):.2f}")
print(f"  staging (prev neighbourhoods of all levels + top next block): median {np.median(us(rows[:, 2] - rows[:, 1])):.2f}")
prev = rows[:, 2]
for i, lvl in enumerate((3, 2, 1, 0)):
    a, b, n = rows[:, 3 + 3 * i], rows[:, 4 + 3 * i], rows[:, 5 + 3 * i]
    ok = a != 0
    setup, it = us(a - prev)[ok], us(b - a)[ok]
    per = it[n[ok] > 0] / n[ok][n[ok] > 0]
    print(f"  level {lvl}: derivatives + patch matrix median {np.median(setup):.2f} us; iterations median {np.median(n[ok]):.0f} (max {n[ok].max():.0f}), "
          f"{np.median(it):.2f} us, {np.median(per):.2f} us per iteration")
    prev = np.where(b != 0, b, prev)
print(f"  tail (record): median {np.median(us(rows[:, 15] - prev)):.2f}")
# per launch: duration = max over features
for L in range(64):
    if live[L].any():
        r = t[L][live[L]]
        d = us(r[:, 15].max() - r[:, 0].min())
        print(f"  launch {L}: {int(live[L].sum())} features, span {d:.1f} us, slowest feature {us(r[:, 15] - r[:, 0]).max():.1f} us, median feature {np.median(us(r[:, 15] - r[:, 0])):.1f}") if L % 8 == 0 else None

# sub-phase stamps of the development build (slots 16 .. 31): pair index inside the segment, blocks fetched ahead that were used, and per
# level the barrier at its top and the end of the fetch code
if rows.shape[1] >= 32:
    fi = rows[:, 16]
    for sel, name in ((fi == 0, "first pair of a segment"), (fi > 0, "later pairs")):
        r = rows[sel]
        if not len(r): continue
        print(f"{name}: {len(r)} feature-frames, median {np.median(us(r[:, 15] - r[:, 0])):.1f} us; staging {np.median(us(r[:, 2] - r[:, 1])):.2f}; "
              f"top neighbourhood ahead used {int((r[:, 17].astype(np.int64) & 1).sum())}, top block ahead {int(((r[:, 17].astype(np.int64) >> 1) & 1).sum())}")
        prev = r[:, 2]
        for i, lvl in enumerate((3, 2, 1, 0)):
            a, b = r[:, 3 + 3 * i], r[:, 4 + 3 * i]
            bar, fetch = r[:, 18 + 3 * i], r[:, 19 + 3 * i]
            ok = (a != 0) & (bar != 0)
            print(f"   level {lvl}: previous level end -> barrier passed {np.median(us(bar - prev)[ok]):.2f}, fetch code {np.median(us(fetch - bar)[ok]):.2f}, "
                  f"-> iterations start {np.median(us(a - fetch)[ok]):.2f}")
            prev = np.where(b != 0, b, prev)
