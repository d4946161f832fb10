"""Development helper (library built with make DEV=-DVSTAB_DEV): per-workgroup entry / exit stamps of one launch of the
fused warp kernel -> shader clock, workgroup durations, residency over time."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vs = importlib.import_module("video-annotator_amd")
w, h = int(os.environ.get("QW", 3840)), int(os.environ.get("QH", 2160))
K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
p = vs.map_params(K, Ko, np.eye(3))
frames = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(4)]
outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(8): vs.warp_nv12(frames[i % 4], p, cw, ch, 0, 0, out=outs[i % 4])
torch.cuda.synchronize()
nwg = 8192
buf = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")
vs._L.vstab_dev_set_timing.argtypes = [ctypes.c_void_p]
vs._L.vstab_dev_set_timing(ctypes.c_void_p(buf.data_ptr()))
for i in range(3): vs.warp_nv12(frames[i % 4], p, cw, ch, 0, 0, out=outs[i % 4])   # the last launch's stamps survive
torch.cuda.synchronize()
vs._L.vstab_dev_set_timing(ctypes.c_void_p(0))
t = buf.cpu().numpy()
t = t[t[:, 2] != 0]
rt0, ck0, rt1, ck1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2].astype(np.float64), t[:, 3].astype(np.float64)
dur_us = (rt1 - rt0) / 100.0  # memrealtime ticks at 100 MHz
clk = (ck1 - ck0) / np.maximum(rt1 - rt0, 1) * 100.0  # MHz
start = (rt0 - rt0.min()) / 100.0; end = (rt1 - rt0.min()) / 100.0
print(f"workgroups {len(t)}  span {end.max():.1f} us  shader clock median {np.median(clk):.0f} MHz (p10 {np.percentile(clk,10):.0f}, p90 {np.percentile(clk,90):.0f})")
print(f"workgroup duration us: median {np.median(dur_us):.2f} p10 {np.percentile(dur_us,10):.2f} p90 {np.percentile(dur_us,90):.2f} max {dur_us.max():.2f}; sum {dur_us.sum():.0f} us -> mean residency {dur_us.sum()/end.max():.0f} workgroups")
edges = np.linspace(0, end.max(), 25)
for a, b in zip(edges[:-1], edges[1:]):
    m = (a + b) / 2
    print(f"  t={m:6.1f} us  resident {int(((start <= m) & (end > m)).sum()):5d}  started {int(((start >= a) & (start < b)).sum()):5d}")
# per-queue view (tile index -> queue: the 16-chunk pairing of k_warp_fused)
rows = int(os.environ.get("VSTAB_ROWS", 8 if ((cw + 63) // 64) * ((ch + 31) // 32) >= 1536 else 4))
tx, ty = (cw + 63) // 64, (ch + 4 * rows - 1) // (4 * rows)
idx_all = np.nonzero(buf.cpu().numpy()[:, 2])[0]
trow = idx_all // tx
chunk = np.zeros(ty, dtype=int)
for c in range(16):
    chunk[(c * ty) >> 4:((c + 1) * ty) >> 4] = c
pair = {}
for k in range(8):
    heavy, light = (7 - k, k) if k < 4 else (k + 4, 19 - k)
    pair[heavy] = k; pair[light] = k
qq = np.array([pair[chunk[r]] for r in trow])
for k in range(8):
    m = qq == k
    print(f"  queue {k}: tiles {m.sum():4d}  sum {dur_us[m].sum():7.0f} us  mean {dur_us[m].mean():5.2f}  last end {end[m].max():5.1f} us  max {dur_us[m].max():5.1f}")
for c in range(16):
    m = np.array([chunk[r] == c for r in trow])
    print(f"  chunk {c:2d}: tiles {m.sum():4d} mean dur {dur_us[m].mean():5.2f} us")
order = np.argsort(-dur_us)[:12]
print("slowest tiles (tile_x, tile_y, start us, duration us):", [(int(idx_all[i] % tx), int(idx_all[i] // tx), round(float(start[i]), 1), round(float(dur_us[i]), 1)) for i in order])
late = np.argsort(-end)[:12]
print("last tiles to end (tile_x, tile_y, start us, duration us):", [(int(idx_all[i] % tx), int(idx_all[i] // tx), round(float(start[i]), 1), round(float(dur_us[i]), 1)) for i in late])
