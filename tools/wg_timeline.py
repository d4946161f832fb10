"""Development helper (tools/dev/libvstab_dev.so, `make -C video-annotator_amd dev`): eight wall-clock stamps per wave of
one launch of the fused warp kernel -> how long each phase of a tile takes, workgroup durations, residency over time.
env: QW, QH, QMODE as tools/quick_warp_time.py."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devlib
vs = devlib.load()
w, h = int(os.environ.get("QW", 3840)), int(os.environ.get("QH", 2160))
mode = int(os.environ.get("QMODE", 0))
K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
p = vs.map_params(K, Ko, np.eye(3))
frames = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(4)]
outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(8): vs.warp_nv12(frames[i % 4], p, cw, ch, mode, 0, out=outs[i % 4])
torch.cuda.synchronize()
nwg = 16384
buf = torch.zeros((nwg, 4, 8), dtype=torch.int64, device="cuda")
vs._L.vstab_dev_set_timing.argtypes = [ctypes.c_void_p]
vs._L.vstab_dev_set_timing(ctypes.c_void_p(buf.data_ptr()))
for i in range(3): vs.warp_nv12(frames[i % 4], p, cw, ch, mode, 0, out=outs[i % 4])   # the last launch's stamps survive
torch.cuda.synchronize()
vs._L.vstab_dev_set_timing(ctypes.c_void_p(0))
t = buf.cpu().numpy().astype(np.float64)
live = t[:, 0, 7] != 0
blk = np.nonzero(live)[0]  # blockIdx.x of every workgroup that ran a tile
t = t[live] / 100.0  # microseconds
t0 = t[:, :, 0].min()
t -= t0
start, end = t[:, :, 0].min(axis=1), t[:, :, 7].max(axis=1)
dur = end - start
print(f"workgroups {len(t)}  span {end.max():.2f} us")
print(f"workgroup duration us: median {np.median(dur):.2f} p10 {np.percentile(dur,10):.2f} p90 {np.percentile(dur,90):.2f} max {dur.max():.2f}; "
      f"sum {dur.sum():.0f} us -> mean residency {dur.sum()/end.max():.0f} workgroups")
names = ["probe + barrier", "load issue", "map", "convert", "barrier", "sample + blend", "store"]
for wv, label in ((0, "wave 0 (probes)"), (1, "wave 1"), (3, "wave 3")):
    d = np.diff(t[:, wv, :], axis=1)
    print(f"  {label}: " + "  ".join(f"{n} {np.median(d[:, k]):.2f} (p90 {np.percentile(d[:, k], 90):.2f})" for k, n in enumerate(names)))
edges = np.linspace(0, end.max(), 25)
for a, b in zip(edges[:-1], edges[1:]):
    m = (a + b) / 2
    print(f"  t={m:6.1f} us  resident {int(((start <= m) & (end > m)).sum()):5d}  started {int(((start >= a) & (start < b)).sum()):5d}")
# per XCD (blockIdx.x % 8: one band of output rows each): when its first / last workgroup ran, and how much work it had
print("per XCD: workgroups, first start, last end, sum of workgroup durations (us)")
for k in range(8):
    m = (blk & 7) == k
    print(f"  xcd {k}: {int(m.sum()):5d}  {start[m].min():6.2f}  {end[m].max():6.2f}  {dur[m].sum():8.1f}   median tile {np.median(dur[m]):.2f}")
print("residency per XCD over time (128 workgroup slots each):")
for m_ in np.linspace(0, end.max(), 13)[1:-1]:
    print(f"  t={m_:5.1f} us  " + " ".join(f"{int((((blk & 7) == k) & (start <= m_) & (end > m_)).sum()):4d}" for k in range(8)))
# dispatch order: is workgroup b ever started before workgroup b - 8 * 128 ... (in-order dispatch across the XCDs?)
order = np.argsort(blk)
st_sorted = start[order]
print(f"start times in blockIdx order: non-decreasing steps {int((np.diff(st_sorted) >= -0.02).sum())} of {len(st_sorted) - 1}; largest step back {(-np.diff(st_sorted)).max():.2f} us")
