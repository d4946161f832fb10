"""Quick timing of the fused warp kernel (development helper).
env: QW, QH source size; QMODE 0..4 (map mode; 0 = createMap.cl preset cameras); QFMT 0 BGR / 1 NV12."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("QDEV"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import devlib
    vs = devlib.load(None if os.environ["QDEV"] == "1" else os.environ["QDEV"])   # QDEV=1: the development build; QDEV=<path>: that build
else:
    vs = importlib.import_module("video-annotator_amd")
w, h = int(os.environ.get("QW", 3840)), int(os.environ.get("QH", 2160))
mode, fmt = int(os.environ.get("QMODE", 0)), int(os.environ.get("QFMT", 0))
if mode in (0, 5):
    K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
else:
    in_fish, out_fish = mode in (1, 2), mode in (2, 4)
    cw, ch = w, h
    K = vs.lens_camera(1 if in_fish else 0, 150.0 if in_fish else 100.0, w, h)
    Ko = vs.lens_camera(1 if out_fish else 0, 150.0 if out_fish else 100.0, cw, ch)
p = vs.map_params(K, Ko, np.eye(3))
nf = 16
frames = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(nf)]
if fmt == 0:
    outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device="cuda") for _ in range(nf)]
    out_bytes = cw * ch * 3
else:
    outs = [vs.nv12_out_planes(cw, ch) for _ in range(nf)]
    out_bytes = cw * ch + 2 * ((cw + 1) // 2) * ((ch + 1) // 2)
if os.environ.get("QRS"):   # a rotation per output row (rolling shutter): the last row turned by 0.4 degrees about y
    a = np.deg2rad(0.4)
    rb = (np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.asarray(p[8:17], np.float64).reshape(3, 3)).astype(np.float32)
    run = lambda i: vs.warp_nv12_rs(frames[i % nf], p, rb, cw, ch, mode, fmt, out=outs[i % nf])
else:
    run = lambda i: vs.warp_nv12(frames[i % nf], p, cw, ch, mode, fmt, out=outs[i % nf])
for i in range(nf): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
ns = int(os.environ.get("QSTREAMS", 1))   # > 1: consecutive frames on alternating streams (their kernels may overlap)
if ns == 1:
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
else:
    import time
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for i in range(2 * ns):
        with torch.cuda.stream(streams[i % ns]): run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % ns]): run(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
b = w * h * 1.5 + out_bytes
print(f"warp {w}x{h} -> {cw}x{ch} mode {mode} fmt {fmt}: {ms*1000:.1f} us/frame  {b/ms/1e6:.1f} GB/s  ({b/ms/1e6/8000*100:.1f}% of 8 TB/s)")
