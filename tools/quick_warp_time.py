"""Quick 4K timing of the fused warp kernel (development helper)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vs = importlib.import_module("video-annotator_amd")
w, h = int(os.environ.get("QW", 3840)), int(os.environ.get("QH", 2160))
K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
R = np.eye(3)
p = vs.map_params(K, Ko, R)
nf = 16
frames = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(nf)]
outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device="cuda") for _ in range(nf)]
for i in range(nf): vs.warp_nv12_bgr(frames[i], p, cw, ch, out=outs[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for i in range(n): vs.warp_nv12_bgr(frames[i % nf], p, cw, ch, out=outs[i % nf])
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
b = w * h * 1.5 + cw * ch * 3
print(f"warp {w}x{h}: {ms*1000:.1f} us/frame  {b/ms/1e6:.1f} GB/s  ({b/ms/1e6/8000*100:.1f}% of 8 TB/s)")
