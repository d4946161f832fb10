"""Timing of the warp with a pre-written quantised map (development helper).  env: QW, QH."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vs = importlib.import_module("video-annotator_amd")
w, h = int(os.environ.get("QW", 3840)), int(os.environ.get("QH", 2160))
K = vs.get_preset_camera(4, w, h); Ko, (cw, ch) = vs.get_output_camera(K, w, h)
p = vs.map_params(K, Ko, np.eye(3))
q = vs.quantised_map(p, cw, ch)
nf = 16
frames = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(nf)]
outs = [torch.empty((ch, cw, 3), dtype=torch.uint8, device="cuda") for _ in range(nf)]
for name, fn in (("direct", lambda i: vs.warp_nv12(frames[i % nf], p, cw, ch, out=outs[i % nf])),
                 ("mapped", lambda i: vs.warp_nv12_mapped(frames[i % nf], q, cw, ch, out=outs[i % nf]))):
    for i in range(nf): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): fn(i)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} {w}x{h}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us/frame")
