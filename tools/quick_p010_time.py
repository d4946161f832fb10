"""Time vstab_warp_p010 (P010 -> BGR16) and vstab_warp_p010_planar (P010 -> P010) alone at 4K (config 5 operators): exact / fp16 blend, with and
without a rotation per row, in the reference kernel's map arithmetic (QMODE, default 5 = VSTAB_MAP_CREATEMAP_CL_OPENCL)."""
import os, sys, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import oracle
if os.environ.get("QDEV"):   # QDEV=1: the development build (tools/dev/libvstab_dev.so) and its VSTAB_PLANAR_* switches
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import devlib
    vs = devlib.load()
else:
    vs = importlib.import_module("video-annotator_amd")
only = os.environ.get("QONLY")   # "planar" / "bgr16": that operator only
w, h = 3840, 2160
K = oracle.get_preset_camera(4, w, h)
Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
rng = np.random.default_rng(0)
ys = [torch.from_numpy((rng.integers(0, 1024, (h, w), dtype=np.uint16) << 6).view(np.int16)).cuda() for _ in range(8)]
us = [torch.from_numpy((rng.integers(0, 1024, (h // 2, w), dtype=np.uint16) << 6).view(np.int16)).cuda() for _ in range(8)]
outs = [torch.empty((ch, cw, 3), dtype=torch.int16, device="cuda") for _ in range(8)]
p = oracle.map_params(K, Ko, oracle.rodrigues((0.004, -0.002, 0.001)))
rb = oracle.map_params(K, Ko, oracle.rodrigues((0.006, -0.001, 0.002)))[8:]
mode = int(os.environ.get("QMODE", "5"))
oy = [torch.empty((ch, cw), dtype=torch.int16, device="cuda") for _ in range(8)]
ouv = [torch.empty(((ch + 1) // 2, 2 * ((cw + 1) // 2)), dtype=torch.int16, device="cuda") for _ in range(8)]
def run(planar, i, rot, blend):
    if planar:
        vs.warp_p010_planar(ys[i], us[i], p, cw, ch, rot, mode, blend, out_y=oy[i], out_uv=ouv[i])
    else:
        vs.warp_p010(ys[i], us[i], p, cw, ch, rot, mode, blend, out=outs[i])
for planar in (False, True):
  if only and only != ("planar" if planar else "bgr16"):
      continue
  for name, blend, rot in (("exact", 0, None), ("fp16", 1, None), ("exact+rs", 0, rb), ("fp16+rs", 1, rb)):
    name = ("planar " if planar else "bgr16  ") + name
    for i in range(8):
        run(planar, i, rot, blend)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 400
    for i in range(n):
        run(planar, i % 8, rot, blend)
    e1.record(); torch.cuda.synchronize()
    us_per = e0.elapsed_time(e1) * 1e3 / n
    algo = w * h * 3 + cw * ch * (3 if planar else 6)
    print(f"{name:16s} {us_per:7.1f} us per 4K frame  -> {algo / us_per / 1e3:7.1f} GB/s algorithmic ({algo/1e6:.1f} MB)")
