"""Where does the frame period go?  Times the pull loop at 4K with the warp output shrunk / tracking off
(development helper).  usage: python tools/pipe_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
vs = importlib.import_module("video-annotator_amd")
import bench
w, h = 3840, 2160
dev = torch.device("cuda")
K = vs.lens_camera(1, 118.0 * 1.2, w, h)
clip, _ = bench.shaky_ring(torch, dev, w, h, K, 64, seed=0)


def run(name, steps=600, warm=64, **cfg):
    stab = vs.Stabilizer(clip, total=steps + warm + 200, smooth_radius=30, seed=1, lens_mode=1, in_projection=1, out_projection=0,
                         in_dfov=118.0 * 1.2, out_dfov=100.0, **cfg)
    ow, oh = stab.out_size
    outs = [torch.empty((oh, ow, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
    for i in range(warm):
        stab.pull_into(outs[i % 8])
    torch.cuda.synchronize()
    p0 = stab.profile()
    t0 = time.perf_counter()
    for i in range(steps):
        stab.pull_into(outs[i % 8])
    el_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    p1 = stab.profile()
    d = {k: (p1[k] - p0[k]) / steps * 1e3 for k in p1 if k.startswith("host_")}
    print(f"{name:34s} {el / steps * 1e6:7.1f} us/frame (host loop {el_host / steps * 1e6:5.1f})   out {ow}x{oh}   " + " ".join(f"{k[5:-3]}={v:.1f}" for k, v in d.items()), flush=True)
    stab.close()


run("full 4K out")
run("tiny out (64x36)", out_width=64, out_height=36)
run("tracking off, 4K out", tracking=0)
run("tracking off, tiny out", tracking=0, out_width=64, out_height=36)
