"""Isolated timing of the LK tracker at 4K (development helper): run under rocprofv3 --kernel-trace --stats and
read k_lk_track's average; also prints the wall time of the stateless call (which includes the pyramids)."""
import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
vs = importlib.import_module("video-annotator_amd")
import bench
w, h = 3840, 2160
K = vs.get_preset_camera(4, w, h)
frames, _ = bench.shaky_ring(torch, torch.device("cuda"), w, h, K, 6, seed=0)
grays = [f[:h] for f in frames]
pts = vs.good_features(grays[0])
print("features", len(pts))
for i in range(3):
    vs.pyr_lk(grays[0], grays[1], pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for i in range(n):
    out, st = vs.pyr_lk(grays[0], grays[1], pts)
torch.cuda.synchronize()
print("pyr_lk call %.1f us, tracked %d" % ((time.perf_counter() - t0) / n * 1e6, int(st.sum())))
