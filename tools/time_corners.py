import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
vs = importlib.import_module("video-annotator_amd")
dev = torch.device("cuda:0")
w, h = 3840, 2160
K = vs.get_preset_camera(4, w, h)
frames, _ = bench.shaky_ring(torch, dev, w, h, K, 2, 0)
g = frames[0][:h]
for _ in range(3): pts = vs.good_features(g)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): pts = vs.good_features(g)
print("good_features 4K: %.1f us, corners %d" % ((time.perf_counter() - t) / 10 * 1e6, len(pts)))
e = vs.min_eig(g)
mx = float(e.max()); thr = mx * 0.01
print("max", mx, "pixels > thr", int((e > thr).sum()), "of", e.numel())
import torch.nn.functional as F
m = F.max_pool2d(e[None, None], 3, 1, 1)[0, 0]
cand = ((e == m) & (e > thr)); cand[0] = cand[-1] = False; cand[:, 0] = cand[:, -1] = False
print("candidates", int(cand.sum()))
