"""Gauss-Newton iterations per feature of the pyramidal tracker on bench-shaped frame pairs (CPU, oracle): the GPU tracker
runs one workgroup per feature, so a launch lasts as long as its slowest feature."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import oracle, synth
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
K = oracle.get_preset_camera(4, W, H)
frames, _ = synth.shaky_clip(3, K, W, H, 6, sigma=0.3 * np.pi / 180)
for k in range(5):
    g0, g1 = np.ascontiguousarray(frames[k][:H]), np.ascontiguousarray(frames[k + 1][:H])
    pts = oracle.good_features(g0)
    _, st, it = oracle.pyr_lk_iterations(g0, g1, pts)
    tot = it.sum(1)
    print(f"pair {k}: {len(pts)} features, iterations per feature: median {int(np.median(tot))}, p90 {int(np.percentile(tot, 90))}, max {tot.max()}"
          f" (levels {it[tot.argmax()].tolist()}); features at the 30-iteration limit on some level: {(it == 30).any(1).sum()}; tracked {int(st.sum())}")
