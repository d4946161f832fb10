"""Run both corner detectors at 4K and 1080p a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import importlib
import torch
import synth
vs = importlib.import_module("video-annotator_amd")
for w, h in ((3840, 2160), (1920, 1080)):
    import numpy as np
    g = torch.from_numpy(np.ascontiguousarray(synth.luma(41, w, h, rects=400))).cuda()
    for det in (vs.DETECTOR_AUTO, vs.DETECTOR_TWO_PASS):
        info = {}
        for _ in range(20):
            c = vs.good_features(g, detector=det, info=info)
        print(w, h, "detector", info["detector_used"], "corners", len(c))
