import importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle, synth
vs = importlib.import_module("video-annotator_amd")
W, H, RS, N = 640, 360, 5, 40
K = oracle.get_preset_camera(4, W, H)
frames, rots = synth.shaky_clip(3, K, W, H, N, sigma=0.004)
Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
dev = [torch.from_numpy(f).cuda() for f in frames]
stab = vs.Stabilizer(dev, total=N, smooth_radius=RS, seed=11)
outs = []
while True:
    o = stab.pull()
    if o is None: break
    outs.append(o.cpu().numpy())
for i in range(len(outs)):
    p = oracle.map_params(K, Ko, stab.warp_rotation(i))
    exp = oracle.warp_nv12(frames[i + 1], p, cw, ch)
    d = (outs[i] != exp).any(axis=2)
    st = vs.warp_nv12_bgr(dev[i + 1], p, cw, ch).cpu().numpy()
    d2 = (st != exp).any(axis=2)
    # which source frame does the output match?
    match = [j for j in range(N) if np.array_equal(outs[i], oracle.warp_nv12(frames[j], p, cw, ch))] if d.any() and i < 12 else []
    print(i, "pipeline mismatches", int(d.sum()), "stateless mismatches", int(d2.sum()), "matches source frame", match)
