"""Per-SLOT Gauss-Newton iteration history of the pyramidal tracker over consecutive bench frames (CPU, oracle).

The GPU tracker runs one workgroup per feature slot and chains launch k + 1 behind launch k, so today's frame period is
the slowest slot of every frame: mean over frames of (max over slots).  If slow features are NOT the same slots frame
after frame, chaining each slot (or group of slots) only to its own predecessor would let fast frames of a slot absorb its
slow ones: the period would tend to max over groups of (mean over frames of the group's max).  This prints both.
usage: python tools/lk_slot_history.py [frames=40] [w h]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
K = oracle.get_preset_camera(4, w, h)
clip, _ = bench.shaky_ring(torch, torch.device("cpu"), w, h, K, 64, seed=0)   # the bench's ring, rank 0
luma = [np.ascontiguousarray(f[:h].numpy()) for f in clip[:n + 1]]
pts = oracle.good_features(luma[0])
alive = np.ones(len(pts), bool)
hist = np.zeros((n, len(pts)), np.int32)   # iterations (all levels) of slot s in frame pair k; 0 once the slot is lost
for k in range(n):
    nxt, st, it = oracle.pyr_lk_iterations(luma[k], luma[k + 1], pts[alive])
    tot = it.sum(1)
    idx = np.nonzero(alive)[0]
    hist[k, idx] = tot
    pts[idx] = nxt
    alive[idx[st == 0]] = False
    if k - 0 >= 20:   # the reference re-detects after 20 frames (FrameSourceWarp.cpp:415); slots are re-dealt then
        pass
print(f"{w}x{h}, {len(pts)} slots, {n} consecutive frame pairs, {int(alive.sum())} slots alive at the end")
per_frame_max = hist.max(1)
print(f"today's chain: mean over frames of max over slots = {per_frame_max.mean():.1f} iterations (median frame {np.median(per_frame_max):.0f}, max {per_frame_max.max()})")
print(f"median slot-frame: {np.median(hist[hist > 0]):.0f} iterations, p90 {np.percentile(hist[hist > 0], 90):.0f}")
for groups in (1, 2, 4, 8, 16, len(pts)):
    gid = np.arange(len(pts)) * groups // len(pts)
    gmax = np.stack([hist[:, gid == g].max(1) for g in range(groups)], 1)   # (frames, groups): a group's launch lasts as long as its slowest slot
    print(f"  {groups:3d} independent chains: max over groups of mean over frames = {gmax.mean(0).max():.1f} iterations")
slow = (hist >= 30)
print("slots that run >= 30 iterations, by frame:", [np.nonzero(slow[k])[0].tolist() for k in range(n)])
rep = sum(int(len(set(np.nonzero(slow[k])[0]) & set(np.nonzero(slow[k + 1])[0])) > 0) for k in range(n - 1))
print(f"frames whose slow slots include a slow slot of the frame before: {rep} of {n - 1}")
