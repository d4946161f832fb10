"""Isolated timings of the tracking kernels at 4K (development helper)."""
import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
vs = importlib.import_module("video-annotator_amd")
w, h = 3840, 2160
imgs = [torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda") for _ in range(8)]
def timeit(fn, n=100):
    for _ in range(5): fn(0)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
l1 = [vs.pyr_down(im) for im in imgs]; l2 = [vs.pyr_down(im) for im in l1]
print("pyr_down L0->L1 %.1f us" % timeit(lambda i: vs.pyr_down(imgs[i % 8])))
print("pyr_down L1->L2 %.1f us" % timeit(lambda i: vs.pyr_down(l1[i % 8])))
print("pyr_down L2->L3 %.1f us" % timeit(lambda i: vs.pyr_down(l2[i % 8])))
nv = [torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, device="cuda") for _ in range(8)]
print("pack_nv12 %.1f us" % timeit(lambda i: vs.pack_nv12(nv[i % 8][:h], nv[i % 8][h:])))
print("min_eig %.1f us (incl. alloc+sync)" % timeit(lambda i: vs.min_eig(imgs[i % 8]), 20))
