"""Development helper: isolated timings of the pyramid kernels (preallocated outputs, back-to-back launches)."""
import ctypes, importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
vs = importlib.import_module("video-annotator_amd")
L = vs._L
def timeit(fn, n=200):
    for _ in range(10): fn(0)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
vp = lambda t: ctypes.c_void_p(t.data_ptr())
for (w, h) in ((3840, 2160), (1920, 1080)):
    imgs = [torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda") for _ in range(8)]
    w1, h1 = (w + 1) // 2, (h + 1) // 2; w2, h2 = (w1 + 1) // 2, (h1 + 1) // 2; w3, h3 = (w2 + 1) // 2, (h2 + 1) // 2
    l1 = torch.empty((h1, w1), dtype=torch.uint8, device="cuda"); l2 = torch.empty((h2, w2), dtype=torch.uint8, device="cuda"); l3 = torch.empty((h3, w3), dtype=torch.uint8, device="cuda")
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def a(i): assert L.vstab_pyr_down(vp(imgs[i % 8]), ctypes.c_size_t(w), w, h, vp(l1), ctypes.c_size_t(w1), s) == 0
    def b(i): assert L.vstab_pyr_down_x2(vp(l1), ctypes.c_size_t(w1), w1, h1, vp(l2), ctypes.c_size_t(w2), vp(l3), ctypes.c_size_t(w3), s) == 0
    a(0)
    print(f"{w}x{h}: level 1 {timeit(a):.1f} us, levels 2+3 {timeit(b):.1f} us (alone, back to back)")
