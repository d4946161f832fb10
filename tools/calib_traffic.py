"""PMC calibration: kernels with a KNOWN byte count in the two access widths the pipeline uses
(16 B/lane, 8 B/lane and 4 B/lane coalesced copies of one 4K NV12 frame = 12,441,600 B read + written)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vs = importlib.import_module("video-annotator_amd")
w, h = 3840, 2160
src = [torch.randint(0, 256, (h * 3 // 2, w + 16), dtype=torch.uint8, device="cuda") for _ in range(24)]
for it in range(24):
    f = src[it]
    vs.pack_nv12(f[:h, :w], f[h:, :w])            # pitch 3856 (16-B aligned) -> k_pack_nv12<uint4>
for it in range(24):
    f = src[it]
    vs.pack_nv12(f[:h, 8:w + 8], f[h:, 8:w + 8])  # base offset 8 -> k_pack_nv12<uint2> (the warp kernel's staging width)
for it in range(24):
    f = src[it]
    vs.pack_nv12(f[:h, 4:w + 4], f[h:, 4:w + 4])  # base offset 4 -> k_pack_nv12<unsigned int>
torch.cuda.synchronize()
print("calibration launches done")
