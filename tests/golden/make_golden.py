"""Generates the committed golden vectors.  Run in the BUILD container (needs /root/reference for
the createMap.cl build in oracle/_ref):  python tests/golden/make_golden.py

  createmap_ref.npz   outputs of the reference's OWN createMap.cl (oracle/_ref build) -- the only
                      executable ground truth the reference offers.
  oracle_kat.npz      known-answer vectors of the CPU restatement (oracle/), pinning it against
                      accidental change and giving the GPU tests a fixture that does not depend
                      on rebuilding the oracle.
  planar_kat.npz      the same for the plane-wise NV12 -> NV12 / P010 -> P010 warp (round 5; the definition is this
                      repository's: include/vstab.h VSTAB_OUT_NV12_PLANAR).  `python tests/golden/make_golden.py planar`
                      writes this file alone.

Fixtures are data only: inputs (seeds / parameters) and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
import synth  # noqa: E402

ROTS = [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3), (0.6, -0.4, 0.2)]


def cameras(w, h, preset=oracle.GOPRO_H4B_WIDE169_MEASURED, scale=1.0):
    K = oracle.get_preset_camera(preset, w, h)
    Ko, size = oracle.get_output_camera(K, w, h, scale)
    return K, Ko, size


def make_createmap_ref():
    out = {}
    # (1) whole small maps: 128x72 input camera
    K, Ko, (cw, ch) = cameras(128, 72)
    out["small_size"] = np.array([cw, ch])
    for i, rv in enumerate(ROTS):
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        mx, my = oracle.create_map_ref(p, cw, ch)
        out[f"small_params_{i}"], out[f"small_mapx_{i}"], out[f"small_mapy_{i}"] = p, mx, my
    # (2) 4K config 3: 32x32 crops (corners, centre, optical-axis neighbourhood)
    K, Ko, (cw, ch) = cameras(3840, 2160)
    out["uhd_size"] = np.array([cw, ch])
    crops = [(0, 0), (cw - 32, 0), (0, ch - 32), (cw - 32, ch - 32), (cw // 2 - 16, ch // 2 - 16),
             (int(Ko[0, 2]) - 16, int(Ko[1, 2]) - 16), (700, 1500)]
    out["uhd_crops"] = np.array(crops)
    for i, rv in enumerate(ROTS[:3]):
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        mx, my = oracle.create_map_ref(p, cw, ch)
        out[f"uhd_params_{i}"] = p
        out[f"uhd_mapx_{i}"] = np.stack([mx[y:y + 32, x:x + 32] for x, y in crops])
        out[f"uhd_mapy_{i}"] = np.stack([my[y:y + 32, x:x + 32] for x, y in crops])
    np.savez_compressed(os.path.join(HERE, "createmap_ref.npz"), **out)
    print("createmap_ref.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_oracle_kat():
    out = {}
    # cvtColor: exhaustive-ish sweep, Y in 16 steps x all (U,V) on a 17-grid, as a 2-row NV12 frame
    ys = np.arange(0, 256, 5, dtype=np.uint8)
    uvals = np.linspace(0, 255, 18).astype(np.uint8)
    cols = []
    for u in uvals:
        for v in uvals:
            cols.append((u, v))
    w = 2 * len(cols)
    frames = []
    for yv in ys:
        f = np.empty((3, w), np.uint8)
        f[0:2] = yv
        f[2, 0::2] = [c[0] for c in cols]
        f[2, 1::2] = [c[1] for c in cols]
        frames.append(f)
    nv = np.concatenate([np.concatenate([f[0:2] for f in frames]), np.concatenate([f[2:3] for f in frames])])
    out["cvt_nv12"] = nv
    out["cvt_bgr"] = oracle.cvt_nv12_bgr(nv)
    # fused warp on a seeded 128x72 frame, 4 rotations (border + far-outside pixels included)
    K, Ko, (cw, ch) = cameras(128, 72)
    frame = synth.nv12(11, 128, 72)
    out["warp_seed"] = np.array([11, 128, 72])
    for i, rv in enumerate(ROTS):
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        out[f"warp_params_{i}"] = p
        out[f"warp_bgr_{i}"] = oracle.warp_nv12(frame, p, cw, ch)
    # remap special cases: NaN / inf / huge / exact half-bucket coordinates on a 16x9 gray image
    src = synth.luma(3, 16, 9, rects=3)
    mx = np.array([[np.nan, np.inf, -np.inf, 3e9, -3e9, 1e30, -0.5, -1.0, -1.015625, 15.0, 15.5, 16.0, 7.015625, 7.046875, 0.0, 14.984375]], np.float32)
    my = np.array([[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, -0.5, 3.0, 3.0, 8.0, 8.5, 9.0, 2.515625, 2.546875, 0.0, 7.984375]], np.float32)
    out["remap_src"], out["remap_mx"], out["remap_my"] = src, mx, my
    out["remap_dst"] = oracle.remap_bilinear(src, mx, my)
    # cameras
    cams = []
    for preset, w_, h_, sc, crop in [(4, 1920, 1080, 1.0, 0), (4, 1920, 1080, 0.5, 0), (4, 1920, 1080, 1.0, 1),
                                     (4, 3840, 2160, 1.0, 0), (5, 3840, 2160, 1.0, 0), (3, 3840, 2160, 1.0, 0),
                                     (1, 1920, 1440, 0.5, 0), (0, 1920, 1440, 1.0, 0), (2, 2704, 2028, 1.0, 1)]:
        K = oracle.get_preset_camera(preset, w_, h_)
        Ko, sz = oracle.get_output_camera(K, w_, h_, sc, bool(crop), 1.0)
        cams.append(np.concatenate([[preset, w_, h_, sc, crop], K.reshape(-1), Ko.reshape(-1), sz]))
    out["cameras"] = np.array(cams)
    # SG filter: weights for m = 30 and a filtered seeded trajectory incl. zero-filled start-up
    out["sg_w30"] = oracle.sg_weights(30)
    rng = np.random.default_rng(5)
    m = 5
    filt = oracle.RotationFilter(m)
    R = np.eye(3)
    traj, outs = [], []
    for _ in range(40):
        R = oracle.rodrigues(rng.normal(0, 0.01, 3)) @ R
        filt.add(R)
        traj.append(R.copy())
        outs.append(filt.filter())
    out["sg_m"] = np.array([m])
    out["sg_traj"], out["sg_filtered"] = np.array(traj), np.array(outs)
    # corner detector + LK on seeded 320x180 frames
    g0 = synth.luma(21, 320, 180)
    out["gftt_seed"] = np.array([21, 320, 180])
    pts = oracle.good_features(g0, 200, 0.01, 30.0)
    out["gftt_corners"] = pts
    g1 = synth.shifted(g0, 1.7, -2.3)
    nxt, st = oracle.pyr_lk(g0, g1, pts)
    out["lk_shift"] = np.array([1.7, -2.3])
    out["lk_next"], out["lk_status"] = nxt, st
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **out)
    print("oracle_kat.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_planar_kat():
    """Known answers of the PLANE-WISE warp's definition (oracle/vstab_oracle.c: vo_warp_planar_mapped; include/vstab.h, VSTAB_OUT_NV12_PLANAR):
    a seeded 128x72 NV12 frame and its 10-bit twin (low six bits of every word junk), four rotations, both 10-bit blends."""
    sys.path.insert(0, os.path.dirname(HERE))
    from test_p010_cpu import p010_frame
    out = {}
    K, Ko, (cw, ch) = cameras(128, 72)
    frame = synth.nv12(11, 128, 72)
    y16, uv16, _, _ = p010_frame(11, 128, 72)
    out["seed"] = np.array([11, 128, 72])
    out["p010_y"], out["p010_uv"] = y16, uv16
    for i, rv in enumerate(ROTS):
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        out[f"params_{i}"] = p
        out[f"nv12_y_{i}"], out[f"nv12_uv_{i}"] = oracle.warp_nv12_planar(frame, p, cw, ch, 0)
        for blend in (0, 1):
            out[f"p010_y_{i}_{blend}"], out[f"p010_uv_{i}_{blend}"] = oracle.warp_p010_planar(y16, uv16, p, cw, ch, 0, None, blend)
    np.savez_compressed(os.path.join(HERE, "planar_kat.npz"), **out)
    print("planar_kat.npz", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "planar":   # (the other two files date from rounds 1 - 2 and are left as they are)
        make_planar_kat()
    else:
        make_createmap_ref()
        make_oracle_kat()
        make_planar_kat()
