"""SURVEY.md 8(f) row 2 as it is written: the plane-wise NV12 -> NV12 (P010 -> P010) warp, no colour round trip.

The C++ prototype stops at BGR (FrameSourceWarp.cpp:313) and libdewobble is not in the reference tree, so the operator is
DEFINED in oracle/vstab_oracle.c (vo_remap_plane, vo_chroma_maps, vo_warp_planar_mapped: "parity unpinned").  Here the C
definition is checked against an independent numpy restatement, against closed-form cases, and against the path it is
the plane-wise twin of (the BGR warp followed by BGR -> NV12)."""
import numpy as np

import oracle
import synth
from test_p010_cpu import p010_frame


def np_remap_plane(src, mx, my, border, depth=8, blend=0):
    """cv::remap(INTER_LINEAR, BORDER_CONSTANT, border) of a (h, w, cn) plane of sample VALUES, pixel by pixel."""
    sh, sw, cn = src.shape
    dh, dw = mx.shape
    out = np.zeros((dh, dw, cn), np.int64)
    pad = np.empty((sh + 2, sw + 2, cn), np.int64)
    pad[:] = np.asarray(border, np.int64)
    pad[1:-1, 1:-1] = src
    for y in range(dh):
        for x in range(dw):
            ax, ay = np.float32(mx[y, x]) * np.float32(32), np.float32(my[y, x]) * np.float32(32)
            ok = abs(ax) < 2 ** 31 and abs(ay) < 2 ** 31      # NaN and out-of-range: cvRound -> INT_MIN -> far outside
            sx, sy = (int(np.rint(ax)), int(np.rint(ay))) if ok else (-2 ** 31, -2 ** 31)
            X, Y, fx, fy = max(min(sx >> 5, 32767), -32768), max(min(sy >> 5, 32767), -32768), sx & 31, sy & 31
            if X >= sw or X + 1 < 0 or Y >= sh or Y + 1 < 0:
                out[y, x] = border
                continue
            w = [(32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy]
            taps = [pad[Y + 1, X + 1], pad[Y + 1, X + 2], pad[Y + 2, X + 1], pad[Y + 2, X + 2]]
            if blend == 0:
                out[y, x] = (sum(t * k for t, k in zip(taps, w)) + 512) >> 10
            else:
                for c in range(cn):
                    acc = np.float16(0)
                    for t, k in zip(taps, w):
                        acc = np.float16(float(t[c]) * (k / 1024.0) + float(acc))
                    out[y, x, c] = min(int(np.rint(np.float32(acc))), 1023)
    return out


def np_warp_planar(y, uv, mx, my, depth=8, blend=0):
    """The definition, restated: luma with the map, chroma with the map of the even pixels halved; limited-range black outside."""
    sh = 0 if depth == 8 else 6
    Y = (y.astype(np.int64) >> sh)[..., None]
    UV = (uv.astype(np.int64) >> sh).reshape(uv.shape[0], -1, 2)
    oy = np_remap_plane(Y, mx, my, [16 if depth == 8 else 64], depth, blend)[..., 0]
    cmx, cmy = (mx[0::2, 0::2] * np.float32(0.5)).astype(np.float32), (my[0::2, 0::2] * np.float32(0.5)).astype(np.float32)
    c = 128 if depth == 8 else 512
    ouv = np_remap_plane(UV, cmx, cmy, [c, c], depth, blend).reshape(cmx.shape[0], -1)
    dt = np.uint8 if depth == 8 else np.uint16
    return (oy << sh).astype(dt), (ouv << sh).astype(dt)


def _cams(w, h, rvec):
    K = oracle.get_preset_camera(4, w, h)
    Ko, (dw, dh) = oracle.get_output_camera(K, w, h)
    return oracle.map_params(K, Ko, oracle.rodrigues(rvec)), dw, dh, K, Ko


def test_planar_chain_matches_numpy_8bit_every_mode_and_per_row():
    w, h = 48, 28
    f = synth.nv12(3, w, h, full_range=True)
    p, dw, dh, K, Ko = _cams(w, h, (0.03, -0.02, 0.05))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.05, -0.01, 0.02)))[8:]
    for mode in range(5):
        for (ow, oh) in ((dw, dh), (41, 23)):      # preset size and an odd one (ceil chroma)
            mx, my = oracle.create_map_ex(p, ow, oh, mode)
            gy, guv = oracle.warp_nv12_planar(f, p, ow, oh, mode)
            ey, euv = np_warp_planar(f[:h], f[h:], mx, my)
            assert gy.shape == (oh, ow) and guv.shape == ((oh + 1) // 2, 2 * ((ow + 1) // 2))
            assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (mode, ow, oh)
    for mode in (0, 1):
        mx, my = oracle.create_map_rs(p, rb, dw, dh, mode)
        gy, guv = oracle.warp_nv12_planar(f, p, dw, dh, mode, rb)
        ey, euv = np_warp_planar(f[:h], f[h:], mx, my)
        assert np.array_equal(gy, ey) and np.array_equal(guv, euv), mode
    # chroma maps: the even pixels' entries, halved
    mx, my = oracle.create_map(p, 41, 23)
    cmx, cmy = oracle.chroma_maps(mx, my)
    assert cmx.shape == (12, 21) and np.array_equal(cmx, mx[0::2, 0::2] * np.float32(0.5)) and np.array_equal(cmy, my[0::2, 0::2] * np.float32(0.5))


def test_planar_chain_matches_numpy_10bit_both_blends():
    w, h = 48, 28
    y, uv, y10, uv10 = p010_frame(7, w, h)
    p, dw, dh, K, Ko = _cams(w, h, (0.02, 0.03, -0.04))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.04, 0.0, -0.02)))[8:]
    for rot_bottom, (mx, my) in ((None, oracle.create_map(p, dw, dh)), (rb, oracle.create_map_rs(p, rb, dw, dh))):
        for blend in (0, 1):
            gy, guv = oracle.warp_p010_planar(y, uv, p, dw, dh, 0, rot_bottom, blend)
            ey, euv = np_warp_planar(y, uv, mx, my, 10, blend)
            assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (rot_bottom is not None, blend)
            assert (gy & 63).max() == 0 and (guv & 63).max() == 0          # P010 words: low six bits clear
    # the junk below the ten significant bits of the input is ignored
    a = oracle.warp_p010_planar(y, uv, p, dw, dh)
    b = oracle.warp_p010_planar(y & 0xFFC0, uv & 0xFFC0, p, dw, dh)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # samples that are 4 x an 8-bit frame: the exact blend gives 4 x the 8-bit result within 2 levels (rounding once at 10 bits)
    f8 = synth.nv12(5, w, h)
    y8, uv8 = oracle.warp_nv12_planar(f8, p, dw, dh)
    yq, uvq = oracle.warp_p010_planar(f8[:h].astype(np.uint16) << 8, f8[h:].astype(np.uint16) << 8, p, dw, dh)
    assert np.abs((yq >> 6).astype(int) - 4 * y8.astype(int)).max() <= 2 and np.abs((uvq >> 6).astype(int) - 4 * uv8.astype(int)).max() <= 2


def test_planar_closed_forms():
    """Pinhole cameras with power-of-two focal lengths and integer principal points make the rect -> rect map exact in fp32."""
    w, h = 64, 40
    f = synth.nv12(11, w, h)
    K = np.array([[64.0, 0, 32], [0, 64.0, 20], [0, 0, 1]])
    I = np.eye(3)
    # identical cameras: the output IS the input, plane by plane
    gy, guv = oracle.warp_nv12_planar(f, oracle.map_params(K, K, I), w, h, 3)
    assert np.array_equal(gy, f[:h]) and np.array_equal(guv, f[h:])
    # principal point moved by an EVEN number of pixels: both planes shift, limited-range black where the source ends
    Ko = K.copy()
    Ko[0, 2] -= 6
    Ko[1, 2] += 4                                                     # out(x, y) = in(x + 6, y - 4)
    gy, guv = oracle.warp_nv12_planar(f, oracle.map_params(K, Ko, I), w, h, 3)
    ey = np.full((h, w), 16, np.uint8)
    ey[4:, : w - 6] = f[: h - 4, 6:]
    euv = np.full((h // 2, w), 128, np.uint8)
    euv[2:, : w - 6] = f[h: h + h // 2 - 2, 6:]
    assert np.array_equal(gy, ey) and np.array_equal(guv, euv)
    # moved by ONE luma pixel: luma shifts by a pixel, chroma by half a chroma sample = the rounded mean of two neighbours
    Ko = K.copy()
    Ko[0, 2] -= 1
    gy, guv = oracle.warp_nv12_planar(f, oracle.map_params(K, Ko, I), w, h, 3)
    assert np.array_equal(gy[:, : w - 1], f[:h, 1:]) and (gy[:, w - 1] == 16).all()
    uv = f[h:].reshape(h // 2, w // 2, 2).astype(int)
    mean = (uv[:, :-1] + uv[:, 1:] + 1) >> 1                         # (16 a + 16 b) * 32 + 512 >> 10 = (a + b + 1) >> 1
    assert np.array_equal(guv.reshape(h // 2, w // 2, 2)[:, :-1], mean)
    assert np.array_equal(guv.reshape(h // 2, w // 2, 2)[:, -1], (uv[:, -1] + 128 + 1) >> 1)   # last column blends with the border
    # a frame of one colour stays that colour wherever the footprint is inside the source, and is black outside it
    flat = np.empty((h * 3 // 2, w), np.uint8)
    flat[:h] = 90
    flat[h:, 0::2] = 200
    flat[h:, 1::2] = 60
    p, dw, dh, _, _ = _cams(w, h, (0.2, -0.3, 0.1))                   # a rotation large enough to look past the source
    gy, guv = oracle.warp_nv12_planar(flat, p, dw, dh)
    mx, my = oracle.create_map(p, dw, dh)
    inside = (mx >= 0) & (mx <= w - 1.04) & (my >= 0) & (my <= h - 1.04)
    far = (mx < -1.1) | (mx > w + 0.1) | (my < -1.1) | (my > h + 0.1)
    assert inside.any() and far.any()
    assert (gy[inside] == 90).all() and (gy[far] == 16).all()
    ci, cf = inside[0::2, 0::2], far[0::2, 0::2]
    g = guv.reshape(guv.shape[0], -1, 2)
    # chroma positions are the luma positions halved: inside / far carry over with half a sample of slack
    cmx, cmy = oracle.chroma_maps(mx, my)
    cin = (cmx >= 0) & (cmx <= w // 2 - 1.04) & (cmy >= 0) & (cmy <= h // 2 - 1.04)
    assert (g[cin & ci] == (200, 60)).all() and (g[cf & ((cmx < -1.1) | (cmx > w // 2 + 0.1) | (cmy < -1.1) | (cmy > h // 2 + 0.1))] == 128).all()


def test_planar_agrees_with_the_bgr_round_trip_on_smooth_content():
    """The plane-wise warp and the colour round trip (NV12 -> BGR -> blend -> NV12, vo_warp_nv12_ex out_format 1) are different
    operators (one blends YUV samples, the other blends converted pixels and re-quantises twice); on in-gamut content they must stay
    within a few levels of each other -- a guard against a wrong siting or a transposed plane, not a parity claim."""
    w, h = 160, 96
    f = synth.nv12(21, w, h).astype(int)
    f[:h] = 60 + (f[:h] * 120) // 255                 # in gamut: no BGR channel clips on the round trip (with the synthetic frame's
    f[h:] = 128 + (f[h:] - 128) // 4                  # saturated colours 40 % of the pixels clip and the two operators part by up to 19 levels)
    f = f.astype(np.uint8)
    p, dw, dh, _, _ = _cams(w, h, (0.01, -0.02, 0.015))
    gy, guv = oracle.warp_nv12_planar(f, p, dw, dh)
    ry, ruv = oracle.warp_nv12_ex(f, p, dw, dh, 0, 1)
    bgr = oracle.warp_nv12_ex(f, p, dw, dh, 0, 0)
    mx, my = oracle.create_map(p, dw, dh)
    inside = (mx >= 1) & (mx <= w - 2) & (my >= 1) & (my <= h - 2)
    assert not ((bgr == 0) | (bgr == 255))[inside].any()
    dy = np.abs(gy.astype(int) - ry.astype(int))[inside]
    assert dy.max() <= 2 and dy.mean() < 0.2
    duv = np.abs(guv.reshape(ruv.shape).astype(int) - ruv.astype(int))[inside[0::2, 0::2]]
    assert duv.max() <= 2 and duv.mean() < 0.3


def test_planar_oracle_matches_the_committed_golden_vectors():
    """tests/golden/planar_kat.npz (tests/golden/make_golden.py planar): the definition's outputs for a seeded 128 x 72 frame, four rotations, 8-bit and
    10-bit (both blends), pinned against accidental change of the oracle."""
    import os
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "planar_kat.npz"))
    seed, w, h = (int(v) for v in kat["seed"])
    frame = synth.nv12(seed, w, h)
    for i in range(4):
        p = kat[f"params_{i}"]
        ch, cw = kat[f"nv12_y_{i}"].shape
        y, uv = oracle.warp_nv12_planar(frame, p, cw, ch, 0)
        assert np.array_equal(y, kat[f"nv12_y_{i}"]) and np.array_equal(uv, kat[f"nv12_uv_{i}"]), i
        for blend in (0, 1):
            y, uv = oracle.warp_p010_planar(kat["p010_y"], kat["p010_uv"], p, cw, ch, 0, None, blend)
            assert np.array_equal(y, kat[f"p010_y_{i}_{blend}"]) and np.array_equal(uv, kat[f"p010_uv_{i}_{blend}"]), (i, blend)
