"""BASELINE.json config 5 -- the 10-bit pixel path (P010 in, 10-bit BGR out, exact or fp16 blend, optional rotation
per row).  The reference is 8-bit throughout, so the arithmetic is DEFINED in oracle/vstab_oracle.c (vo_warp_p010);
here the C definition is checked against an independent numpy restatement and against the 8-bit path it extends."""
import numpy as np

import oracle
import synth

CY, CUB, CUG, CVG, CVR = 1220542, 2116026, -409993, -852492, 1673527


def p010_frame(seed, w, h, junk=True):
    """10-bit planes from the 8-bit synthetic frame (x4 + 2 extra bits of detail), low 6 bits junk that must be ignored."""
    rng = np.random.default_rng(seed)
    f = synth.nv12(seed, w, h).astype(np.uint16)
    y10 = np.minimum(f[:h] * 4 + rng.integers(0, 4, (h, w), dtype=np.uint16), 1023)
    uv10 = np.minimum(f[h:] * 4 + rng.integers(0, 4, (h // 2, w), dtype=np.uint16), 1023)
    lo = rng.integers(0, 64, (h, w), dtype=np.uint16) if junk else 0
    lo2 = rng.integers(0, 64, (h // 2, w), dtype=np.uint16) if junk else 0
    return (y10 << 6) | lo, (uv10 << 6) | lo2, y10, uv10


def np_bgr10(y10, uv10):
    h, w = y10.shape
    Y = y10.astype(np.int64)
    U = np.repeat(np.repeat(uv10[:, 0::2], 2, 0), 2, 1)[:h, :w].astype(np.int64) - 512
    V = np.repeat(np.repeat(uv10[:, 1::2], 2, 0), 2, 1)[:h, :w].astype(np.int64) - 512
    yy = np.maximum(Y - 64, 0) * CY
    b = (yy + (1 << 19) + CUB * U) >> 20
    g = (yy + (1 << 19) + CVG * V + CUG * U) >> 20
    r = (yy + (1 << 19) + CVR * V) >> 20
    return np.clip(np.stack([b, g, r], -1), 0, 1023).astype(np.uint16)


def np_remap10(bgr, mx, my, blend):
    sh, sw, _ = bgr.shape
    dh, dw = mx.shape
    out = np.zeros((dh, dw, 3), np.uint16)
    pad = np.zeros((sh + 2, sw + 2, 3), np.int64)
    pad[1:-1, 1:-1] = bgr
    for y in range(dh):
        for x in range(dw):
            ax, ay = np.float32(mx[y, x]) * np.float32(32), np.float32(my[y, x]) * np.float32(32)
            if not (abs(ax) < 2 ** 31 and abs(ay) < 2 ** 31):
                continue
            sx, sy = int(np.rint(ax)), int(np.rint(ay))
            X, Y, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
            if X >= sw or X + 1 < 0 or Y >= sh or Y + 1 < 0:
                continue
            w = [(32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy]
            taps = [pad[Y + 1, X + 1], pad[Y + 1, X + 2], pad[Y + 2, X + 1], pad[Y + 2, X + 2]]
            if blend == 0:
                out[y, x] = (sum(t * k for t, k in zip(taps, w)) + 512) >> 10
            else:
                for c in range(3):
                    acc = np.float16(0)
                    for t, k in zip(taps, w):   # p * (w/1024) + acc is exact in float64; one rounding to binary16 = a fused multiply-add
                        acc = np.float16(float(t[c]) * (k / 1024.0) + float(acc))
                    out[y, x, c] = min(int(np.rint(np.float32(acc))), 1023)
    return out


def test_p010_conversion_matches_numpy_and_extends_the_8bit_path():
    w, h = 64, 36
    y, uv, y10, uv10 = p010_frame(3, w, h)
    got = oracle.cvt_p010_bgr10(y, uv)
    assert np.array_equal(got, np_bgr10(y10, uv10))
    # known answers: black, white, mid grey (neutral chroma), and the low 6 bits do not matter
    for Y, exp in ((64, 0), (940, 1020), (502, 510)):
        px = oracle.cvt_p010_bgr10(np.full((2, 2), Y << 6, np.uint16), np.full((1, 2), 512 << 6, np.uint16))
        assert (px == exp).all(), (Y, px[0, 0])
    assert np.array_equal(oracle.cvt_p010_bgr10(y & 0xFFC0, uv & 0xFFC0), got)
    # samples that are exactly 4x an 8-bit frame convert to ~4x the 8-bit BGR (same constants, two more bits)
    f8 = synth.nv12(5, w, h)
    b8 = oracle.cvt_nv12_bgr(f8).astype(int)
    b10 = oracle.cvt_p010_bgr10(f8[:h].astype(np.uint16) << 8, f8[h:].astype(np.uint16) << 8).astype(int)
    unclipped = (b8 > 0) & (b8 < 255)
    assert np.abs(b10 - 4 * b8)[unclipped].max() <= 4


def test_p010_warp_chain_matches_numpy_for_both_blends_and_per_row_rotation():
    w, h = 48, 28
    y, uv, y10, uv10 = p010_frame(7, w, h)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.03, -0.02, 0.05)))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.05, -0.01, 0.02)))[8:]
    bgr = np_bgr10(y10, uv10)
    for rot_bottom, (mx, my) in ((None, oracle.create_map(p, cw, ch)), (rb, oracle.create_map_rs(p, rb, cw, ch))):
        for blend in (0, 1):
            got = oracle.warp_p010(y, uv, p, cw, ch, rot_bottom, 0, blend)
            assert np.array_equal(got, np_remap10(bgr, mx, my, blend)), (rot_bottom is not None, blend)
    # the fp16 blend stays within two 10-bit levels of the exact one, and is not identical to it
    y, uv, _, _ = p010_frame(8, 320, 180)
    K = oracle.get_preset_camera(4, 320, 180)
    Ko, (cw, ch) = oracle.get_output_camera(K, 320, 180)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.01, 0.02, -0.03)))
    a = oracle.warp_p010(y, uv, p, cw, ch, blend=0).astype(int)
    b = oracle.warp_p010(y, uv, p, cw, ch, blend=1).astype(int)
    assert np.abs(a - b).max() <= 2 and (a != b).any()
    # other projection pairs go through the same remap
    for mode in (1, 2, 3, 4):
        mx, my = oracle.create_map_ex(p, 40, 24, mode)
        assert np.array_equal(oracle.warp_p010(y[:28, :48], uv[:14, :48], p, 40, 24, None, mode, 0),
                              np_remap10(np_bgr10(y[:28, :48] >> 6, uv[:14, :48] >> 6), mx, my, 0)), mode


def test_bgr10_to_p010_matches_numpy_and_extends_the_8bit_conversion():
    """vo_cvt_bgr10_p010 (the 10-bit path's encoder hand-off, DEFINED here): an independent numpy restatement in int64; the
    float BT.601 formula within one level; and for samples that are 4 x an 8-bit frame the result is 4 x the 8-bit
    BGR -> NV12 conversion within 4 levels.  Round trip P010 -> BGR -> P010 stays within 3 levels where nothing saturates."""
    rng = np.random.default_rng(5)
    for (w, h) in ((64, 36), (33, 17)):
        bgr = rng.integers(0, 1024, (h, w, 3), dtype=np.uint16)
        y, uv = oracle.cvt_bgr10_p010(bgr)
        B, G, R = (bgr[..., k].astype(np.int64) for k in range(3))
        ey = np.clip((269484 * R + 528482 * G + 102760 * B + (1 << 19) + (64 << 20)) >> 20, 0, 1023)
        eu = np.clip((-155188 * R - 305135 * G + 460324 * B + (1 << 19) + (512 << 20)) >> 20, 0, 1023)[::2, ::2]
        ev = np.clip((460324 * R - 385875 * G - 74448 * B + (1 << 19) + (512 << 20)) >> 20, 0, 1023)[::2, ::2]
        assert np.array_equal(y, (ey << 6).astype(np.uint16))
        assert np.array_equal(uv[:, 0::2], (eu << 6).astype(np.uint16)) and np.array_equal(uv[:, 1::2], (ev << 6).astype(np.uint16))
        assert (y & 63).max() == 0 and (uv & 63).max() == 0
        fy = 64 + (0.257 * R + 0.504 * G + 0.098 * B)
        assert np.abs((y >> 6).astype(np.float64) - fy).max() <= 1.0
    # 4 x an 8-bit frame
    b8 = rng.integers(0, 256, (36, 64, 3), dtype=np.uint8)
    y8, uv8 = oracle.cvt_bgr_nv12(b8)
    y10, uv10 = oracle.cvt_bgr10_p010(b8.astype(np.uint16) * 4)
    assert np.abs((y10 >> 6).astype(int) - 4 * y8.astype(int)).max() <= 4
    assert np.abs((uv10 >> 6).astype(int) - 4 * uv8.reshape(uv10.shape).astype(int)).max() <= 4
    # round trip through the 10-bit colour conversion on a smooth frame (chroma constant over 2 x 2 blocks)
    yy, uu, _, _ = p010_frame(11, 64, 36, junk=False)
    mid = oracle.cvt_p010_bgr10(yy, uu)
    back_y, back_uv = oracle.cvt_bgr10_p010(mid)
    inner = ((mid > 0) & (mid < 1023)).all(axis=2) & (yy >> 6 >= 64) & (yy >> 6 <= 940)   # in gamut: nothing clipped on the way
    assert inner.mean() > 0.2
    assert np.abs((back_y >> 6).astype(int) - (yy >> 6).astype(int))[inner].max() <= 3
