"""CPU tests of the libdewobble-style lens surface and NV12 output as the oracle DEFINES them
(render.ts:611-617,669-683,711-717,275-281).  libdewobble is not in the
reference tree, so these check the definition against independent fp64 models and against the
createMap.cl restatement it must coincide with."""
import math

import numpy as np

import oracle


def test_sincos_accuracy_and_endpoints():
    t = np.linspace(0.0, math.pi, 1_000_001).astype(np.float32)
    s, c = oracle.sincosf(t)
    t64 = t.astype(np.float64)
    assert np.abs(s - np.sin(t64)).max() < 1.5e-7 and np.abs(c - np.cos(t64)).max() < 1.5e-7
    s0, c0 = oracle.sincosf(np.float32([0.0]))
    assert s0[0] == 0.0 and c0[0] == 1.0


def test_lens_camera_focal_lengths():
    K = oracle.lens_camera(oracle.PROJ_FISH, 150.0, 3840, 2160)
    assert math.isclose(K[0, 0], 0.5 * math.hypot(3840, 2160) / math.radians(75.0), rel_tol=1e-14)
    assert (K[0, 2], K[1, 2]) == (1920.0, 1080.0)                          # render.ts:682-683
    K = oracle.lens_camera(oracle.PROJ_RECT, 90.0, 1920, 1080, cx=900.0, cy=500.5)
    assert math.isclose(K[1, 1], 0.5 * math.hypot(1920, 1080), rel_tol=1e-14)  # tan(45 deg) = 1
    assert (K[0, 2], K[1, 2]) == (900.0, 500.5)


def model_map(Kin, Kout, R, dw, dh, in_proj, out_proj):
    """fp64 model: output pixel -> ray (output projection) -> rotate -> input projection."""
    x, y = np.meshgrid(np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64))
    a, b = (x - Kout[0, 2]) / Kout[0, 0], (y - Kout[1, 2]) / Kout[1, 1]
    if out_proj == oracle.PROJ_FISH:
        th = np.hypot(a, b)
        s = np.where(th > 0, np.sin(th) / np.where(th > 0, th, 1), 1.0)
        ray = np.stack([a * s, b * s, np.cos(th)], -1)
        valid = th < math.pi
    else:
        ray = np.stack([a, b, np.ones_like(a)], -1)
        valid = np.ones_like(a, bool)
    w = ray @ np.asarray(R).T
    valid &= w[..., 2] > 0
    with np.errstate(all="ignore"):
        px, py = w[..., 0] / w[..., 2], w[..., 1] / w[..., 2]
        if in_proj == oracle.PROJ_FISH:
            r = np.hypot(px, py)
            k = np.where(r > 0, np.arctan(r) / np.where(r > 0, r, 1), 1.0)
            px, py = px * k, py * k
    return Kin[0, 2] + px * Kin[0, 0], Kin[1, 2] + py * Kin[1, 1], valid


def test_generalised_map_follows_the_projection_model():
    w, h, dw, dh = 640, 360, 480, 270
    for in_proj, in_fov in [(oracle.PROJ_FISH, 150.0), (oracle.PROJ_RECT, 100.0)]:
        for out_proj, out_fov in [(oracle.PROJ_RECT, 110.0), (oracle.PROJ_FISH, 170.0), (oracle.PROJ_FISH, 300.0)]:
            Kin = oracle.lens_camera(in_proj, in_fov, w, h)
            Kout = oracle.lens_camera(out_proj, out_fov, dw, dh)
            for rv in [(0, 0, 0), (0.05, -0.02, 0.1), (0.0, 1.2, 0.0)]:
                R = oracle.rodrigues(rv)
                p = oracle.map_params(Kin, Kout, R)
                mode = oracle.map_mode(in_proj, out_proj)
                mx, my = oracle.create_map_ex(p, dw, dh, mode)
                ex, ey, valid = model_map(Kin, Kout, R, dw, dh, in_proj, out_proj)
                # pixels whose ray is within a hair of the validity boundary may fall either way in fp32
                sure = valid & np.isfinite(ex) & (np.abs(ex) < 1e5) & (np.abs(ey) < 1e5)
                got = np.isfinite(mx)
                assert (got & ~valid).sum() <= 4 and (sure & ~got).sum() <= 4, (in_proj, out_proj, rv)
                both = sure & got
                tol = 2e-3 + 1e-9 * (ex * ex + ey * ey)  # the error grows like 1/w.z^2 towards the horizon
                assert both.sum() > 0.05 * dw * dh or rv[1] > 1
                assert (np.abs(mx - ex)[both] < tol[both]).all() and (np.abs(my - ey)[both] < tol[both]).all(), (in_proj, out_proj, rv)


def test_fish_to_rect_is_createmap_cl_except_where_that_degenerates():
    w, h = 640, 360
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    Ko = Ko.copy()
    Ko[0, 2], Ko[1, 2] = round(Ko[0, 2]), round(Ko[1, 2])     # integer centre: one pixel sits on the optical axis
    for rv in [(0, 0, 0), (0.02, -0.03, 0.01), (0.3, 0.2, -0.1)]:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        ax, ay = oracle.create_map(p, cw, ch)
        bx, by = oracle.create_map_ex(p, cw, ch, oracle.MAP_FISH_TO_RECT)
        same = (ax.view(np.uint32) == bx.view(np.uint32)) & (ay.view(np.uint32) == by.view(np.uint32))
        assert (~same).sum() <= 1
        if rv == (0, 0, 0):
            cx, cy = int(Ko[0, 2]), int(Ko[1, 2])
            assert np.isnan(ax[cy, cx]) and bx[cy, cx] == np.float32(K[0, 2]) and by[cy, cx] == np.float32(K[1, 2])
            assert (~same).sum() == 1
    # rays behind the camera: createMap.cl mirrors them, the generalised map calls them outside
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.0, 2.6, 0.0)))
    ax, _ = oracle.create_map(p, cw, ch)
    bx, _ = oracle.create_map_ex(p, cw, ch, oracle.MAP_FISH_TO_RECT)
    assert np.isfinite(ax).all() and np.isnan(bx).any()


def test_bgr_to_nv12_known_answers_and_layout():
    # BT.601 limited range: white, black, red, green, blue
    px = np.array([[[255, 255, 255], [0, 0, 0]], [[0, 0, 255], [0, 255, 0]]], np.uint8)
    y, uv = oracle.cvt_bgr_nv12(px)
    assert y.tolist() == [[235, 16], [82, 145]] and uv.tolist() == [[[128, 128]]]   # chroma = top-left pixel only
    blue = np.zeros((2, 2, 3), np.uint8)
    blue[..., 0] = 255
    y, uv = oracle.cvt_bgr_nv12(blue)
    assert y[0, 0] == 41 and uv[0, 0].tolist() == [240, 110]
    red = np.zeros((3, 5, 3), np.uint8)                                              # odd size: ceil() chroma planes
    red[..., 2] = 255
    y, uv = oracle.cvt_bgr_nv12(red)
    assert y.shape == (3, 5) and uv.shape == (2, 3, 2) and (uv == [90, 240]).all()


def test_nv12_round_trip_is_close():
    """NV12 -> BGR (reference cvtColor) -> NV12 (this conversion) returns the luma within 1 level for in-gamut colours."""
    rng = np.random.default_rng(5)
    h, w = 36, 64
    nv = np.empty((h * 3 // 2, w), np.uint8)
    nv[:h] = rng.integers(60, 200, (h, w))
    nv[h:] = rng.integers(118, 138, (h // 2, w))
    y, uv = oracle.cvt_bgr_nv12(oracle.cvt_nv12_bgr(nv))
    assert np.abs(y.astype(int) - nv[:h]).max() <= 1
    assert np.abs(uv.reshape(h // 2, w)[:, 0::2].astype(int) - nv[h:, 0::2]).max() <= 1


def test_warp_chain_nv12_output_is_conversion_of_bgr_output():
    import synth
    w, h = 128, 72
    f = synth.nv12(3, w, h)
    Kin = oracle.lens_camera(oracle.PROJ_FISH, 140.0, w, h)
    Kout = oracle.lens_camera(oracle.PROJ_RECT, 100.0, 101, 57)
    p = oracle.map_params(Kin, Kout, oracle.rodrigues((0.02, 0.01, -0.03)))
    bgr = oracle.warp_nv12_ex(f, p, 101, 57, oracle.MAP_FISH_TO_RECT, 0)
    y, uv = oracle.warp_nv12_ex(f, p, 101, 57, oracle.MAP_FISH_TO_RECT, 1)
    ey, euv = oracle.cvt_bgr_nv12(bgr)
    assert np.array_equal(y, ey) and np.array_equal(uv, euv)


def test_rolling_shutter_map_definition():
    """vo_create_map_rs (config 5, defined by this project): row 0 uses the first rotation exactly, the matrix of row y is
    the fp32 interpolation t = y / (rows - 1), and for a small rotation difference the map lies between the two per-frame
    maps (it is their interpolation to first order)."""
    w, h = 640, 360
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.01, 0.02, -0.01)))
    pb = oracle.map_params(K, Ko, oracle.rodrigues((0.02, 0.01, 0.01)))
    mx, my = oracle.create_map_rs(p, pb[8:], cw, ch)
    tx, ty = oracle.create_map(p, cw, ch)
    bx, by = oracle.create_map(pb, cw, ch)
    assert np.array_equal(mx[0], tx[0]) and np.array_equal(my[0], ty[0])
    y = ch // 3
    q = p.copy()
    t = np.float32(y) / np.float32(ch - 1)
    q[8:] = (np.float64(t) * (pb[8:] - p[8:]).astype(np.float64) + p[8:].astype(np.float64)).astype(np.float32)   # fmaf
    rx, ry = oracle.create_map(q, cw, ch)
    assert np.array_equal(mx[y], rx[y]) and np.array_equal(my[y], ry[y])
    lo, hi = np.minimum(tx, bx) - 0.05, np.maximum(tx, bx) + 0.05
    ok = ~np.isnan(mx)
    assert ((mx >= lo) & (mx <= hi))[ok].all()
    # identical rotations: the per-frame map, bit for bit
    sx, sy = oracle.create_map_rs(p, p[8:], cw, ch)
    assert np.array_equal(sx, tx, equal_nan=True) and np.array_equal(sy, ty, equal_nan=True)
