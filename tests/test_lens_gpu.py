"""GPU parity tests of the lens surface (projection pairs) and NV12 output through the C ABI
(SURVEY.md 8(f) rows 1-2).  Bar: bit-exact against the CPU oracle for every map float and byte."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu

ROTS = [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3), (0.0, 1.2, 0.0), (0.0, 2.6, 0.0)]
LENSES = [  # (in_proj, in_dfov, out_proj, out_dfov)
    (oracle.PROJ_FISH, 150.0, oracle.PROJ_RECT, 110.0),
    (oracle.PROJ_FISH, 150.0, oracle.PROJ_FISH, 165.0),     # the CLI's "buffer" re-projection, render.ts:711-717
    (oracle.PROJ_RECT, 100.0, oracle.PROJ_RECT, 80.0),
    (oracle.PROJ_RECT, 100.0, oracle.PROJ_FISH, 300.0),     # output rays beyond 90 degrees
]


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_lens_camera_matches_oracle(vs):
    for proj, fov, w, h, cx, cy in [(0, 90.0, 1920, 1080, None, None), (1, 150.0, 3840, 2160, None, None), (1, 200.0, 640, 480, 300.5, 10.0)]:
        K = vs.lens_camera(proj, fov, w, h, -1.0 if cx is None else cx, -1.0 if cy is None else cy)
        assert np.allclose(K, oracle.lens_camera(proj, fov, w, h, cx, cy), rtol=1e-15, atol=0)
    for bad in [(0, 180.0), (1, 360.0), (2, 90.0), (0, 0.0)]:
        with pytest.raises(vs.VstabError):
            vs.lens_camera(bad[0], bad[1], 640, 480)


def test_create_map_ex_bit_exact_vs_oracle(vs, cuda):
    for (w, h, dw, dh) in [(640, 360, 481, 271), (1920, 1080, 1280, 720)]:
        for ip, ifov, op, ofov in LENSES:
            Kin = oracle.lens_camera(ip, ifov, w, h)
            Kout = oracle.lens_camera(op, ofov, dw, dh)          # integer / half-integer centre: the axis pixel exists
            mode = oracle.map_mode(ip, op)
            for rv in ROTS:
                p = oracle.map_params(Kin, Kout, oracle.rodrigues(rv))
                mx, my = vs.create_map(p, dw, dh, mode=mode)
                ox, oy = oracle.create_map_ex(p, dw, dh, mode)
                assert np.array_equal(np.isnan(ox), np.isnan(mx.cpu().numpy())), (mode, rv)
                ok = ~np.isnan(ox)
                assert np.array_equal(bits(mx.cpu().numpy())[ok], bits(ox)[ok]), (w, mode, rv)
                assert np.array_equal(bits(my.cpu().numpy())[ok], bits(oy)[ok]), (w, mode, rv)


def test_create_map_ex_axis_pixel_and_rays_behind_camera(vs, cuda):
    w, h = 640, 360
    Kin = oracle.lens_camera(oracle.PROJ_FISH, 150.0, w, h)
    Kout = oracle.lens_camera(oracle.PROJ_RECT, 100.0, w, h)
    p = oracle.map_params(Kin, Kout, np.eye(3))
    mx, my = vs.create_map(p, w, h, mode=vs.MAP_FISH_TO_RECT)
    assert float(mx[h // 2, w // 2]) == w / 2 and float(my[h // 2, w // 2]) == h / 2    # not NaN, unlike createMap.cl
    p = oracle.map_params(Kin, Kout, oracle.rodrigues((0.0, 2.6, 0.0)))
    mx, _ = vs.create_map(p, w, h, mode=vs.MAP_FISH_TO_RECT)
    assert bool(mx.isnan().any())


def check_warp(vs, cuda, f, p, dw, dh, mode, fmt, pad=0):
    import torch
    fd = dev(f, cuda)
    if fmt == vs.OUT_BGR8:
        out = torch.full((dh, dw * 3 + pad), 7, dtype=torch.uint8, device=cuda)
        view = out[:, pad:].unflatten(1, (dw, 3)) if pad == 0 else out[:, pad:pad + dw * 3].unflatten(1, (dw, 3))
        got = vs.warp_nv12(fd, p, dw, dh, mode, fmt, out=view).cpu().numpy()
        exp = oracle.warp_nv12_ex(f, p, dw, dh, mode, 0)
        assert np.array_equal(got, exp), (mode, fmt, dw, dh, pad)
        if pad:
            assert bool((out[:, :pad] == 7).all())
        return exp
    cw = (dw + 1) // 2
    yb = torch.full((dh, dw + pad + 8), 7, dtype=torch.uint8, device=cuda)
    cb = torch.full(((dh + 1) // 2, 2 * cw + pad + 8), 7, dtype=torch.uint8, device=cuda)
    y, c = vs.warp_nv12(fd, p, dw, dh, mode, fmt, out=(yb[:, pad:pad + dw], cb[:, pad:pad + 2 * cw]))
    ey, ec = oracle.warp_nv12_ex(f, p, dw, dh, mode, 1)
    assert np.array_equal(y.cpu().numpy(), ey), (mode, fmt, dw, dh, pad)
    assert np.array_equal(c.cpu().numpy().reshape(ec.shape), ec), (mode, fmt, dw, dh, pad)
    assert bool((yb[:, pad + dw:] == 7).all()) and bool((cb[:, pad + 2 * cw:] == 7).all())     # nothing written past the rows
    if pad:
        assert bool((yb[:, :pad] == 7).all()) and bool((cb[:, :pad] == 7).all())
    return ey


def test_warp_ex_all_modes_and_formats_bit_exact(vs, cuda):
    w, h = 640, 360
    f = synth.nv12(21, w, h)
    for ip, ifov, op, ofov in LENSES:
        Kin = oracle.lens_camera(ip, ifov, w, h)
        mode = oracle.map_mode(ip, op)
        for (dw, dh) in [(640, 360), (333, 201)]:
            Kout = oracle.lens_camera(op, ofov, dw, dh)
            for rv in [ROTS[0], ROTS[2], ROTS[3]]:
                p = oracle.map_params(Kin, Kout, oracle.rodrigues(rv))
                for fmt in (vs.OUT_BGR8, vs.OUT_NV12):
                    exp = check_warp(vs, cuda, f, p, dw, dh, mode, fmt)
                    if rv == ROTS[0]:
                        assert (exp != 0).mean() > 0.1                     # the test is not comparing black frames


def test_warp_ex_unaligned_destinations(vs, cuda):
    w, h = 320, 180
    f = synth.nv12(4, w, h)
    Kin = oracle.lens_camera(oracle.PROJ_FISH, 140.0, w, h)
    for (dw, dh, pad) in [(67, 35, 1), (130, 75, 3), (64, 32, 2), (5, 3, 1)]:
        Kout = oracle.lens_camera(oracle.PROJ_RECT, 100.0, dw, dh)
        p = oracle.map_params(Kin, Kout, oracle.rodrigues((0.01, 0.02, -0.05)))
        for fmt in (vs.OUT_BGR8, vs.OUT_NV12):
            check_warp(vs, cuda, f, p, dw, dh, vs.MAP_FISH_TO_RECT, fmt, pad=pad)


def test_reference_mode_nv12_output_is_conversion_of_reference_bgr(vs, cuda):
    """createMap.cl mode with NV12 output = the bit-exact BGR frame pushed through the BGR->NV12 arithmetic."""
    w, h = 1920, 1080
    f = synth.nv12(8, w, h)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)                         # 1759 x 998: odd width
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.02, -0.03, 0.01)))
    fd = dev(f, cuda)
    bgr = vs.warp_nv12_bgr(fd, p, cw, ch).cpu().numpy()
    y, c = vs.warp_nv12(fd, p, cw, ch, vs.MAP_CREATEMAP_CL, vs.OUT_NV12)
    ey, ec = oracle.cvt_bgr_nv12(bgr)
    assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(c.cpu().numpy().reshape(ec.shape), ec)


def test_fish_to_rect_mode_equals_reference_mode_away_from_the_axis(vs, cuda):
    w, h = 640, 360
    f = synth.nv12(2, w, h)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.03, 0.01, -0.02)))
    fd = dev(f, cuda)
    a = vs.warp_nv12(fd, p, cw, ch, vs.MAP_CREATEMAP_CL)
    b = vs.warp_nv12(fd, p, cw, ch, vs.MAP_FISH_TO_RECT)
    assert bool((a == b).all())


def test_warp_ex_rejects_bad_arguments(vs, cuda):
    import torch
    f = dev(synth.nv12(1, 64, 36), cuda)
    p = np.zeros(17, np.float32)
    with pytest.raises(vs.VstabError):
        vs.warp_nv12(f, p, 32, 18, mode=9)
    with pytest.raises(vs.VstabError):
        vs.warp_nv12(f, p, 32, 18, out_format=5, out=(torch.empty((18, 32), dtype=torch.uint8, device=cuda),) * 2)
    y = torch.empty((18, 32), dtype=torch.uint8, device=cuda)
    narrow = torch.empty((9, 30), dtype=torch.uint8, device=cuda)
    with pytest.raises(vs.VstabError):
        vs.warp_nv12(f, p, 32, 18, out_format=vs.OUT_NV12, out=(y, narrow))
    with pytest.raises(vs.VstabError):
        vs.create_map(p, 32, 18, mode=7)


# ---------------------------------------------------------------------------------------------
# pipeline in lens mode
# ---------------------------------------------------------------------------------------------
W, H, N = 640, 360, 24


@pytest.fixture(scope="module")
def clip():
    K = oracle.lens_camera(oracle.PROJ_FISH, 150.0, W, H)
    frames, rots = synth.shaky_clip(7, K, W, H, N, sigma=0.004)
    return K, frames, rots


def run(vs, cuda, frames, nv12=False, **cfg):
    import torch
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames]
    stab = vs.Stabilizer(dev_frames, total=len(frames), **cfg)
    outs = []
    while True:
        o = stab.pull_nv12() if nv12 else stab.pull()
        if o is None:
            break
        outs.append(tuple(t.cpu().numpy() for t in o) if nv12 else o.cpu().numpy())
    return stab, outs


LENS_CFG = dict(lens_mode=1, in_projection=1, out_projection=0, in_dfov=150.0, out_dfov=110.0, out_width=480, out_height=270)


def test_pipeline_lens_mode_sg(vs, cuda, clip):
    K, frames, rots = clip
    r = 4
    stab, outs = run(vs, cuda, frames, smooth_radius=r, seed=3, **LENS_CFG)
    assert len(outs) == N - 1 and stab.out_size == (480, 270)
    Kout = oracle.lens_camera(oracle.PROJ_RECT, 110.0, 480, 270)
    assert np.allclose(stab.K_in, K, rtol=1e-15) and np.allclose(stab.K_out, Kout, rtol=1e-15)
    log = stab.frame_log()
    # rotation estimates track the ground truth through the lens description (fisheye input)
    errs = [oracle.rotation_angle(lg["R"] @ (rots[k] @ rots[k - 1].T).T) for k, lg in enumerate(log, start=1)]
    assert np.median(errs) < 2e-3 and all(lg["inliers"] >= 40 for lg in log), (np.median(errs), max(errs))
    # smoothing identical to the oracle SG filter, pixels identical to the oracle's generalised warp
    filt = oracle.RotationFilter(r)
    accs = [lg["R_accum"] for lg in log]
    exp_R = []
    fed = 0
    for i in range(N - 1):
        while fed < min(i + r + 1, N - 1):
            filt.add(accs[fed]); fed += 1
        if i + r + 1 > N - 1:
            filt.add(accs[-1])                                   # EOF padding, FrameSourceWarp.cpp:456-461
        corrected = filt.filter()
        exp_R.append(np.linalg.inv(corrected @ np.linalg.inv(accs[i])))
    for i in (0, 3, N - 2):
        assert np.allclose(stab.warp_rotation(i), exp_R[i], atol=1e-10), i
        p = oracle.map_params(K, Kout, stab.warp_rotation(i))
        assert np.array_equal(outs[i], oracle.warp_nv12_ex(frames[i + 1], p, 480, 270, oracle.MAP_FISH_TO_RECT, 0)), i


def test_pipeline_fixed_and_none_modes_and_nv12_pull(vs, cuda, clip):
    K, frames, rots = clip
    n = 12
    Kout = oracle.lens_camera(oracle.PROJ_FISH, 150.0, W, H)
    cfg = dict(lens_mode=1, in_projection=1, out_projection=1, in_dfov=150.0, seed=5, smooth_radius=2)   # out_* default to the input
    stab, outs = run(vs, cuda, frames[:n], nv12=True, smoother=vs.SMOOTHER_FIXED, **cfg)
    assert len(outs) == n - 1 and stab.out_size == (W, H)
    log = stab.frame_log()
    for i in (0, 5, n - 2):
        Rw = stab.warp_rotation(i)
        assert np.allclose(Rw, log[i]["R_accum"], atol=1e-12)    # stab=fixed: undo the whole measured rotation
        p = oracle.map_params(K, Kout, Rw)
        ey, ec = oracle.warp_nv12_ex(frames[i + 1], p, W, H, oracle.MAP_FISH_TO_FISH, 1)
        assert np.array_equal(outs[i][0], ey) and np.array_equal(outs[i][1].reshape(ec.shape), ec), i
    # the held view really is steady: the accumulated estimate follows the true accumulated rotation
    assert oracle.rotation_angle(log[-1]["R_accum"] @ (rots[n - 1] @ rots[0].T).T) < 0.01
    # stab=none (tracking off): identity warp = pure re-projection; fish -> fish with the same lens is the identity map
    stab, outs = run(vs, cuda, frames[:4], nv12=False, tracking=0, **cfg)
    assert np.allclose(stab.warp_rotation(0), np.eye(3))
    exp = oracle.cvt_nv12_bgr(frames[1])
    inner = (slice(2, H - 2), slice(2, W - 2))
    assert np.abs(outs[0][inner].astype(int) - exp[inner]).max() <= 1   # map = identity up to fp32 rounding of the 1/32-px phase


def test_pipeline_rect_input_lens(vs, cuda):
    """in_p = rect: the estimator normalises instead of un-distorting; checked by ground truth."""
    K = oracle.lens_camera(oracle.PROJ_RECT, 90.0, W, H)
    frames, rots = synth.shaky_clip(9, K, W, H, 10, sigma=0.003, projection="rect")
    stab, outs = run(vs, cuda, frames, smooth_radius=2, seed=2, lens_mode=1, in_projection=0, out_projection=0, in_dfov=90.0)
    log = stab.frame_log()
    errs = [oracle.rotation_angle(lg["R"] @ (rots[k] @ rots[k - 1].T).T) for k, lg in enumerate(log, start=1)]
    assert np.median(errs) < 2e-3, errs


def test_pipeline_rejects_bad_lens(vs, cuda, clip):
    import torch
    K, frames, _ = clip
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames[:3]]
    for bad in (dict(in_dfov=0.0), dict(in_dfov=150.0, out_projection=0, out_dfov=180.0), dict(in_dfov=150.0, in_projection=5)):
        cfg = dict(lens_mode=1, in_projection=1, out_projection=0)
        cfg.update(bad)
        with pytest.raises(vs.VstabError):
            vs.Stabilizer(dev_frames, total=3, **cfg)
    with pytest.raises(vs.VstabError):
        vs.Stabilizer(dev_frames, total=3, smoother=9)


def test_debug_overlay_marks_tracked_features(vs, cuda, clip):
    """The filter surface's `debug` option (render.ts:678): green 7x7 squares where the warp sends the features tracked
    into each emitted frame -- product vs oracle (inverse of the map on the oracle's own tracked points)."""
    K, frames, rots = clip
    r = 1
    cfg = dict(lens_mode=1, in_projection=1, out_projection=0, in_dfov=150.0, out_dfov=110.0, out_width=480, out_height=270,
               smooth_radius=r, seed=3)
    plain_stab, plain = run(vs, cuda, frames[:5], **cfg)
    dbg_stab, dbg = run(vs, cuda, frames[:5], debug=1, **cfg)
    Kout = oracle.lens_camera(oracle.PROJ_RECT, 110.0, 480, 270)
    h = H
    corners = oracle.good_features(np.ascontiguousarray(frames[0][:h]))
    nxt, st = oracle.pyr_lk(frames[0][:h], frames[1][:h], corners)
    tracked = nxt[st > 0]
    R = dbg_stab.warp_rotation(0)
    assert np.allclose(R, plain_stab.warp_rotation(0), atol=0)
    centres, ok = oracle.project_to_output(tracked, K, Kout, R, in_fish=True, out_fish=False)
    exp = oracle.draw_markers(plain[0].copy(), centres, 3, (0, 255, 0))
    assert np.array_equal(dbg[0], exp)
    assert (dbg[0] != plain[0]).any(axis=-1).sum() > 20 * 49            # the markers are really there
    # NV12 pull: markers in the luma plane only
    _, dn = run(vs, cuda, frames[:5], nv12=True, debug=1, **cfg)
    _, pn = run(vs, cuda, frames[:5], nv12=True, **cfg)
    ey = oracle.draw_markers(pn[0][0].copy(), centres, 3, 235)
    assert np.array_equal(dn[0][0], ey) and np.array_equal(dn[0][1], pn[0][1])


def test_quantised_map_warp_equals_direct_warp(vs, cuda):
    """vstab_quantised_map + vstab_warp_nv12_mapped (the map written once for a run of frames with equal parameters)
    give the bytes of vstab_warp_nv12_ex for every mode, both formats, both tile shapes and odd sizes."""
    for (w, h, dw, dh) in [(640, 360, 583, 331), (1920, 1080, 1759, 998), (3840, 2160, 3524, 1999), (320, 180, 67, 35)]:
        f = synth.nv12(w + dh, w, h)
        fd = dev(f, cuda)
        for ip, ifov, op, ofov in LENSES[:2] if w > 2000 else LENSES:
            Kin, Kout = oracle.lens_camera(ip, ifov, w, h), oracle.lens_camera(op, ofov, dw, dh)
            mode = oracle.map_mode(ip, op)
            for rv in [ROTS[0], ROTS[2]]:
                p = oracle.map_params(Kin, Kout, oracle.rodrigues(rv))
                q = vs.quantised_map(p, dw, dh, mode)
                a = vs.warp_nv12(fd, p, dw, dh, mode, vs.OUT_BGR8)
                b = vs.warp_nv12_mapped(fd, q, dw, dh, vs.OUT_BGR8)
                assert bool((a == b).all()), (w, mode, rv)
                ay, ac = vs.warp_nv12(fd, p, dw, dh, mode, vs.OUT_NV12)
                by, bc = vs.warp_nv12_mapped(fd, q, dw, dh, vs.OUT_NV12)
                assert bool((ay == by).all()) and bool((ac == bc).all()), (w, mode, rv)
    # the reference kernel's map too (NaN on the optical axis included)
    K = oracle.get_preset_camera(4, 640, 360)
    Ko, (cw, ch) = oracle.get_output_camera(K, 640, 360)
    Ko = Ko.copy(); Ko[0, 2], Ko[1, 2] = round(Ko[0, 2]), round(Ko[1, 2])
    p = oracle.map_params(K, Ko, np.eye(3))
    fd = dev(synth.nv12(5, 640, 360), cuda)
    a = vs.warp_nv12(fd, p, cw, ch, vs.MAP_CREATEMAP_CL)
    b = vs.warp_nv12_mapped(fd, vs.quantised_map(p, cw, ch, vs.MAP_CREATEMAP_CL), cw, ch)
    assert bool((a == b).all()) and int(a[int(Ko[1, 2]), int(Ko[0, 2])].sum()) == 0


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: rolling-shutter warp (a rotation per output row).  No reference counterpart; the arithmetic is
# defined in oracle/vstab_oracle.c (vo_create_map_rs) and reproduced bit for bit by k_warp_fused's RS instantiations.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,mode,fmt", [(640, 360, 0, 0), (640, 360, 1, 0), (640, 360, 0, 1), (1920, 1080, 0, 0), (322, 182, 1, 0)])
def test_rolling_shutter_warp_bit_exact_vs_oracle(vs, cuda, w, h, mode, fmt):
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    frame = synth.nv12(21, w, h)
    fd = dev(frame, cuda)
    for top, bot in [((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), ((0.01, -0.02, 0.005), (0.03, -0.01, -0.01)), ((0.2, 0.1, -0.3), (0.25, 0.12, -0.28))]:
        p = oracle.map_params(K, Ko, oracle.rodrigues(top))
        rb = oracle.map_params(K, Ko, oracle.rodrigues(bot))[8:]
        exp = oracle.warp_nv12_rs(frame, p, rb, cw, ch, mode, fmt)
        got = vs.warp_nv12_rs(fd, p, rb, cw, ch, mode, fmt)
        if fmt == 0:
            assert np.array_equal(got.cpu().numpy(), exp), (top, bot, int((got.cpu().numpy() != exp).sum()))
        else:
            assert np.array_equal(got[0].cpu().numpy(), exp[0]) and np.array_equal(got[1].cpu().numpy().reshape(exp[1].shape), exp[1])
    # equal rotations top and bottom = the per-frame warp
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.02, 0.01, -0.03)))
    same = vs.warp_nv12_rs(fd, p, p[8:], cw, ch, mode, 0)
    assert np.array_equal(same.cpu().numpy(), vs.warp_nv12(fd, p, cw, ch, mode, 0).cpu().numpy())


def test_rolling_shutter_warp_at_4k_and_bad_modes(vs, cuda):
    w, h = 3840, 2160
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    frame = synth.nv12(22, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.004, -0.002, 0.001)))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.006, -0.001, 0.002)))[8:]   # 0.2 degrees of shake during the readout
    got = vs.warp_nv12_rs(dev(frame, cuda), p, rb, cw, ch).cpu().numpy()
    assert np.array_equal(got, oracle.warp_nv12_rs(frame, p, rb, cw, ch))
    with pytest.raises(vs.VstabError):
        vs.warp_nv12_rs(dev(frame, cuda), p, rb, cw, ch, mode=2)   # fisheye output: no per-row variant


def test_fused_warp_extreme_box_shapes(vs, cuda):
    """Staging of a tile's source box walks it in units of 8 x 2 pixels with an integer divmod by the box width in units
    (the old magic-number division failed for wide, flat boxes such as 992 x 10).  Anisotropic pinhole cameras
    give exactly such boxes -- a 64 x 32 output tile reading ~970 x 10 source pixels, and the transpose, ~16 x 480 -- and
    the result must still equal the oracle in every byte."""
    for sw, sh, dw, dh, sx, sy in [(2048, 32, 128, 64, 15.0, 0.25), (64, 1024, 128, 64, 0.125, 15.0), (4096, 64, 200, 70, 15.5, 0.3)]:
        frame = synth.nv12(61, sw, sh)
        Ki = np.array([[100.0 * sx, 0, sw / 2], [0, 100.0 * sy, sh / 2], [0, 0, 1]])
        Ko = np.array([[100.0, 0, dw / 2], [0, 100.0, dh / 2], [0, 0, 1]])
        for rot in [(0.0, 0.0, 0.0), (0.0, 0.0, 0.002)]:
            p = oracle.map_params(Ki, Ko, oracle.rodrigues(rot))
            got = vs.warp_nv12(dev(frame, cuda), p, dw, dh, vs.MAP_RECT_TO_RECT, vs.OUT_BGR8).cpu().numpy()
            exp = oracle.warp_nv12_ex(frame, p, dw, dh, 3, 0)
            assert np.array_equal(got, exp), (sw, sh, rot, int((got != exp).sum()))
            assert exp.any()
