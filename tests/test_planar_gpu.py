"""GPU parity of the plane-wise warp (SURVEY.md 8(f) row 2 as written: NV12 -> NV12 / P010 -> P010, no colour round trip) through the
C ABI: vstab_warp_nv12_ex / _rs with VSTAB_OUT_NV12_PLANAR and vstab_warp_p010_planar.  Bar: every byte / word equals the checker --
the CPU oracle's chain for the IEEE maps, and for the default arithmetic (VSTAB_MAP_CREATEMAP_CL_OPENCL) the REFERENCE's own createMap
kernel run on this GPU followed by the oracle's plane-wise remap."""
import numpy as np
import pytest

import expect
import oracle
import synth
from test_p010_cpu import p010_frame

pytestmark = pytest.mark.gpu

ROTS = [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3)]


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def cams(w, h, rvec, preset=4):
    K = oracle.get_preset_camera(preset, w, h)
    Ko, (dw, dh) = oracle.get_output_camera(K, w, h)
    return oracle.map_params(K, Ko, oracle.rodrigues(rvec)), dw, dh, K, Ko


def planes(vs, cuda, dw, dh, pad=0, dtype=None, fill=7):
    import torch
    dtype = dtype or torch.uint8
    cw = (dw + 1) // 2
    yb = torch.full((dh, dw + pad), fill, dtype=dtype, device=cuda)
    cb = torch.full(((dh + 1) // 2, 2 * cw + pad), fill, dtype=dtype, device=cuda)
    return yb, cb, yb[:, :dw], cb[:, :2 * cw]


def run8(vs, cuda, f, p, dw, dh, mode, rot_bottom=None, pad=0):
    yb, cb, yv, cv = planes(vs, cuda, dw, dh, pad)
    fd = dev(f, cuda)
    if rot_bottom is None:
        vs.warp_nv12(fd, p, dw, dh, mode, vs.OUT_NV12_PLANAR, out=(yv, cv))
    else:
        vs.warp_nv12_rs(fd, p, rot_bottom, dw, dh, mode, vs.OUT_NV12_PLANAR, out=(yv, cv))
    if pad:
        assert bool((yb[:, dw:] == 7).all()) and bool((cb[:, cv.shape[1]:] == 7).all())      # nothing written beyond a row
    return yv.cpu().numpy(), cv.cpu().numpy()


def test_planar_bit_exact_vs_oracle_sizes_and_rotations(vs, cuda):
    """IEEE map (mode 0): 128 x 72 ... 1080p, preset cameras, odd outputs, rotations that look past the source (black border)."""
    for (w, h) in [(128, 72), (320, 180), (640, 368), (1920, 1080)]:
        f = synth.nv12(w + h, w, h, full_range=(w < 1000))
        for rv in (ROTS if w < 1000 else ROTS[1:2]):
            p, dw, dh, _, _ = cams(w, h, rv)
            for (ow, oh) in ([(dw, dh), (dw - 1, dh - 3)] if w < 1000 else [(dw, dh)]):
                gy, guv = run8(vs, cuda, f, p, ow, oh, vs.MAP_CREATEMAP_CL, pad=(16 if w == 320 else 0))
                ey, euv = oracle.warp_nv12_planar(f, p, ow, oh, 0)
                assert np.array_equal(gy, ey), (w, rv, ow, oh, int((gy != ey).sum()))
                assert np.array_equal(guv, euv), (w, rv, ow, oh, int((guv != euv).sum()))


def test_planar_every_projection_pair_and_per_row_rotation(vs, cuda):
    w, h, dw, dh = 640, 360, 481, 271
    f = synth.nv12(9, w, h, full_range=True)
    lenses = [(oracle.PROJ_FISH, 150.0, oracle.PROJ_RECT, 110.0), (oracle.PROJ_FISH, 150.0, oracle.PROJ_FISH, 165.0),
              (oracle.PROJ_RECT, 100.0, oracle.PROJ_RECT, 80.0), (oracle.PROJ_RECT, 100.0, oracle.PROJ_FISH, 300.0)]
    for ip, ifov, op, ofov in lenses:
        Kin, Kout = oracle.lens_camera(ip, ifov, w, h), oracle.lens_camera(op, ofov, dw, dh)
        mode = oracle.map_mode(ip, op)
        for rv in ROTS + [(0.0, 1.2, 0.0)]:
            p = oracle.map_params(Kin, Kout, oracle.rodrigues(rv))
            gy, guv = run8(vs, cuda, f, p, dw, dh, mode)
            ey, euv = oracle.warp_nv12_planar(f, p, dw, dh, mode)
            assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (mode, rv)
    # a rotation per output row (modes 0 and 1)
    p, dw, dh, K, Ko = cams(w, h, (0.03, -0.02, 0.05))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.05, -0.01, 0.02)))[8:]
    for mode in (0, 1):
        gy, guv = run8(vs, cuda, f, p, dw, dh, mode, rot_bottom=rb)
        ey, euv = oracle.warp_nv12_planar(f, p, dw, dh, mode, rb)
        assert np.array_equal(gy, ey) and np.array_equal(guv, euv), mode


def test_planar_default_arithmetic_vs_the_reference_kernel(vs, cuda):
    """VSTAB_MAP_CREATEMAP_CL_OPENCL (the handle's default): the checker's map IS the reference's createMap kernel on this GPU."""
    if not oracle.ref_gfx950_available():
        pytest.fail("oracle/_ref/createMap.gfx950.co is missing: __graft_entry__.build() compiles it where /root/reference exists")
    for (w, h) in [(320, 180), (1280, 720)]:
        f = synth.nv12(31 + w, w, h)
        for rv in ROTS[:2] if w > 1000 else ROTS:
            p, dw, dh, K, Ko = cams(w, h, rv)
            gy, guv = run8(vs, cuda, f, p, dw, dh, vs.MAP_CREATEMAP_CL_OPENCL)
            ey, euv = expect.warp_planar(f, p, dw, dh, expect.OPENCL)
            assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (w, rv, int((gy != ey).sum()), int((guv != euv).sum()))
        rb = oracle.map_params(K, Ko, oracle.rodrigues((0.02, 0.01, -0.01)))[8:]
        if w < 1000:  # per row: the checker launches the reference kernel once per output row
            gy, guv = run8(vs, cuda, f, p, dw, dh, vs.MAP_CREATEMAP_CL_OPENCL, rot_bottom=rb)
            ey, euv = expect.warp_planar(f, p, dw, dh, expect.OPENCL, rb)
            assert np.array_equal(gy, ey) and np.array_equal(guv, euv)


def test_planar_4k_config_3_shape(vs, cuda):
    """BASELINE config 3's warp (4K NV12 -> 3524 x 1999) plane-wise, default arithmetic, against the reference kernel's map."""
    w, h = 3840, 2160
    f = synth.nv12(77, w, h)
    p, dw, dh, _, _ = cams(w, h, (0.01, -0.02, 0.015))
    assert (dw, dh) == (3524, 1999)
    gy, guv = run8(vs, cuda, f, p, dw, dh, vs.MAP_CREATEMAP_CL_OPENCL)
    ey, euv = expect.warp_planar(f, p, dw, dh, expect.OPENCL)
    assert np.array_equal(gy, ey), int((gy != ey).sum())
    assert np.array_equal(guv, euv), int((guv != euv).sum())
    # size-independent properties at full size, no oracle involved: identical pinhole cameras -> the planes come back as they are;
    # principal point moved by (38, -6) -> both planes shifted, limited-range black where the source ends
    K = np.array([[2048.0, 0, 1920], [0, 2048.0, 1080], [0, 0, 1]])
    gy, guv = run8(vs, cuda, f, oracle.map_params(K, K, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT)
    assert np.array_equal(gy, f[:h]) and np.array_equal(guv, f[h:])
    Ko = K.copy()
    Ko[0, 2] -= 38
    Ko[1, 2] += 6
    gy, guv = run8(vs, cuda, f, oracle.map_params(K, Ko, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT)
    ey = np.full((h, w), 16, np.uint8)
    ey[6:, : w - 38] = f[: h - 6, 38:]
    euv = np.full((h // 2, w), 128, np.uint8)
    euv[3:, : w - 38] = f[h: h + h // 2 - 3, 38:]
    assert np.array_equal(gy, ey) and np.array_equal(guv, euv)


def test_planar_unaligned_and_pitched_planes_take_the_gather_path(vs, cuda):
    """Planes the 16-byte staging loads cannot take (odd base offsets, pitches not multiples of 16, a width that is not a multiple
    of 16) are sampled straight from global memory: same bytes."""
    import torch
    for (w, h, off, pitch) in [(328, 180, 0, 328), (320, 180, 8, 344), (320, 180, 2, 322)]:
        f = synth.nv12(5 + off, w, h, full_range=True)
        p, dw, dh, _, _ = cams(w, h, (0.02, -0.03, 0.01))
        buf = torch.zeros((h * 3 // 2) * pitch + 64, dtype=torch.uint8, device=cuda)
        view = buf[off:off + (h * 3 // 2) * pitch].view(h * 3 // 2, pitch)[:, :w]
        view.copy_(dev(f, cuda))
        yb, cb, yv, cv = planes(vs, cuda, dw, dh, pad=5)
        vs.warp_nv12(view, p, dw, dh, vs.MAP_CREATEMAP_CL, vs.OUT_NV12_PLANAR, out=(yv, cv))
        ey, euv = oracle.warp_nv12_planar(f, p, dw, dh, 0)
        assert np.array_equal(yv.cpu().numpy(), ey) and np.array_equal(cv.cpu().numpy(), euv), (w, off, pitch)


def test_planar_extreme_boxes(vs, cuda):
    """Strong magnification / minification and a roll of 90 degrees: boxes over the LDS budget (tall tiles split in two, then the
    gather path), tiles wholly outside the source."""
    w, h = 1280, 720
    f = synth.nv12(13, w, h)
    K = oracle.get_preset_camera(4, w, h)
    for scale, rv, (dw, dh) in [(0.25, (0.0, 0.0, 0.0), (448, 252)), (1.0, (0.0, 0.0, 1.5708), (1100, 700)), (3.0, (0.01, 0.0, 0.0), (1500, 900)),
                                (1.0, (0.9, 0.0, 0.0), (800, 450))]:
        Ko, _ = oracle.get_output_camera(K, w, h, scale=scale)
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        gy, guv = run8(vs, cuda, f, p, dw, dh, vs.MAP_CREATEMAP_CL)
        ey, euv = oracle.warp_nv12_planar(f, p, dw, dh, 0)
        assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (scale, rv)


def run10(vs, cuda, y, uv, p, dw, dh, mode, blend, rot_bottom=None):
    import torch
    yb, cb, yv, cv = planes(vs, cuda, dw, dh, pad=8, dtype=torch.int16)
    vs.warp_p010_planar(dev(y.view(np.int16), cuda), dev(uv.view(np.int16), cuda), p, dw, dh, rot_bottom, mode, blend, out_y=yv, out_uv=cv)
    assert bool((yb[:, dw:] == 7).all())
    return yv.cpu().numpy().view(np.uint16), cv.cpu().numpy().view(np.uint16)


def test_planar_p010_both_blends_per_row_and_default_arithmetic(vs, cuda):
    for (w, h) in [(320, 180), (1280, 720)]:
        y, uv, _, _ = p010_frame(3 + w, w, h)
        p, dw, dh, K, Ko = cams(w, h, (0.02, -0.03, 0.01))
        rb = oracle.map_params(K, Ko, oracle.rodrigues((0.04, -0.02, 0.0)))[8:]
        for blend in (0, 1):
            for rot_bottom in (None, rb):
                gy, guv = run10(vs, cuda, y, uv, p, dw, dh, vs.MAP_CREATEMAP_CL, blend, rot_bottom)
                ey, euv = oracle.warp_p010_planar(y, uv, p, dw, dh, 0, rot_bottom, blend)
                assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (w, blend, rot_bottom is not None, int((gy != ey).sum()), int((guv != euv).sum()))
            if w < 1000:
                gy, guv = run10(vs, cuda, y, uv, p, dw, dh, vs.MAP_CREATEMAP_CL_OPENCL, blend)
                ey, euv = expect.warp_p010_planar(y, uv, p, dw, dh, None, blend, expect.OPENCL)
                assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (w, blend)
    # odd output size, another projection pair
    y, uv, _, _ = p010_frame(11, 640, 360)
    Kin, Kout = oracle.lens_camera(oracle.PROJ_FISH, 150.0, 640, 360), oracle.lens_camera(oracle.PROJ_FISH, 165.0, 481, 271)
    p = oracle.map_params(Kin, Kout, oracle.rodrigues((0.02, -0.03, 0.01)))
    gy, guv = run10(vs, cuda, y, uv, p, 481, 271, vs.MAP_FISH_TO_FISH, 0)
    ey, euv = oracle.warp_p010_planar(y, uv, p, 481, 271, 2, None, 0)
    assert np.array_equal(gy, ey) and np.array_equal(guv, euv)


def test_planar_p010_gather_path_extreme_boxes_and_unaligned_planes(vs, cuda):
    """The 10-bit kernel off its fast path: boxes over the LDS budget (split tiles, then samples straight from global memory), tiles wholly
    outside the source, planes the 16-byte LDS-DMA cannot take (base offset / pitch not multiples of 16 bytes); both blends."""
    import torch
    w, h = 1280, 720
    y, uv, _, _ = p010_frame(21, w, h)
    K = oracle.get_preset_camera(4, w, h)
    for scale, rv, (dw, dh), blend in [(0.25, (0.0, 0.0, 0.0), (448, 252), 1), (1.0, (0.0, 0.0, 1.5708), (1100, 700), 0), (3.0, (0.01, 0.0, 0.0), (1500, 900), 1),
                                       (1.0, (0.9, 0.0, 0.0), (800, 450), 0)]:
        Ko, _ = oracle.get_output_camera(K, w, h, scale=scale)
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        gy, guv = run10(vs, cuda, y, uv, p, dw, dh, vs.MAP_CREATEMAP_CL, blend)
        ey, euv = oracle.warp_p010_planar(y, uv, p, dw, dh, 0, None, blend)
        assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (scale, rv, blend)
    for (w, h, off, pitch) in [(328, 180, 0, 328), (320, 180, 4, 344), (320, 180, 2, 322)]:   # off / pitch in 16-bit samples (chroma pairs stay 4-byte aligned)
        y, uv, _, _ = p010_frame(31 + off, w, h)
        p, dw, dh, _, _ = cams(w, h, (0.02, -0.03, 0.01))
        rows = h * 3 // 2
        buf = torch.zeros(rows * pitch + 64, dtype=torch.int16, device=cuda)
        view = buf[off:off + rows * pitch].view(rows, pitch)[:, :w]
        view.copy_(dev(np.concatenate([y, uv]).view(np.int16), cuda))
        for blend in (0, 1):
            yb, cb, yv, cv = planes(vs, cuda, dw, dh, pad=6, dtype=torch.int16)
            vs.warp_p010_planar(view[:h], view[h:], p, dw, dh, None, vs.MAP_CREATEMAP_CL, blend, out_y=yv, out_uv=cv)
            ey, euv = oracle.warp_p010_planar(y, uv, p, dw, dh, 0, None, blend)
            assert np.array_equal(yv.cpu().numpy().view(np.uint16), ey) and np.array_equal(cv.cpu().numpy().view(np.uint16), euv), (w, off, pitch, blend)


def test_planar_p010_4k_config_5_shape(vs, cuda):
    """BASELINE config 5's operator at full size with P010 planes out: binary16 blend, a rotation per output row, the reference kernel's
    arithmetic row by row (tests/expect.py); and the exact blend in the CPU-reproducible arithmetic against the oracle."""
    w, h = 3840, 2160
    y, uv, _, _ = p010_frame(5, w, h)
    p, dw, dh, K, Ko = cams(w, h, (0.004, -0.002, 0.001))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.006, -0.001, 0.002)))[8:]
    gy, guv = run10(vs, cuda, y, uv, p, dw, dh, vs.MAP_CREATEMAP_CL_OPENCL, 1, rb)
    ey, euv = expect.warp_p010_planar(y, uv, p, dw, dh, rb, 1, expect.OPENCL)
    assert np.array_equal(gy, ey) and np.array_equal(guv, euv), (int((gy != ey).sum()), int((guv != euv).sum()))
    gy, guv = run10(vs, cuda, y, uv, p, dw, dh, vs.MAP_CREATEMAP_CL, 0)
    ey, euv = oracle.warp_p010_planar(y, uv, p, dw, dh, 0, None, 0)
    assert np.array_equal(gy, ey) and np.array_equal(guv, euv)


def test_planar_argument_errors(vs, cuda):
    import torch
    f = dev(synth.nv12(1, 64, 36), cuda)
    p, dw, dh, _, _ = cams(64, 36, (0, 0, 0))
    yv, cv = vs.nv12_out_planes(dw, dh, cuda)
    with pytest.raises(vs.VstabError):          # chroma rows shorter than 2 * ceil(width / 2) bytes
        vs.warp_nv12(f, p, dw, dh, 0, vs.OUT_NV12_PLANAR, out=(yv, torch.empty((cv.shape[0], cv.shape[1] - 2), dtype=torch.uint8, device=cuda)))
    q = vs.quantised_map(p, dw, dh)
    with pytest.raises(vs.VstabError) as e:      # the quantised map holds no chroma positions
        vs.warp_nv12_mapped(f, q, dw, dh, vs.OUT_NV12_PLANAR, out=(yv, cv))
    assert e.value.status == vs.ERR_UNSUPPORTED
    with pytest.raises(vs.VstabError):
        vs.warp_p010_planar(torch.zeros((36, 64), dtype=torch.int16, device=cuda), torch.zeros((18, 64), dtype=torch.int16, device=cuda), p, dw, dh, blend=2)


def test_pipeline_pull_planar_matches_the_checker_frame_by_frame(vs, cuda):
    """vstab_pull_frame_nv12_planar on the pipeline object (default map arithmetic): every emitted frame is the plane-wise warp of its
    input frame under the rotation the handle reports -- the checker: the reference's createMap kernel + the oracle's plane-wise remap.
    Mixed with BGR pulls on one handle (the same frame either way up to the output format); tracking off bypasses the quantised-map
    cache; a read-out rotation per frame takes the per-row kernel."""
    import torch
    W, H, n, r = 640, 360, 12, 3
    K = oracle.get_preset_camera(4, W, H)
    frames, _ = synth.shaky_clip(3, K, W, H, n, sigma=0.004)
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames]
    # every frame copied into the library's ring (a decoder that recycles its surfaces; a callback source that promises nothing): the same planes out
    for kw, src in (({"ring_hold": 0, "total": n}, dev_frames), ({"hold": 0}, iter(dev_frames))):
        stab = vs.Stabilizer(src, smooth_radius=r, seed=5, **kw)
        for i in range(n - 1):
            yuv = stab.pull_nv12(planar=True)
            p = oracle.map_params(K, Ko, stab.warp_rotation(i))
            ey, euv = expect.warp_planar(frames[i + 1], p, cw, ch)
            assert np.array_equal(yuv[0].cpu().numpy(), ey) and np.array_equal(yuv[1].cpu().numpy(), euv), (kw, i)
        stab.close()
    stab = vs.Stabilizer(dev_frames, total=n, smooth_radius=r, seed=5)
    for i in range(n - 1):
        p = None
        if i % 3 == 2:      # a BGR pull in between: same handle, same look-ahead
            o = stab.pull()
            assert o is not None
            p = oracle.map_params(K, Ko, stab.warp_rotation(i))
            assert np.array_equal(o.cpu().numpy(), expect.warp(frames[i + 1], p, cw, ch)), i
            continue
        yuv = stab.pull_nv12(planar=True)
        assert yuv is not None, i
        p = oracle.map_params(K, Ko, stab.warp_rotation(i))
        ey, euv = expect.warp_planar(frames[i + 1], p, cw, ch)
        assert np.array_equal(yuv[0].cpu().numpy(), ey) and np.array_equal(yuv[1].cpu().numpy(), euv), i
    assert stab.pull_nv12(planar=True) is None
    stab.close()
    # tracking off: constant parameters (the BGR pulls of such a handle take the cached map; the plane-wise warp evaluates its own)
    stab = vs.Stabilizer(dev_frames, total=6, smooth_radius=1, tracking=0)
    p = oracle.map_params(K, Ko, np.eye(3))
    for i in range(5):
        if i == 2:
            assert np.array_equal(stab.pull().cpu().numpy(), expect.warp(frames[i + 1], p, cw, ch))
            continue
        y, uv = stab.pull_nv12(planar=True)
        ey, euv = expect.warp_planar(frames[i + 1], p, cw, ch)
        assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(uv.cpu().numpy(), euv), i
    stab.close()
    # read-out rotations: the warp takes a rotation per output row
    ro = [oracle.rodrigues((0.002 * (k % 3), -0.003, 0.001 * k)) for k in range(6)]
    stab = vs.Stabilizer(dev_frames[:6], total=6, smooth_radius=1, tracking=0, readouts=ro)
    for i in range(5):
        y, uv = stab.pull_nv12(planar=True)
        W_rot = stab.warp_rotation(i)
        p = oracle.map_params(K, Ko, W_rot)
        rb = oracle.map_params(K, Ko, ro[i + 1] @ W_rot)[8:]
        ey, euv = expect.warp_planar(frames[i + 1], p, cw, ch, expect.OPENCL, rb)
        assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(uv.cpu().numpy(), euv), i
    stab.close()


def test_pipeline_pull_p010_planar(vs, cuda):
    """pixel_depth = 10 handles: vstab_pull_frame_p010_planar = the plane-wise 10-bit warp of the ORIGINAL 16-bit planes under the
    handle's rotation, both blends; the 8-bit pulls are refused."""
    import torch
    W, H, n, r = 640, 360, 8, 2
    K = oracle.get_preset_camera(4, W, H)
    frames8, _ = synth.shaky_clip(3, K, W, H, n, sigma=0.004)
    rng = np.random.default_rng(9)
    wide = [((f.astype(np.uint16) << 8) | (rng.integers(0, 4, f.shape, dtype=np.uint16) << 6) | rng.integers(0, 64, f.shape, dtype=np.uint16)) for f in frames8]
    dev_frames = [torch.from_numpy(x.view(np.int16)).to(cuda) for x in wide]
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    for blend, kw in ((0, {}), (1, {}), (1, {"ring_hold": 0})):   # ring_hold 0: the 16-bit planes are copied into library memory on ingest
        stab = vs.Stabilizer(dev_frames, total=n, bit_depth=10, smooth_radius=r, seed=5, pixel_depth=10, blend=blend, **kw)
        for i in range(n - 1):
            oy = torch.empty((ch, cw), dtype=torch.int16, device=cuda)
            ouv = torch.empty(((ch + 1) // 2, 2 * ((cw + 1) // 2)), dtype=torch.int16, device=cuda)
            assert stab.pull_p010_planar_into(oy, ouv), i
            p = oracle.map_params(K, Ko, stab.warp_rotation(i))
            ey, euv = expect.warp_p010_planar(wide[i + 1][:H], wide[i + 1][H:], p, cw, ch, None, blend)
            assert np.array_equal(oy.cpu().numpy().view(np.uint16), ey) and np.array_equal(ouv.cpu().numpy().view(np.uint16), euv), (blend, i)
        with pytest.raises(vs.VstabError):
            stab.pull_nv12(planar=True)
        stab.close()


def test_planar_against_the_committed_golden_vectors(vs, cuda):
    """The HIP kernels against tests/golden/planar_kat.npz directly (a fixture that does not depend on rebuilding the oracle): 128 x 72, four rotations,
    NV12 and P010 (exact and binary16 blend), CPU-reproducible map arithmetic."""
    import os
    import torch
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "planar_kat.npz"))
    seed, w, h = (int(v) for v in kat["seed"])
    frame = synth.nv12(seed, w, h)
    for i in range(4):
        p = kat[f"params_{i}"]
        dh, dw = kat[f"nv12_y_{i}"].shape
        gy, guv = run8(vs, cuda, frame, p, dw, dh, vs.MAP_CREATEMAP_CL)
        assert np.array_equal(gy, kat[f"nv12_y_{i}"]) and np.array_equal(guv, kat[f"nv12_uv_{i}"]), i
        for blend in (0, 1):
            gy, guv = run10(vs, cuda, kat["p010_y"], kat["p010_uv"], p, dw, dh, vs.MAP_CREATEMAP_CL, blend)
            assert np.array_equal(gy, kat[f"p010_y_{i}_{blend}"]) and np.array_equal(guv, kat[f"p010_uv_{i}_{blend}"]), (i, blend)
