"""GPU parity tests of vstab_warp_p010 (BASELINE.json config 5: P010 in, 10-bit BGR out, exact or fp16 blend, optional
rotation per output row) against its definition in the oracle (vo_warp_p010).  Bar: every 16-bit sample equal."""
import numpy as np
import pytest

import expect
import oracle
from test_p010_cpu import p010_frame

pytestmark = pytest.mark.gpu


def dev16(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(cuda)


def host(t):
    return t.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("w,h", [(640, 360), (322, 182), (66, 34)])
def test_p010_warp_bit_exact_vs_oracle(vs, cuda, w, h):
    y, uv, _, _ = p010_frame(31, w, h)
    yd, ud = dev16(y, cuda), dev16(uv, cuda)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    for rot in [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3), (0.0, 2.6, 0.0)]:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rot))
        for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
            exp = oracle.warp_p010(y, uv, p, cw, ch, None, 0, blend)
            got = host(vs.warp_p010(yd, ud, p, cw, ch, None, 0, blend))
            assert np.array_equal(got, exp), (rot, blend, int((got != exp).sum()))
    # rotation per output row
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.01, -0.02, 0.005)))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.03, -0.01, -0.01)))[8:]
    for mode in (0, 1):
        for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
            assert np.array_equal(host(vs.warp_p010(yd, ud, p, cw, ch, rb, mode, blend)), oracle.warp_p010(y, uv, p, cw, ch, rb, mode, blend)), (mode, blend)
    assert np.array_equal(host(vs.warp_p010(yd, ud, p, cw, ch, p[8:], 0)), host(vs.warp_p010(yd, ud, p, cw, ch, None, 0)))  # equal rotations = one rotation


def test_p010_warp_lens_modes_pitched_planes_and_bad_arguments(vs, cuda):
    import torch
    w, h = 320, 180
    y, uv, _, _ = p010_frame(32, w, h)
    Kin = oracle.lens_camera(oracle.PROJ_FISH, 150.0, w, h)
    for mode, (ip, ifov, op, ofov) in ((1, (1, 150.0, 0, 110.0)), (2, (1, 150.0, 1, 165.0)), (3, (0, 100.0, 0, 80.0)), (4, (0, 100.0, 1, 300.0))):
        Ki, Ko = oracle.lens_camera(ip, ifov, w, h), oracle.lens_camera(op, ofov, 301, 171)
        p = oracle.map_params(Ki, Ko, oracle.rodrigues((0.05, -0.1, 0.2)))
        got = host(vs.warp_p010(dev16(y, cuda), dev16(uv, cuda), p, 301, 171, None, mode, vs.BLEND_EXACT))
        assert np.array_equal(got, oracle.warp_p010(y, uv, p, 301, 171, None, mode, 0)), mode
        # a rotation per output row with every projection pair (the 8-bit kernel has it for the fisheye-input pairs only)
        rb = oracle.map_params(Ki, Ko, oracle.rodrigues((0.07, -0.09, 0.18)))[8:]
        got = host(vs.warp_p010(dev16(y, cuda), dev16(uv, cuda), p, 301, 171, rb, mode, vs.BLEND_FP16))
        assert np.array_equal(got, oracle.warp_p010(y, uv, p, 301, 171, rb, mode, 1)), ("rs", mode)
    # planes that are views into wider buffers, destination with a padded pitch
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.01, 0.0, -0.02)))
    Y = torch.zeros((h + 3, w + 10), dtype=torch.int16, device=cuda)
    U = torch.zeros((h // 2 + 1, w + 6), dtype=torch.int16, device=cuda)
    Y[2:h + 2, 4:w + 4] = dev16(y, cuda)
    U[1:, 2:w + 2] = dev16(uv, cuda)
    out = torch.zeros((ch, cw + 5, 3), dtype=torch.int16, device=cuda)
    vs.warp_p010(Y[2:h + 2, 4:w + 4], U[1:, 2:w + 2], p, cw, ch, out=out[:, :cw])
    assert np.array_equal(host(out[:, :cw]), oracle.warp_p010(y, uv, p, cw, ch))
    assert int(out[:, cw:].abs().sum()) == 0
    for bad in (dict(mode=7), dict(blend=2)):
        with pytest.raises(vs.VstabError):
            vs.warp_p010(dev16(y, cuda), dev16(uv, cuda), p, cw, ch, **bad)
    with pytest.raises(vs.VstabError):
        vs.warp_p010(dev16(y, cuda)[:, 1:], dev16(uv, cuda)[:, 1:], p, cw, ch)     # odd width / chroma pairs not 4-byte aligned


def test_p010_warp_at_4k_config5(vs, cuda):
    """BASELINE config 5 at full size: 3840x2160 P010, rotation per row, fp16 blend and exact blend."""
    w, h = 3840, 2160
    y, uv, _, _ = p010_frame(33, w, h)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.004, -0.002, 0.001)))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.006, -0.001, 0.002)))[8:]
    yd, ud = dev16(y, cuda), dev16(uv, cuda)
    exact = host(vs.warp_p010(yd, ud, p, cw, ch, rb, 0, vs.BLEND_EXACT))
    half = host(vs.warp_p010(yd, ud, p, cw, ch, rb, 0, vs.BLEND_FP16))
    assert np.array_equal(exact, oracle.warp_p010(y, uv, p, cw, ch, rb, 0, 0))
    assert np.array_equal(half, oracle.warp_p010(y, uv, p, cw, ch, rb, 0, 1))
    d = np.abs(exact.astype(np.int32) - half.astype(np.int32))
    assert d.max() <= 2 and 0.0 < (d > 0).mean() < 0.5
    # the 8-bit product path on the same frame narrowed as vstab_pack_p010 does it (two bits of luma and chroma truncated:
    # up to 0.75 + 2.02 * 0.75 levels in blue before rounding): the 10-bit result, reduced to 8 bits, stays within 4 levels
    # and within 1 on average
    f8 = np.concatenate([(y >> 8).astype(np.uint8), (uv >> 8).astype(np.uint8)], 0)
    import torch
    b8 = vs.warp_nv12_rs(torch.from_numpy(f8).to(cuda), p, rb, cw, ch).cpu().numpy().astype(np.int32)
    d8 = np.abs((exact.astype(np.int32) >> 2) - b8)
    assert d8.max() <= 4 and d8.mean() < 1.0


def test_pipeline_object_with_10bit_pixels(vs, cuda):
    """vstab_config.pixel_depth = 10: the tracker sees the narrowed luma, so every decision and rotation equals the 8-bit
    pipeline's on the narrowed clip; each emitted frame is vstab_warp_p010 of the ORIGINAL 16-bit planes under that
    rotation (checked against the oracle's definition), for both blends."""
    import ctypes
    import torch
    import synth
    W, H, n, r = 640, 360, 12, 3
    K = oracle.get_preset_camera(4, W, H)
    frames8, _ = synth.shaky_clip(3, K, W, H, n, sigma=0.004)
    rng = np.random.default_rng(9)
    # P010 planes whose top 8 bits are the 8-bit clip (what vstab_pack_p010 narrows to), with two more bits of detail and junk below
    wide = [((f.astype(np.uint16) << 8) | (rng.integers(0, 4, f.shape, dtype=np.uint16) << 6) | rng.integers(0, 64, f.shape, dtype=np.uint16)) for f in frames8]
    dev = [torch.from_numpy(x.view(np.int16)).to(cuda) for x in wide]
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)

    surface = torch.empty_like(dev[0])

    def run(depth, blend, recycle=False):
        state = {"i": 0, "loaded": -1}

        def fill(out, advance):
            i = state["i"]
            if i >= n:
                return vs.EOF
            t = dev[i]
            if recycle:  # a decoder with ONE surface: every callback overwrites what the previous one handed out (vstab_frame.hold = 0)
                if state["loaded"] != i:
                    surface.copy_(dev[i])
                    torch.cuda.synchronize()
                    state["loaded"] = i
                t = surface
            o = out.contents
            o.y, o.uv = t.data_ptr(), t.data_ptr() + H * t.stride(0) * 2
            o.pitch_y = o.pitch_uv = t.stride(0) * 2
            o.width, o.height, o.mem, o.pts, o.hold, o.bit_depth = W, H, 0, i, 0, 10
            if advance:
                state["i"] += 1
            return 0
        pull = vs.PULL_FN(lambda u, o: fill(o, True))
        peek = vs.PULL_FN(lambda u, o: fill(o, False))
        src = vs.Source(pull, peek, None)
        cfg = vs.default_config(smooth_radius=r, seed=5, pixel_depth=depth, blend=blend)
        h = ctypes.c_void_p()
        assert vs.lib.vstab_create(ctypes.byref(cfg), ctypes.byref(src), ctypes.byref(h)) == vs.OK, vs.lib.vstab_last_error()
        outs, rots = [], []
        while True:
            if depth == 10:
                o = torch.empty((ch, cw, 3), dtype=torch.int16, device=cuda)
                st = vs.lib.vstab_pull_frame_bgr16(h, o.data_ptr(), o.stride(0) * 2)
            else:
                o = torch.empty((ch, cw, 3), dtype=torch.uint8, device=cuda)
                st = vs.lib.vstab_pull_frame(h, o.data_ptr(), o.stride(0))
            if st == vs.EOF:
                break
            assert st == vs.OK, vs.lib.vstab_last_error()
            outs.append(o.cpu().numpy())
            R = np.zeros(9)
            assert vs.lib.vstab_get_warp_rotation(h, len(outs) - 1, R.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == vs.OK
            rots.append(R.reshape(3, 3))
        # the wrong pull function for the handle is refused
        o8 = torch.empty((ch, cw, 3), dtype=torch.uint8, device=cuda)
        o16 = torch.empty((ch, cw, 3), dtype=torch.int16, device=cuda)
        if depth == 10:
            assert vs.lib.vstab_pull_frame(h, o8.data_ptr(), o8.stride(0)) == vs.ERR_INVALID
        else:
            assert vs.lib.vstab_pull_frame_bgr16(h, o16.data_ptr(), o16.stride(0) * 2) == vs.ERR_INVALID
        vs.lib.vstab_destroy(h)
        return outs, rots
    outs8, rots8 = run(8, 0)
    for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
        outs, rots = run(10, blend)
        assert len(outs) == len(outs8) == n - 1
        for i in range(n - 1):
            assert np.array_equal(rots[i], rots8[i]), i                     # same tracker input -> same rotations, to the bit
            p = oracle.map_params(K, Ko, rots[i])
            y16, uv16 = wide[i + 1][:H], wide[i + 1][H:]
            assert np.array_equal(outs[i].view(np.uint16), expect.warp_p010(y16, uv16, p, cw, ch, None, blend)), (blend, i)
    # upstream recycling one surface: the library calls it again only once the 16-bit planes and the narrowed luma have been copied
    outs_r, rots_r = run(10, vs.BLEND_EXACT, recycle=True)
    outs_e, rots_e = run(10, vs.BLEND_EXACT)
    assert len(outs_r) == n - 1 and all(np.array_equal(a, b) for a, b in zip(outs_r, outs_e)) and all(np.array_equal(a, b) for a, b in zip(rots_r, rots_e))
    assert max(oracle.rotation_angle(R) for R in rots8) > 1e-4


def test_p010_ring_source_with_readout_rotations(vs, cuda):
    """vstab_ring_source_create_ex (what bench.py --workload 4k-p010 drives): P010 ring frames used in place, a read-out
    rotation per ring frame, fp16 blend; every emitted frame against the oracle under the rotations the handle reports."""
    import torch
    import synth
    W, H, n, r = 640, 360, 10, 2
    K = oracle.get_preset_camera(4, W, H)
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    frames8, _ = synth.shaky_clip(5, K, W, H, n, sigma=0.004)
    rng = np.random.default_rng(4)
    wide = [((f.astype(np.uint16) << 8) | rng.integers(0, 256, f.shape, dtype=np.uint16)) for f in frames8]
    dev = [torch.from_numpy(x.view(np.int16)).to(cuda) for x in wide]
    readouts = [oracle.rodrigues(np.array([0.003 * np.sin(k), -0.002 * np.cos(k), 0.004 * np.sin(2 * k + 1)])) for k in range(n)]
    stab = vs.Stabilizer(dev, total=n, bit_depth=10, readouts=readouts, smooth_radius=r, seed=3, pixel_depth=10, blend=vs.BLEND_FP16)
    assert stab.out_size == (cw, ch)
    i = 0
    while True:
        o = torch.empty((ch, cw, 3), dtype=torch.int16, device=cuda)
        if not stab.pull_bgr16_into(o):
            break
        Wr = stab.warp_rotation(i)
        p = oracle.map_params(K, Ko, Wr)
        rb = oracle.map_params(K, Ko, readouts[i + 1] @ Wr)[8:]
        exp = expect.warp_p010(wide[i + 1][:H], wide[i + 1][H:], p, cw, ch, rb, 1)
        assert np.array_equal(o.cpu().numpy().view(np.uint16), exp), i
        i += 1
    assert i == n - 1


def test_p010_identity_map_is_the_colour_conversion_at_4k(vs, cuda):
    """Size-independent property (cf. test_warp_properties_at_full_size...): pinhole cameras with a power-of-two focal
    length and integer principal points make the rect -> rect map exact, so with identical cameras the 10-bit warp IS the
    10-bit colour conversion, for either blend (one tap has weight 1024: 1.0 in binary16)."""
    w, h = 3840, 2160
    y, uv, _, _ = p010_frame(35, w, h)
    K = np.array([[2048.0, 0, 1920.0], [0, 2048.0, 1080.0], [0, 0, 1]])
    p = oracle.map_params(K, K, np.eye(3))
    exp = oracle.cvt_p010_bgr10(y, uv)
    for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
        assert np.array_equal(host(vs.warp_p010(dev16(y, cuda), dev16(uv, cuda), p, w, h, None, vs.MAP_RECT_TO_RECT, blend)), exp), blend


@pytest.mark.parametrize("w,h", [(640, 360), (323, 181), (66, 34), (2, 2), (1, 1)])
def test_bgr16_to_p010_bit_exact_vs_oracle(vs, cuda, w, h):
    """vstab_cvt_bgr16_p010 (the 10-bit path's encoder hand-off) against its definition, odd sizes and pitched planes included."""
    import torch
    rng = np.random.default_rng(w * 7 + h)
    bgr = rng.integers(0, 1024, (h, w, 3), dtype=np.uint16)
    ey, euv = oracle.cvt_bgr10_p010(bgr)
    y, uv = vs.cvt_bgr16_p010(dev16(bgr, cuda))
    assert np.array_equal(host(y), ey) and np.array_equal(host(uv), euv)
    # pitched source and destinations
    big = torch.zeros((h, w + 5, 3), dtype=torch.int16, device=cuda)
    big[:, :w] = dev16(bgr, cuda)
    oy = torch.full((h, w + 6), -1, dtype=torch.int16, device=cuda)
    ouv = torch.full(((h + 1) // 2, 2 * ((w + 1) // 2) + 4), -1, dtype=torch.int16, device=cuda)
    vs.cvt_bgr16_p010(big[:, :w], oy[:, :w], ouv[:, :2 * ((w + 1) // 2)])
    assert np.array_equal(host(oy)[:, :w], ey) and np.array_equal(host(ouv)[:, :2 * ((w + 1) // 2)], euv)
    assert (host(oy)[:, w:] == 0xffff).all() and (host(ouv)[:, 2 * ((w + 1) // 2):] == 0xffff).all()   # nothing written beyond the planes


def test_pipeline_emits_p010_planes(vs, cuda):
    """vstab_pull_frame_p010: the 10-bit pipeline's frames as P010 planes = the oracle's BGR frame through the oracle's BGR -> P010."""
    import torch
    import synth
    W, H, n, r = 640, 360, 8, 2
    K = oracle.get_preset_camera(4, W, H)
    Ko, (cw, ch) = oracle.get_output_camera(K, W, H)
    frames8, _ = synth.shaky_clip(6, K, W, H, n, sigma=0.004)
    rng = np.random.default_rng(8)
    wide = [((f.astype(np.uint16) << 8) | rng.integers(0, 256, f.shape, dtype=np.uint16)) for f in frames8]
    dev = [torch.from_numpy(x.view(np.int16)).to(cuda) for x in wide]
    stab = vs.Stabilizer(dev, total=n, bit_depth=10, smooth_radius=r, seed=3, pixel_depth=10, blend=vs.BLEND_EXACT)
    i = 0
    while True:
        oy = torch.empty((ch, cw), dtype=torch.int16, device=cuda)
        ouv = torch.empty(((ch + 1) // 2, 2 * ((cw + 1) // 2)), dtype=torch.int16, device=cuda)
        if not stab.pull_p010_into(oy, ouv):
            break
        p = oracle.map_params(K, Ko, stab.warp_rotation(i))
        ey, euv = oracle.cvt_bgr10_p010(expect.warp_p010(wide[i + 1][:H], wide[i + 1][H:], p, cw, ch, None, 0))
        assert np.array_equal(host(oy), ey) and np.array_equal(host(ouv), euv), i
        i += 1
    assert i == n - 1


@pytest.mark.parametrize("w,h", [(640, 360), (328, 182), (72, 34)])
def test_p010_warp_with_planes_out_equals_warp_then_conversion(vs, cuda, w, h):
    """vstab_warp_p010_planes (P010 out of the warp kernel itself) == the oracle's warp followed by the oracle's BGR -> P010,
    both blends, per-row rotation, odd output sizes, pitched output planes; planes the tiled kernel cannot take are refused
    with ERR_UNSUPPORTED (vstab_pull_frame_p010 then converts a 16-bit BGR frame)."""
    import torch
    y, uv, _, _ = p010_frame(33, w, h)
    yd, ud = dev16(y, cuda), dev16(uv, cuda)
    K = oracle.get_preset_camera(4, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.03, -0.01, -0.01)))[8:]
    for rot in [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3)]:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rot))
        for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
            for rot_bottom in (None, rb):
                ey, euv = oracle.cvt_bgr10_p010(oracle.warp_p010(y, uv, p, cw, ch, rot_bottom, 0, blend))
                gy, guv = vs.warp_p010_planes(yd, ud, p, cw, ch, rot_bottom, 0, blend)
                assert np.array_equal(host(gy), ey) and np.array_equal(host(guv), euv), (rot, blend, rot_bottom is not None)
    # pitched, only 2-byte / 4-byte aligned output planes (the scalar store path)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.02, -0.03, 0.01)))
    ey, euv = oracle.cvt_bgr10_p010(oracle.warp_p010(y, uv, p, cw, ch, None, 0, 0))
    oy = torch.full((ch, cw + 7), -1, dtype=torch.int16, device=cuda)
    ouv = torch.full(((ch + 1) // 2, 2 * ((cw + 1) // 2) + 6), -1, dtype=torch.int16, device=cuda)
    vs.warp_p010_planes(yd, ud, p, cw, ch, None, 0, 0, out_y=oy[:, 1:cw + 1], out_uv=ouv[:, 2:2 * ((cw + 1) // 2) + 2])
    assert np.array_equal(host(oy)[:, 1:cw + 1], ey) and np.array_equal(host(ouv)[:, 2:2 * ((cw + 1) // 2) + 2], euv)
    assert (host(oy)[:, 0] == 0xffff).all() and (host(oy)[:, cw + 1:] == 0xffff).all() and (host(ouv)[:, :2] == 0xffff).all()
    # unaligned source planes: not this entry point's job
    big = torch.zeros((h, w + 8), dtype=torch.int16, device=cuda)
    big[:, 1:w + 1] = yd
    with pytest.raises(vs.VstabError) as e:
        vs.warp_p010_planes(big[:, 1:w + 1], ud, p, cw, ch)
    assert e.value.status == vs.ERR_UNSUPPORTED
