"""GPU parity tests of the pixel path (a1, a2, a9, a10 and the fused kernel) through the C ABI.

Bar: bit-exact against the CPU oracle for every byte and every map float.  Against the golden
vectors of the reference's own createMap.cl the only difference is atan rounding (documented
tolerance in the test)."""
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROTS = [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3), (0.6, -0.4, 0.2), (0.0, 1.7, 0.0)]


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def cams(w, h, preset=4, scale=1.0, crop=False):
    K = oracle.get_preset_camera(preset, w, h)
    Ko, size = oracle.get_output_camera(K, w, h, scale, crop)
    return K, Ko, size


def test_pack_nv12_pitched_and_unaligned(vs, cuda):
    rng = np.random.default_rng(0)
    for w, h, py, puv, off in [(64, 36, 64, 64, 0), (64, 36, 80, 96, 0), (30, 18, 37, 41, 3), (1920, 1080, 2048, 1920, 0)]:
        ybuf = rng.integers(0, 256, (h, py + off), dtype=np.uint8)
        uvbuf = rng.integers(0, 256, (h // 2, puv + off), dtype=np.uint8)
        yd, uvd = dev(ybuf, cuda), dev(uvbuf, cuda)
        out = vs.pack_nv12(yd[:, off:off + w], uvd[:, off:off + w]).cpu().numpy()
        assert np.array_equal(out, oracle.pack_nv12(ybuf[:, off:off + w], uvbuf[:, off:off + w]))


def test_pack_p010_narrowing(vs, cuda):
    import torch
    rng = np.random.default_rng(3)
    for w, h, pitch, off in [(64, 36, 64, 0), (1920, 1080, 1920, 0), (30, 18, 37, 3), (72, 20, 80, 4)]:
        yb = rng.integers(0, 65536, (h, pitch + off), dtype=np.uint16)
        ub = rng.integers(0, 65536, (h // 2, pitch + off), dtype=np.uint16)
        yd, ud = torch.from_numpy(yb.view(np.int16)).to(cuda), torch.from_numpy(ub.view(np.int16)).to(cuda)
        got = vs.pack_p010(yd[:, off:off + w], ud[:, off:off + w]).cpu().numpy()
        assert np.array_equal(got, oracle.pack_p010(yb[:, off:off + w], ub[:, off:off + w])), (w, h, pitch, off)
    with pytest.raises(vs.VstabError):
        vs.pack_p010(yd[:, :31], ud[:, :31])


def test_cvt_nv12_bgr_bit_exact(vs, cuda):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    assert np.array_equal(vs.cvt_nv12_bgr(dev(kat["cvt_nv12"], cuda)).cpu().numpy(), kat["cvt_bgr"])
    for seed, w, h, full in [(1, 64, 36, True), (2, 130, 74, True), (3, 1920, 1080, False), (4, 6, 2, True)]:
        f = synth.nv12(seed, w, h, full_range=full)
        assert np.array_equal(vs.cvt_nv12_bgr(dev(f, cuda)).cpu().numpy(), oracle.cvt_nv12_bgr(f)), (w, h)


def test_create_map_bit_exact_vs_oracle(vs, cuda):
    for (w, h) in [(128, 72), (1920, 1080)]:
        K, Ko, (cw, ch) = cams(w, h)
        for rv in ROTS:
            p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
            mx, my = vs.create_map(p, cw, ch)
            ox, oy = oracle.create_map(p, cw, ch)
            assert np.array_equal(mx.cpu().numpy().view(np.uint32), ox.view(np.uint32)), (w, rv)
            assert np.array_equal(my.cpu().numpy().view(np.uint32), oy.view(np.uint32)), (w, rv)


def test_create_map_nan_at_optical_axis(vs, cuda):
    """createMap.cl:38-39: radius 0 -> atan(0)/0 = NaN map entry."""
    p = np.array([50, 40, 100, 100, 8, 5, 10, 10, 1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    mx, my = vs.create_map(p, 16, 12)
    ox, oy = oracle.create_map(p, 16, 12)
    assert np.isnan(ox[5, 8]) and np.isnan(oy[5, 8])
    assert np.array_equal(mx.cpu().numpy().view(np.uint32) & 0x7FC00000 == 0x7FC00000, np.isnan(ox))
    ok = ~np.isnan(ox)
    assert np.array_equal(mx.cpu().numpy()[ok], ox[ok]) and np.array_equal(my.cpu().numpy()[ok], oy[ok])


def test_create_map_vs_reference_kernel_golden(vs, cuda):
    ref = np.load(os.path.join(GOLD, "createmap_ref.npz"))
    cw, ch = (int(v) for v in ref["uhd_size"])
    flips = []
    for i in range(3):
        p = ref[f"uhd_params_{i}"]
        mx, my = vs.create_map(p, cw, ch)
        mx, my = mx.cpu().numpy(), my.cpu().numpy()
        for c, (x, y) in enumerate(ref["uhd_crops"]):
            gx, gy = ref[f"uhd_mapx_{i}"][c], ref[f"uhd_mapy_{i}"][c]
            ax, ay = mx[y:y + 32, x:x + 32], my[y:y + 32, x:x + 32]
            # atan differs by <= ~2 ulp relative; it scales the (map - centre) term
            assert (np.abs(ax - gx) <= 3 * 2.0 ** -23 * (np.abs(gx) + abs(p[0]))).all()
            assert (np.abs(ay - gy) <= 3 * 2.0 ** -23 * (np.abs(gy) + abs(p[1]))).all()
            flips.append((np.rint(ax * 32) != np.rint(gx * 32)).mean())
            flips.append((np.rint(ay * 32) != np.rint(gy * 32)).mean())
    assert np.mean(flips) < 0.005   # 1/32-px bucket flips against libm-atanf build of the reference kernel


def test_remap_bilinear_special_cases_and_bgr(vs, cuda):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    out = vs.remap_bilinear(dev(kat["remap_src"], cuda), dev(kat["remap_mx"], cuda), dev(kat["remap_my"], cuda))
    assert np.array_equal(out.cpu().numpy(), kat["remap_dst"])
    K, Ko, (cw, ch) = cams(128, 72)
    bgr = oracle.cvt_nv12_bgr(synth.nv12(5, 128, 72, full_range=True))
    for rv in ROTS:
        mx, my = oracle.create_map(oracle.map_params(K, Ko, oracle.rodrigues(rv)), cw, ch)
        got = vs.remap_bilinear(dev(bgr, cuda), dev(mx, cuda), dev(my, cuda)).cpu().numpy()
        assert np.array_equal(got, oracle.remap_bilinear(bgr, mx, my)), rv


@pytest.mark.parametrize("w,h,preset,scale,crop", [
    (128, 72, 4, 1.0, False), (128, 72, 4, 1.0, True), (130, 74, 1, 0.5, False),
    (1920, 1080, 4, 1.0, False), (1920, 1440, 1, 0.5, False)])
def test_fused_warp_bit_exact_vs_oracle(vs, cuda, w, h, preset, scale, crop):
    K, Ko, (cw, ch) = cams(w, h, preset, scale, crop)
    frame = synth.nv12(11, w, h, full_range=(w < 200))
    fd = dev(frame, cuda)
    for rv in ROTS:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        got = vs.warp_nv12_bgr(fd, p, cw, ch).cpu().numpy()
        exp = oracle.warp_nv12(frame, p, cw, ch)
        assert np.array_equal(got, exp), (rv, int((got != exp).sum()))


@pytest.mark.parametrize("w,h,preset,scale,crop", [(128, 72, 4, 1.0, False), (130, 74, 1, 0.5, True), (1920, 1080, 4, 1.0, False)])
def test_nearest_warp_bit_exact_vs_oracle(vs, cuda, w, h, preset, scale, crop):
    """INTER_NEAREST (FrameSourceWarp.hpp:90 admits it): cvtColor + createMap + cv::remap nearest, operator by operator."""
    K, Ko, (cw, ch) = cams(w, h, preset, scale, crop)
    frame = synth.nv12(12, w, h)
    bgr = oracle.cvt_nv12_bgr(frame)
    fd = dev(frame, cuda)
    for rv in ROTS:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        mx, my = oracle.create_map(p, cw, ch)
        exp = oracle.remap_nearest(bgr, mx, my)
        got = vs.warp_nv12_nearest(fd, p, cw, ch).cpu().numpy()
        assert np.array_equal(got, exp), (rv, int((got != exp).sum()))
        assert (exp != oracle.warp_nv12(frame, p, cw, ch)).any()   # and it is not the bilinear result


def test_fused_warp_golden(vs, cuda):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    seed, w, h = (int(v) for v in kat["warp_seed"])
    fd = dev(synth.nv12(seed, w, h), cuda)
    for i in range(4):
        g = kat[f"warp_bgr_{i}"]
        assert np.array_equal(vs.warp_nv12_bgr(fd, kat[f"warp_params_{i}"], g.shape[1], g.shape[0]).cpu().numpy(), g)


def test_fused_equals_unfused_operators_at_4k(vs, cuda):
    """Size-independent property at BASELINE.json's full size: the fused kernel must equal
    cvtColor -> createMap -> remap run as separate HIP operators (all three parity-tested above)."""
    import torch
    w, h = 3840, 2160
    K, Ko, (cw, ch) = cams(w, h)
    assert (cw, ch) == (3524, 1999)
    g = torch.Generator(device="cpu").manual_seed(3)
    frame = torch.randint(0, 256, (h * 3 // 2, w), dtype=torch.uint8, generator=g).to(cuda)
    for rv in [(0, 0, 0), (0.03, -0.02, 0.015)]:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        fused = vs.warp_nv12_bgr(frame, p, cw, ch)
        mx, my = vs.create_map(p, cw, ch)
        unfused = vs.remap_bilinear(vs.cvt_nv12_bgr(frame), mx, my)
        assert torch.equal(fused, unfused)
    # and a strip of it against the CPU oracle
    exp = oracle.warp_nv12(frame.cpu().numpy(), p, cw, ch)
    assert np.array_equal(fused.cpu().numpy(), exp)


def test_fused_warp_pitched_output_and_unaligned_tail(vs, cuda):
    import torch
    w, h = 128, 72
    K, Ko, (cw, ch) = cams(w, h)
    frame = synth.nv12(4, w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.02, 0.01, -0.04)))
    exp = oracle.warp_nv12(frame, p, cw, ch)
    for dw in (cw, cw - 1, cw - 2, cw - 3, 5):     # widths not divisible by 4 exercise the tail path
        big = torch.zeros((ch, dw * 3 + 7), dtype=torch.uint8, device=cuda)
        view = big[:, :dw * 3].view(ch, dw * 3)
        out = torch.as_strided(big, (ch, dw, 3), (big.stride(0), 3, 1))
        vs.warp_nv12_bgr(dev(frame, cuda), p, dw, ch, out=out)
        assert np.array_equal(out.cpu().numpy(), exp[:, :dw])
        assert int(big[:, dw * 3:].sum()) == 0   # padding untouched


def test_warp_properties_at_full_size_identity_shift_and_half_pixel(vs, cuda):
    """Size-independent properties at BASELINE's 4K (no oracle warp involved).  With pinhole cameras whose focal length
    is a power of two and whose principal points are integers, the rect -> rect map is exact in fp32:
      identical cameras          -> the output IS cvtColor(input), byte for byte;
      principal point moved by k -> the image shifted by k pixels, zeros where the source ends (BORDER_CONSTANT);
      moved by half a pixel      -> every pixel the rounded mean of two neighbours, (a + b + 1) >> 1 (weights 512 / 512)."""
    import torch
    w, h = 3840, 2160
    frame = synth.nv12(44, w, h)
    fd = dev(frame, cuda)
    bgr = vs.cvt_nv12_bgr(fd)
    K = np.array([[2048.0, 0, 1920.0], [0, 2048.0, 1080.0], [0, 0, 1]])
    ident = vs.warp_nv12(fd, oracle.map_params(K, K, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT, vs.OUT_BGR8)
    assert bool((ident == bgr).all())
    Ks = K.copy()
    Ks[0, 2], Ks[1, 2] = 1920.0 + 37, 1080.0 - 5           # output pixel (x, y) looks at source (x - 37, y + 5)
    sh = vs.warp_nv12(fd, oracle.map_params(K, Ks, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT, vs.OUT_BGR8)
    exp = torch.zeros_like(bgr)
    exp[: h - 5, 37:] = bgr[5:, : w - 37]
    assert bool((sh == exp).all())
    Kh = K.copy()
    Kh[0, 2] = 1920.0 - 0.5                                 # source x + 0.5: mean of columns x and x + 1
    half = vs.warp_nv12(fd, oracle.map_params(K, Kh, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT, vs.OUT_BGR8)
    a, b = bgr[:, :-1].to(torch.int32), bgr[:, 1:].to(torch.int32)
    assert bool((half[:, :-1].to(torch.int32) == ((a + b + 1) >> 1)).all())
    assert bool((half[:, -1].to(torch.int32) == ((bgr[:, -1].to(torch.int32) + 1) >> 1)).all())   # last column: half of it, half border
    # NV12 output of the identity is the conversion of that BGR image (the encoder hand-off adds nothing else)
    yo, uvo = vs.warp_nv12(fd, oracle.map_params(K, K, np.eye(3)), w, h, vs.MAP_RECT_TO_RECT, vs.OUT_NV12)
    ey, euv = oracle.cvt_bgr_nv12(bgr.cpu().numpy())
    assert np.array_equal(yo.cpu().numpy(), ey) and np.array_equal(uvo.cpu().numpy().reshape(euv.shape), euv)


@pytest.mark.gpu
def test_preload_kernels_loads_the_code_objects_and_can_be_repeated(vs, cuda):
    """vstab_preload_kernels: the library's five code objects loaded on request (vstab_create calls it): succeeds, and a
    second call is a no-op."""
    vs.preload_kernels()
    vs.preload_kernels()
