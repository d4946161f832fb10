"""a9 pinned against the reference's OWN kernel on the GPU it runs on here.

oracle/_ref/createMap.gfx950.co is /root/reference/opencv/createMap.cl compiled unmodified by ROCm's OpenCL front end
for gfx950 (oracle/Makefile `ref_gfx950`; no stand-ins).  oracle.create_map_ref_gfx950 launches it the way
FrameSourceWarp.cpp:272-304 does.  Against it:

  * VSTAB_MAP_CREATEMAP_CL_OPENCL (mode 5) must be BIT-IDENTICAL: map planes (the literal instruction stream), the
    quantised map and the fused kernel (its shortened form) -- no tolerance.
  * VSTAB_MAP_CREATEMAP_CL (mode 0, every operation IEEE-rounded, the CPU-reproducible regime) deviates in the last
    bits; the deviation is measured on the whole 4K map and pinned here.
"""
import json
import os

import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROTS = [(0.0, 0.0, 0.0), (0.02, -0.03, 0.01), (-0.15, 0.1, 0.3), (0.6, -0.4, 0.2), (0.0, 1.7, 0.0)]
OCL = 5  # VSTAB_MAP_CREATEMAP_CL_OPENCL


def dev(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def cams(w, h, preset=4, scale=1.0, crop=False):
    K = oracle.get_preset_camera(preset, w, h)
    Ko, size = oracle.get_output_camera(K, w, h, scale, crop)
    return K, Ko, size


def ulp_distance(a, b):
    ai, bi = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ai, bi = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai), np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def same_bits(a, b):
    """Bit-identical, NaNs matching NaNs (a NaN's payload is not part of the contract)."""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


@pytest.fixture(scope="module")
def refcl(cuda):
    if not oracle.ref_gfx950_available():
        pytest.fail("oracle/_ref/createMap.gfx950.co or its launcher is missing (run `make -C oracle` where /root/reference exists)")
    return oracle.create_map_ref_gfx950


def test_reference_kernel_runs_and_ignores_the_work_group_shape(refcl, cuda):
    K, Ko, (cw, ch) = cams(640, 360)
    p = oracle.map_params(K, Ko, oracle.rodrigues(ROTS[1]))
    ax, ay = refcl(p, cw, ch)
    bx, by = refcl(p, cw, ch, block=(16, 16))
    cx, cy = refcl(p, cw, ch, block=(256, 1))
    assert same_bits(ax, bx) and same_bits(ay, by) and same_bits(ax, cx) and same_bits(ay, cy)
    assert np.isfinite(ax).all() and np.isfinite(ay).all()
    # it is the same function as the IEEE restatement up to OpenCL's error bounds
    ox, oy = oracle.create_map(p, cw, ch)
    assert np.abs(ax - ox).max() < 1e-3 and np.abs(ay - oy).max() < 1e-3


def test_reference_kernel_nan_on_the_optical_axis(refcl, vs, cuda):
    """createMap.cl:38-39: radius 0 -> atan(0)/0 -> NaN, also in the OpenCL build; mode 5 reproduces it."""
    p = np.array([50, 40, 100, 100, 8, 5, 10, 10, 1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32)
    rx, ry = refcl(p, 16, 12)
    assert np.isnan(rx[5, 8]) and np.isnan(ry[5, 8]) and np.isnan(rx).sum() == 1
    mx, my = vs.create_map(p, 16, 12, mode=OCL)
    assert same_bits(mx.cpu().numpy(), rx) and same_bits(my.cpu().numpy(), ry)


@pytest.mark.parametrize("w,h,preset,scale", [(128, 72, 4, 1.0), (1920, 1080, 4, 1.0), (3840, 2160, 4, 1.0), (1920, 1440, 1, 0.5)])
def test_opencl_mode_map_planes_bit_identical_to_the_reference_kernel(refcl, vs, cuda, w, h, preset, scale):
    K, Ko, (cw, ch) = cams(w, h, preset, scale)
    for rv in ROTS:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        rx, ry = refcl(p, cw, ch)
        mx, my = vs.create_map(p, cw, ch, mode=OCL)
        assert same_bits(mx.cpu().numpy(), rx), (w, rv)
        assert same_bits(my.cpu().numpy(), ry), (w, rv)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_opencl_mode_quantised_map_is_the_rounded_reference_map(refcl, vs, cuda, w, h):
    """The shortened instruction stream (one reciprocal per division, shared with atan's argument reduction) gives the
    integers cv::remap makes of the reference kernel's map -- every entry."""
    K, Ko, (cw, ch) = cams(w, h)
    for rv in ROTS:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        rx, ry = refcl(p, cw, ch)
        q = vs.quantised_map(p, cw, ch, mode=OCL).cpu().numpy().view(np.int32).reshape(ch, -1, 2)[:, :cw]
        ok = ~(np.isnan(rx) | np.isnan(ry))
        with np.errstate(invalid="ignore"):
            ex, ey = np.rint(rx * np.float32(32)), np.rint(ry * np.float32(32))
        inr = ok & (np.abs(ex) < 2 ** 30) & (np.abs(ey) < 2 ** 30)
        assert np.array_equal(q[..., 0][inr], ex[inr].astype(np.int64)), (w, rv)
        assert np.array_equal(q[..., 1][inr], ey[inr].astype(np.int64)), (w, rv)
        assert (q[..., 0][~ok] == np.iinfo(np.int32).min).all()


@pytest.mark.parametrize("w,h,preset,scale,crop", [
    (128, 72, 4, 1.0, False), (130, 74, 1, 0.5, False), (1920, 1080, 4, 1.0, False), (1920, 1080, 4, 1.0, True), (3840, 2160, 4, 1.0, False)])
def test_fused_warp_opencl_mode_equals_reference_kernel_then_remap(refcl, vs, cuda, w, h, preset, scale, crop):
    """The whole a2 + a9 + a10 chain with the map taken from the reference's own kernel: cvtColor (oracle) ->
    createMap (REFERENCE, on this GPU) -> cv::remap (oracle) == the fused kernel in mode 5, every byte."""
    K, Ko, (cw, ch) = cams(w, h, preset, scale, crop)
    frame = synth.nv12(21, w, h, full_range=(w < 200))
    bgr = oracle.cvt_nv12_bgr(frame)
    fd = dev(frame, cuda)
    for rv in ROTS[:4] if w > 2000 else ROTS:
        p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
        rx, ry = refcl(p, cw, ch)
        exp = oracle.remap_bilinear(bgr, rx, ry)
        got = vs.warp_nv12(fd, p, cw, ch, mode=OCL).cpu().numpy()
        assert np.array_equal(got, exp), (w, h, rv, int((got != exp).sum()))
        y, uv = vs.warp_nv12(fd, p, cw, ch, mode=OCL, out_format=vs.OUT_NV12)
        ey, euv = oracle.cvt_bgr_nv12(exp)
        assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(uv.cpu().numpy().reshape(euv.shape), euv), (w, h, rv)


def deviation_table(a, b, centre, bgr=None):
    ok = ~np.isnan(b)
    big = ok & (np.abs(b) > 256)
    d = ulp_distance(a, b)[big]
    rel = np.abs(a - b)[ok] / (np.abs(b[ok]) + abs(float(centre)))
    with np.errstate(invalid="ignore"):
        flips = float((np.rint(a[ok] * np.float32(32)) != np.rint(b[ok] * np.float32(32))).mean())
    return {"identical": float((d == 0).mean()), "gt1ulp": float((d > 1).mean()), "max_ulp": int(d.max()),
            "rel_max_x2p23": float(rel.max() * 2.0 ** 23), "bucket_flips": flips}


def test_ieee_mode_deviation_from_the_reference_kernel_on_gfx950_is_pinned(refcl, vs, cuda):
    """Mode 0 (the default: every operation IEEE-rounded, reproducible by the CPU oracle) against the reference's own
    kernel as it runs on THIS device, on the three 4K golden parameter sets: ULP histogram of the map entries, rate of
    1/32-px bucket flips in cv::remap's quantisation, and the difference in grey levels after cv::remap.  The measured
    table also goes to gpurun_out/ (DESIGN.md section 3 quotes it)."""
    ref = np.load(os.path.join(GOLD, "createmap_ref.npz"))
    cw, ch = (int(v) for v in ref["uhd_size"])
    frame = synth.nv12(3, 3840, 2160)
    bgr = oracle.cvt_nv12_bgr(frame)
    table = {}
    for i in range(3):
        p = ref[f"uhd_params_{i}"]
        rx, ry = refcl(p, cw, ch)
        mx, my = vs.create_map(p, cw, ch)
        mx, my = mx.cpu().numpy(), my.cpu().numpy()
        assert np.array_equal(np.isnan(mx), np.isnan(rx)) and np.array_equal(np.isnan(my), np.isnan(ry))
        tx, ty = deviation_table(mx, rx, p[0]), deviation_table(my, ry, p[1])
        a = vs.remap_bilinear(dev(bgr, cuda), dev(mx, cuda), dev(my, cuda)).cpu().numpy().astype(np.int16)
        b = vs.remap_bilinear(dev(bgr, cuda), dev(rx, cuda), dev(ry, cuda)).cpu().numpy().astype(np.int16)
        d = np.abs(a - b)
        grey = {"bytes_changed": float((d > 0).mean()), "max_levels": int(d.max()), "more_than_one": float((d > 1).mean())}
        table[f"params_{i}"] = {"x": tx, "y": ty, "grey": grey}
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "a9_ieee_vs_refcl_gfx950.json"), "w") as f:
        json.dump(table, f, indent=1)
    for i in range(3):
        for t in (table[f"params_{i}"]["x"], table[f"params_{i}"]["y"]):
            # measured (profiles/history/r03_a9_ieee_vs_reference_gfx950.json): 51-58 % of the entries identical, 9-16 % differ by more
            # than one ULP, at most 21; <= 4.1 x 2^-23 of the (map - centre) term; 0.19-0.31 % of the 1/32-px buckets flip
            assert t["identical"] > 0.45 and t["gt1ulp"] < 0.20 and t["max_ulp"] <= 28 and t["rel_max_x2p23"] <= 5.5 and t["bucket_flips"] < 0.004, (i, t)
        grey = table[f"params_{i}"]["grey"]
        # after cv::remap: 2.0e-4 of the output bytes change, by at most 7 levels, 1.2e-5 of them by more than one
        assert grey["bytes_changed"] < 3e-4 and grey["max_levels"] <= 10 and grey["more_than_one"] < 2e-5, (i, grey)


def test_x86_cross_check_build_vs_the_gfx950_build_of_the_same_kernel(refcl, cuda):
    """The committed golden crops come from the x86 build of createMap.cl with stand-in built-ins (libm atanf, IEEE
    divide / sqrt).  Against the real device build they differ as two OpenCL implementations may -- recorded so that
    nobody mistakes the x86 artefact for the reference's GPU behaviour."""
    ref = np.load(os.path.join(GOLD, "createmap_ref.npz"))
    cw, ch = (int(v) for v in ref["uhd_size"])
    worst = 0
    for i in range(3):
        p = ref[f"uhd_params_{i}"]
        rx, ry = refcl(p, cw, ch)
        for c, (x, y) in enumerate(ref["uhd_crops"]):
            for g, r, centre in ((ref[f"uhd_mapx_{i}"][c], rx[y:y + 32, x:x + 32], p[0]), (ref[f"uhd_mapy_{i}"][c], ry[y:y + 32, x:x + 32], p[1])):
                ok = ~np.isnan(g)
                assert np.array_equal(np.isnan(g), np.isnan(r))
                rel = np.abs(g - r)[ok] / (np.abs(g[ok]) + abs(float(centre)))
                worst = max(worst, float(rel.max()) * 2 ** 23)
    assert worst <= 6.0, worst


def test_pipeline_object_in_opencl_map_precision(refcl, vs, cuda):
    """vstab_config.map_precision = OPENCL: the pipeline's decisions and rotations are those of the default handle (the
    tracker does not see the map), and every emitted frame is cvtColor -> the REFERENCE kernel's map (on this GPU) ->
    cv::remap of the frame it warps, under the rotation the handle reports."""
    import torch
    w, h = 640, 360
    K = oracle.get_preset_camera(oracle.GOPRO_H4B_WIDE169_MEASURED, w, h)
    Ko, (cw, ch) = oracle.get_output_camera(K, w, h)
    frames, _ = synth.shaky_clip(2, K, w, h, 14, sigma=0.004)
    dev_frames = [torch.from_numpy(f).to(cuda) for f in frames]
    outs = {}
    for prec in (0, 1):
        stab = vs.Stabilizer(dev_frames, total=len(frames), smooth_radius=3, seed=3, map_precision=prec)
        got = []
        while True:
            o = stab.pull()
            if o is None:
                break
            got.append(o.cpu().numpy())
        outs[prec] = (got, [stab.warp_rotation(i) for i in range(len(got))], stab.frame_log())
        stab.close()
    (g0, r0, l0), (g1, r1, l1) = outs[0], outs[1]
    assert len(g0) == len(g1) == len(frames) - 1
    assert all(np.array_equal(a, b) for a, b in zip(r0, r1)) and [x["inliers"] for x in l0] == [x["inliers"] for x in l1]
    differ = 0
    for i, (img, R) in enumerate(zip(g1, r1)):
        p = oracle.map_params(K, Ko, R)
        rx, ry = refcl(p, cw, ch)
        assert np.array_equal(img, oracle.remap_bilinear(oracle.cvt_nv12_bgr(frames[i + 1]), rx, ry)), i
        differ += int((img != g0[i]).sum())
    assert differ > 0   # the two precisions are different maps (a few bytes per frame), not the same mode twice


def test_opencl_precision_is_the_default_of_the_pipeline_object(vs):
    """vstab_config_default: the reference's map is what ITS kernel computes on this GPU, so that is the handle's default."""
    cfg = vs.default_config()
    assert cfg.map_precision == vs.MAP_PRECISION_OPENCL == 1 and cfg.abi_version == vs.ABI_VERSION


@pytest.mark.parametrize("w,h", [(322, 182), (640, 360), (1920, 1080)])
def test_rolling_shutter_warp_opencl_mode_is_the_reference_kernel_row_by_row(refcl, vs, cuda, w, h):
    """vstab_warp_nv12_rs in mode 5.  The per-row matrix m(y) is this library's definition (config 5 has no reference
    counterpart); what is done with it is the reference's: row y of the output must be row y of what createMap.cl's code object
    returns when it is handed m(y) as its rotation (oracle.create_map_ref_gfx950_rs: one launch of the reference kernel per
    row), then cv::remap.  BGR and NV12 output; equal first / last rotations reproduce the per-frame warp."""
    K, Ko, (cw, ch) = cams(w, h)
    frame = synth.nv12(23, w, h)
    fd = dev(frame, cuda)
    bgr = oracle.cvt_nv12_bgr(frame)
    for top, bottom in (((0.01, -0.02, 0.005), (0.03, -0.01, -0.01)), ((-0.15, 0.1, 0.3), (-0.13, 0.12, 0.28))):
        p = oracle.map_params(K, Ko, oracle.rodrigues(top))
        rb = oracle.map_params(K, Ko, oracle.rodrigues(bottom))[8:]
        rx, ry = oracle.create_map_ref_gfx950_rs(p, rb, cw, ch)
        exp = oracle.remap_bilinear(bgr, rx, ry)
        got = vs.warp_nv12_rs(fd, p, rb, cw, ch, OCL).cpu().numpy()
        assert np.array_equal(got, exp), (w, top, int((got != exp).sum()))
        if w < 1000:
            y, uv = vs.warp_nv12_rs(fd, p, rb, cw, ch, OCL, vs.OUT_NV12)
            ey, euv = oracle.cvt_bgr_nv12(exp)
            assert np.array_equal(y.cpu().numpy(), ey) and np.array_equal(uv.cpu().numpy().reshape(euv.shape), euv)
        # and it is not the IEEE-mode warp with another name
        assert w < 1000 or not np.array_equal(got, vs.warp_nv12_rs(fd, p, rb, cw, ch, 0).cpu().numpy())
    # the first row's launch of the per-row checker IS the plain reference map's first row; equal rotations: the whole map
    same_x, same_y = oracle.create_map_ref_gfx950_rs(p, p[8:], cw, ch)
    px, py = refcl(p, cw, ch)
    assert same_bits(same_x, px) and same_bits(same_y, py)
    assert np.array_equal(vs.warp_nv12_rs(fd, p, p[8:], cw, ch, OCL).cpu().numpy(), vs.warp_nv12(fd, p, cw, ch, mode=OCL).cpu().numpy())


def test_nearest_warp_opencl_mode_equals_reference_kernel_then_nearest_remap(refcl, vs, cuda):
    """INTER_NEAREST (FrameSourceWarp.hpp:90) with the reference kernel's map: cvRound + saturate_cast<short> of every entry."""
    for w, h in ((640, 360), (130, 74)):
        K, Ko, (cw, ch) = cams(w, h)
        frame = synth.nv12(24, w, h)
        for rv in ROTS[:4]:
            p = oracle.map_params(K, Ko, oracle.rodrigues(rv))
            exp = oracle.remap_nearest(oracle.cvt_nv12_bgr(frame), *refcl(p, cw, ch))
            got = vs.warp_nv12_nearest(dev(frame, cuda), p, cw, ch, mode=OCL).cpu().numpy()
            assert np.array_equal(got, exp), (w, rv)
    with pytest.raises(vs.VstabError):
        vs.warp_nv12_nearest(dev(frame, cuda), p, cw, ch, mode=1)


def _p010(seed, w, h):
    from test_p010_cpu import p010_frame
    y, uv, _, _ = p010_frame(seed, w, h)
    return y, uv


def _dev16(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(cuda)


@pytest.mark.parametrize("w,h", [(640, 360), (328, 182), (72, 34)])
def test_p010_warp_opencl_mode_equals_reference_kernel_then_10bit_remap(refcl, vs, cuda, w, h):
    """The 10-bit pixel path (config 5) with the reference kernel's map: 10-bit conversion (oracle) -> createMap (REFERENCE, on this
    GPU; per row for the rolling-shutter variant) -> 10-bit remap (oracle), both blends, through the LDS-tiled kernel (aligned
    planes), the direct-gather kernel (planes that are 2-byte-aligned views) and with P010 planes out."""
    import torch
    y, uv = _p010(41, w, h)
    yd, ud = _dev16(y, cuda), _dev16(uv, cuda)
    K, Ko, (cw, ch) = cams(w, h)
    bgr10 = oracle.cvt_p010_bgr10(y, uv)
    # unaligned views of the same samples: the direct-gather kernel
    Yb = torch.zeros((h + 1, w + 6), dtype=torch.int16, device=cuda)
    Ub = torch.zeros((h // 2 + 1, w + 6), dtype=torch.int16, device=cuda)
    Yb[1:, 2:w + 2] = yd
    Ub[1:, 2:w + 2] = ud
    for top, bottom in (((0.0, 0.0, 0.0), None), ((0.02, -0.03, 0.01), None), ((0.01, -0.02, 0.005), (0.03, -0.01, -0.01))):
        p = oracle.map_params(K, Ko, oracle.rodrigues(top))
        rb = None if bottom is None else oracle.map_params(K, Ko, oracle.rodrigues(bottom))[8:]
        rx, ry = refcl(p, cw, ch) if rb is None else oracle.create_map_ref_gfx950_rs(p, rb, cw, ch)
        for blend in (vs.BLEND_EXACT, vs.BLEND_FP16):
            exp = oracle.remap_bilinear10(bgr10, rx, ry, blend)
            got = vs.warp_p010(yd, ud, p, cw, ch, rb, OCL, blend).cpu().numpy().view(np.uint16)
            assert np.array_equal(got, exp), (w, top, blend, int((got != exp).sum()))
            got = vs.warp_p010(Yb[1:, 2:w + 2], Ub[1:, 2:w + 2], p, cw, ch, rb, OCL, blend).cpu().numpy().view(np.uint16)
            assert np.array_equal(got, exp), ("direct", w, top, blend)
            oy, ouv = vs.warp_p010_planes(yd, ud, p, cw, ch, rb, OCL, blend)
            ey, euv = oracle.cvt_bgr10_p010(exp)
            assert np.array_equal(oy.cpu().numpy().view(np.uint16), ey) and np.array_equal(ouv.cpu().numpy().view(np.uint16), euv), ("planes", w, top, blend)


def test_p010_warp_opencl_mode_at_4k_config5(refcl, vs, cuda):
    """BASELINE config 5 at full size in the reference kernel's arithmetic: 4K P010, fp16 blend, a rotation per output row
    (1999 launches of the reference kernel, one per row, make the checker's map)."""
    w, h = 3840, 2160
    y, uv = _p010(42, w, h)
    K, Ko, (cw, ch) = cams(w, h)
    p = oracle.map_params(K, Ko, oracle.rodrigues((0.02, -0.03, 0.01)))
    rb = oracle.map_params(K, Ko, oracle.rodrigues((0.022, -0.028, 0.012)))[8:]
    rx, ry = oracle.create_map_ref_gfx950_rs(p, rb, cw, ch)
    exp = oracle.remap_bilinear10(oracle.cvt_p010_bgr10(y, uv), rx, ry, 1)
    got = vs.warp_p010(_dev16(y, cuda), _dev16(uv, cuda), p, cw, ch, rb, OCL, vs.BLEND_FP16).cpu().numpy().view(np.uint16)
    assert np.array_equal(got, exp), int((got != exp).sum())


def test_undistort_only_pipeline_at_1080p_warps_from_the_cached_reference_map(refcl, vs, cuda):
    """BASELINE config 1 geometry through the pipeline object with its defaults (tracking off): from the third frame on the
    handle warps from the quantised map it wrote once (the CACHED kernel) -- that map is the reference kernel's, rounded."""
    import torch
    w, h = 1920, 1080
    K, Ko, (cw, ch) = cams(w, h)
    frames = [synth.nv12(50 + i, w, h) for i in range(6)]
    stab = vs.Stabilizer([torch.from_numpy(f).to(cuda) for f in frames], total=len(frames), smooth_radius=1, tracking=0)
    p = oracle.map_params(K, Ko, np.eye(3))
    rx, ry = refcl(p, cw, ch)
    i = 0
    while True:
        o = stab.pull()
        if o is None:
            break
        assert np.array_equal(o.cpu().numpy(), oracle.remap_bilinear(oracle.cvt_nv12_bgr(frames[i + 1]), rx, ry)), i
        i += 1
    assert i == len(frames) - 1
