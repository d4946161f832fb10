"""CPU oracle for the stabilisation / undistort hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (video-annotator_amd/) never does; it fails loudly when its HIP library is missing.

Pixel / feature arithmetic: oracle/vstab_oracle.c (plain C, loaded with ctypes).
Host geometry (cameras, point undistortion, Savitzky-Golay rotation filter, look-ahead state
machine): numpy restatements below, each citing the reference file:line it follows
(paths relative to /root/reference/opencv/).

Parity pinning: see the header of vstab_oracle.c.  The reference holds no tests or golden
vectors; only createMap.cl can be executed (oracle/_ref).  Everything that
restates OpenCV / gram_savitzky_golay arithmetic is "parity unpinned".
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "_build", "libvstab_oracle.so")
    src = os.path.join(_HERE, "vstab_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_so = os.path.join(_HERE, "_ref", "libcreatemap_ref.so")
    ref_co = os.path.join(_HERE, "_ref", "createMap.gfx950.co")
    runner = os.path.join(_HERE, "_build", "libref_cl_runner.so")
    runner_src = os.path.join(_HERE, "ref_cl_runner.cpp")
    if os.path.exists("/root/reference/opencv/createMap.cl") and not (os.path.exists(ref_so) and os.path.exists(ref_co)):
        need = True
    if not os.path.exists(runner) or os.path.getmtime(runner) < os.path.getmtime(runner_src):
        need = True
    if need:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(so)
        c = ctypes
        u8p, f32p, i16p = c.POINTER(c.c_uint8), c.POINTER(c.c_float), c.POINTER(c.c_int16)
        L.vo_num_threads.restype = c.c_int
        L.vo_set_num_threads.argtypes = [c.c_int]
        L.vo_pack_nv12.argtypes = [u8p, c.c_size_t, u8p, c.c_size_t, c.c_int, c.c_int, u8p]
        L.vo_pack_nv12.restype = c.c_int
        L.vo_pack_p010.argtypes = [u8p, c.c_size_t, u8p, c.c_size_t, c.c_int, c.c_int, u8p]
        L.vo_pack_p010.restype = c.c_int
        L.vo_cvt_nv12_bgr.argtypes = [u8p, c.c_int, c.c_int, u8p]
        L.vo_atanf.argtypes = [c.c_float]
        L.vo_atanf.restype = c.c_float
        L.vo_atanf_array.argtypes = [f32p, f32p, c.c_long]
        L.vo_atanf_max_ulp.argtypes = [c.c_uint32, c.c_uint32, c.c_uint32]
        L.vo_atanf_max_ulp.restype = c.c_double
        L.vo_create_map.argtypes = [f32p, f32p, c.c_int, c.c_int, f32p]
        L.vo_remap_bilinear.argtypes = [u8p, c.c_int, c.c_int, c.c_int, f32p, f32p, u8p, c.c_int, c.c_int]
        L.vo_cvt_bgr10_p010.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p]
        L.vo_remap_nearest.argtypes = [u8p, c.c_int, c.c_int, c.c_int, f32p, f32p, u8p, c.c_int, c.c_int]
        L.vo_warp_nv12_reference_path.argtypes = [u8p, c.c_int, c.c_int, f32p, u8p, c.c_int, c.c_int, u8p]
        L.vo_sincosf_array.argtypes = [f32p, f32p, f32p, c.c_long]
        L.vo_create_map_ex.argtypes = [f32p, f32p, c.c_int, c.c_int, f32p, c.c_int]
        L.vo_cvt_bgr_nv12.argtypes = [u8p, c.c_int, c.c_int, u8p, u8p]
        L.vo_warp_nv12_ex.argtypes = [u8p, c.c_int, c.c_int, f32p, c.c_int, c.c_int, u8p, c.c_int, c.c_int, u8p]
        L.vo_create_map_rs.argtypes = [f32p, f32p, c.c_int, c.c_int, f32p, f32p, c.c_int]
        L.vo_warp_p010.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_int, c.c_int, f32p, f32p, c.c_int, c.c_int, c.c_void_p, c.c_int,
                                   c.c_int, u8p]
        L.vo_remap_bilinear10.argtypes = [c.c_void_p, c.c_int, c.c_int, f32p, f32p, c.c_int, c.c_void_p, c.c_int, c.c_int]
        L.vo_cvt_p010_bgr10.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_int, c.c_int, c.c_void_p]
        L.vo_warp_nv12_rs.argtypes = [u8p, c.c_int, c.c_int, f32p, f32p, c.c_int, c.c_int, u8p, c.c_int, c.c_int, u8p]
        L.vo_chroma_maps.argtypes = [f32p, f32p, c.c_int, c.c_int, f32p, f32p]
        L.vo_remap_plane.argtypes = [c.c_void_p, c.c_size_t, c.c_int, c.c_int, c.c_int, c.c_int, f32p, f32p, c.POINTER(c.c_int), c.c_int, c.c_void_p,
                                     c.c_size_t, c.c_int, c.c_int]
        L.vo_warp_planar_mapped.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_int, c.c_int, c.c_int, f32p, f32p, c.c_int, c.c_void_p,
                                            c.c_void_p, c.c_int, c.c_int, f32p]
        L.vo_min_eig.argtypes = [u8p, c.c_size_t, c.c_int, c.c_int, f32p]
        L.vo_good_features.argtypes = [u8p, c.c_size_t, c.c_int, c.c_int, c.c_int, c.c_double, c.c_double, f32p, f32p]
        L.vo_good_features.restype = c.c_int
        L.vo_pyr_down.argtypes = [u8p, c.c_size_t, c.c_int, c.c_int, u8p, c.c_size_t]
        L.vo_scharr.argtypes = [u8p, c.c_size_t, c.c_int, c.c_int, i16p]
        L.vo_pyramid_levels.argtypes = [c.c_int, c.c_int]
        L.vo_pyramid_levels.restype = c.c_int
        L.vo_pyr_lk.argtypes = [u8p, c.c_size_t, u8p, c.c_size_t, c.c_int, c.c_int, f32p, c.c_int, f32p, u8p]
        L.vo_pyr_lk_iterations.argtypes = [u8p, c.c_size_t, u8p, c.c_size_t, c.c_int, c.c_int, f32p, c.c_int, f32p, u8p, c.c_void_p]
        L.vo_pyr_lk.restype = c.c_int
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own createMap.cl compiled for x86-64 (None if not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libcreatemap_ref.so")
        if not os.path.exists(so):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(so):
            return None
        R = ctypes.CDLL(so)
        f32p = ctypes.POINTER(ctypes.c_float)
        R.createmap_ref_run.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, f32p]
        _REF = R
    return _REF


_RUNNER = None
REF_GFX950_CO = os.path.join(_HERE, "_ref", "createMap.gfx950.co")


def ref_gfx950_available():
    """True when the reference kernel's gfx950 code object and its launcher are built (running it needs a GPU)."""
    return os.path.exists(REF_GFX950_CO) and os.path.exists(os.path.join(_HERE, "_build", "libref_cl_runner.so"))


def create_map_ref_gfx950(params, cols, rows, block=(64, 4)):
    """The reference's own createMap.cl, compiled unmodified by ROCm's OpenCL front end for gfx950
    (oracle/_ref/createMap.gfx950.co), launched on the current GPU exactly as FrameSourceWarp.cpp:272-304 does
    (global size {cols, rows}; arguments of :275-300).  -> (map_x, map_y) float32 (rows, cols)."""
    _runner()
    p, pp = _f32(np.asarray(params, np.float32).reshape(17))
    mx = np.empty((rows, cols), np.float32)
    my = np.empty((rows, cols), np.float32)
    err = ctypes.create_string_buffer(512)
    rc = _RUNNER.refcl_create_map(REF_GFX950_CO.encode(), cols, rows, pp, _p(mx, ctypes.c_float), _p(my, ctypes.c_float), int(block[0]),
                                  int(block[1]), err, 512)
    if rc:
        raise RuntimeError("reference createMap (gfx950 code object): " + err.value.decode())
    return mx, my


def _runner():
    global _RUNNER
    if _RUNNER is None:
        build()
        R = ctypes.CDLL(os.path.join(_HERE, "_build", "libref_cl_runner.so"))
        f32p = ctypes.POINTER(ctypes.c_float)
        R.refcl_create_map.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, f32p, f32p, f32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_char_p, ctypes.c_int]
        R.refcl_create_map.restype = ctypes.c_int
        R.refcl_create_map_rs.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, f32p, f32p, f32p, f32p, ctypes.c_char_p, ctypes.c_int]
        R.refcl_create_map_rs.restype = ctypes.c_int
        _RUNNER = R
    if not os.path.exists(REF_GFX950_CO):
        raise RuntimeError("oracle/_ref/createMap.gfx950.co not built")
    return _RUNNER


def create_map_ref_gfx950_rs(params, rot_bottom, cols, rows):
    """The rolling-shutter map in the reference kernel's arithmetic: row y = row y of the reference kernel's output when it is handed
    the definition's matrix m(y) (include/vstab.h: vstab_warp_nv12_rs) as its rotation -- one launch of the code object per row."""
    R = _runner()
    p, pp = _f32(np.asarray(params, np.float32).reshape(17))
    rb, rbp = _f32(np.asarray(rot_bottom, np.float32).reshape(9))
    mx = np.empty((rows, cols), np.float32)
    my = np.empty((rows, cols), np.float32)
    err = ctypes.create_string_buffer(512)
    if R.refcl_create_map_rs(REF_GFX950_CO.encode(), cols, rows, pp, rbp, _p(mx, ctypes.c_float), _p(my, ctypes.c_float), err, 512):
        raise RuntimeError("reference createMap per row (gfx950 code object): " + err.value.decode())
    return mx, my


def warp_nv12_ref_gfx950(nv12, params, dw, dh, rot_bottom=None, nearest=False):
    """cvtColor (oracle) -> createMap (the REFERENCE's kernel, on this GPU) -> cv::remap (oracle): the checker of the
    OpenCL-precision map (the pipeline object's default).  rot_bottom: the rolling-shutter variant (a launch per row)."""
    mx, my = create_map_ref_gfx950(params, dw, dh) if rot_bottom is None else create_map_ref_gfx950_rs(params, rot_bottom, dw, dh)
    bgr = cvt_nv12_bgr(nv12)
    return remap_nearest(bgr, mx, my) if nearest else remap_bilinear(bgr, mx, my)


def remap_bilinear10(bgr10, mapx, mapy, blend=0):
    """The 10-bit remap (vo_remap_pixel10: both blends) with map planes from anywhere.  bgr10: (h, w, 3) uint16, 0..1023."""
    b = np.ascontiguousarray(bgr10, np.uint16)
    h, w = b.shape[:2]
    mx, mxp = _f32(mapx)
    my, myp = _f32(mapy)
    dh, dw = mx.shape
    out = np.empty((dh, dw, 3), np.uint16)
    lib().vo_remap_bilinear10(b.ctypes.data_as(ctypes.c_void_p), w, h, mxp, myp, int(blend), out.ctypes.data_as(ctypes.c_void_p), dw, dh)
    return out


def warp_p010_ref_gfx950(y, uv, params, dw, dh, rot_bottom=None, blend=0):
    """The config-5 chain with the map from the reference's kernel on this GPU: 10-bit conversion (oracle) -> createMap (reference,
    per row when rot_bottom is given) -> 10-bit remap (oracle)."""
    mx, my = create_map_ref_gfx950(params, dw, dh) if rot_bottom is None else create_map_ref_gfx950_rs(params, rot_bottom, dw, dh)
    return remap_bilinear10(cvt_p010_bgr10(y, uv), mx, my, blend)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, _p(a, ctypes.c_uint8)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, _p(a, ctypes.c_float)


# ---------------------------------------------------------------------------------------------
# pixel / feature steps (C)
# ---------------------------------------------------------------------------------------------
def pack_nv12(y, uv):
    """FrameSourceFfmpegOpenCl.cpp:58-85.  y: (h, pitch_y) u8, uv: (h/2, pitch_uv) u8 views."""
    h, w = y.shape
    yy, yp = _u8(y)
    uu, up = _u8(uv)
    dst = np.empty((h * 3 // 2, w), np.uint8)
    rc = lib().vo_pack_nv12(yp, yy.strides[0], up, uu.strides[0], w, h, _p(dst, ctypes.c_uint8))
    if rc:
        raise ValueError("Mismatched image dimensions")
    return dst


def pack_p010(y16, uv16):
    """P010-style planes ((h, >=w) and (h/2, >=w) uint16 views) -> packed 8-bit NV12 by truncation."""
    h, w = y16.shape
    a = np.ascontiguousarray(y16, dtype=np.uint16)
    b = np.ascontiguousarray(uv16, dtype=np.uint16)
    dst = np.empty((h * 3 // 2, w), np.uint8)
    rc = lib().vo_pack_p010(_p(a, ctypes.c_uint8), a.strides[0], _p(b, ctypes.c_uint8), b.strides[0], w, h, _p(dst, ctypes.c_uint8))
    if rc:
        raise ValueError("Mismatched image dimensions")
    return dst


def cvt_nv12_bgr(nv12):
    """FrameSourceWarp.cpp:401.  nv12: (h*3/2, w) u8 -> (h, w, 3) u8 BGR."""
    rows, w = nv12.shape
    h = rows * 2 // 3
    a, ap = _u8(nv12)
    out = np.empty((h, w, 3), np.uint8)
    lib().vo_cvt_nv12_bgr(ap, w, h, _p(out, ctypes.c_uint8))
    return out


def atanf(x):
    x, xp = _f32(x)
    y = np.empty_like(x)
    lib().vo_atanf_array(xp, _p(y, ctypes.c_float), x.size)
    return y


def create_map(params, cols, rows):
    """createMap.cl:1-51 restated.  params: 17 floats in kernel-argument order."""
    p, pp = _f32(params)
    mx = np.empty((rows, cols), np.float32)
    my = np.empty((rows, cols), np.float32)
    lib().vo_create_map(_p(mx, ctypes.c_float), _p(my, ctypes.c_float), cols, rows, pp)
    return mx, my


def create_map_ref(params, cols, rows):
    """The reference's own createMap.cl (oracle/_ref build)."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref not built")
    p, pp = _f32(params)
    mx = np.zeros((rows, cols), np.float32)
    my = np.zeros((rows, cols), np.float32)
    R.createmap_ref_run(_p(mx, ctypes.c_float), _p(my, ctypes.c_float), cols, rows, pp)
    return mx, my


def remap_bilinear(src, mapx, mapy):
    """FrameSourceWarp.cpp:306-312.  src: (h, w, cn) or (h, w) u8."""
    s, sp = _u8(src)
    cn = 1 if s.ndim == 2 else s.shape[2]
    mx, mxp = _f32(mapx)
    my, myp = _f32(mapy)
    dh, dw = mx.shape
    out = np.empty((dh, dw, cn), np.uint8)
    lib().vo_remap_bilinear(sp, s.shape[1], s.shape[0], cn, mxp, myp, _p(out, ctypes.c_uint8), dw, dh)
    return out[:, :, 0] if s.ndim == 2 else out


def remap_nearest(src, mapx, mapy):
    """cv::remap(INTER_NEAREST, BORDER_CONSTANT 0): the other interpolation FrameSourceWarp.hpp:90 admits."""
    s, sp = _u8(src)
    cn = 1 if s.ndim == 2 else s.shape[2]
    mx, mxp = _f32(mapx)
    my, myp = _f32(mapy)
    dh, dw = mx.shape
    out = np.empty((dh, dw, cn), np.uint8)
    lib().vo_remap_nearest(sp, s.shape[1], s.shape[0], cn, mxp, myp, _p(out, ctypes.c_uint8), dw, dh)
    return out[:, :, 0] if s.ndim == 2 else out


def warp_nv12(nv12, params, dw, dh):
    """cvtColor + createMap + remap exactly as FrameSourceWarp.cpp:401,272-314 chains them."""
    rows, w = nv12.shape
    h = rows * 2 // 3
    a, ap = _u8(nv12)
    p, pp = _f32(params)
    out = np.empty((dh, dw, 3), np.uint8)
    work = np.empty(w * h * 3 + 16 + 2 * dw * dh * 4, np.uint8)
    lib().vo_warp_nv12_reference_path(ap, w, h, pp, _p(out, ctypes.c_uint8), dw, dh, _p(work, ctypes.c_uint8))
    return out


def sincosf(t):
    """sin, cos on [0, pi] as the generalised map defines them (vo_sincosf_pos)."""
    t, tp = _f32(t)
    sn, cs = np.empty_like(t), np.empty_like(t)
    lib().vo_sincosf_array(tp, _p(sn, ctypes.c_float), _p(cs, ctypes.c_float), t.size)
    return sn, cs


MAP_CREATEMAP_CL, MAP_FISH_TO_RECT, MAP_FISH_TO_FISH, MAP_RECT_TO_RECT, MAP_RECT_TO_FISH = range(5)


def create_map_ex(params, cols, rows, mode):
    """Generalised map; mode 0 = createMap.cl."""
    p, pp = _f32(params)
    mx = np.empty((rows, cols), np.float32)
    my = np.empty((rows, cols), np.float32)
    lib().vo_create_map_ex(_p(mx, ctypes.c_float), _p(my, ctypes.c_float), cols, rows, pp, int(mode))
    return mx, my


def cvt_bgr_nv12(bgr):
    """OpenCV BGR -> YUV 4:2:0 arithmetic with NV12 chroma layout.  -> (y (h,w), uv (ceil(h/2), ceil(w/2), 2))."""
    b, bp = _u8(bgr)
    h, w = b.shape[:2]
    y = np.empty((h, w), np.uint8)
    uv = np.empty(((h + 1) // 2, (w + 1) // 2, 2), np.uint8)
    lib().vo_cvt_bgr_nv12(bp, w, h, _p(y, ctypes.c_uint8), _p(uv, ctypes.c_uint8))
    return y, uv


def warp_nv12_ex(nv12, params, dw, dh, mode=0, out_format=0):
    """Generalised warp chain.  out_format 0 -> (dh, dw, 3) BGR; 1 -> (y, uv) as cvt_bgr_nv12."""
    rows, w = nv12.shape
    h = rows * 2 // 3
    a, ap = _u8(nv12)
    p, pp = _f32(params)
    cw, ch = (dw + 1) // 2, (dh + 1) // 2
    out = np.empty(dw * dh * 3 if out_format == 0 else dw * dh + 2 * cw * ch, np.uint8)
    work = np.empty(w * h * 3 + 16 + 2 * dw * dh * 4 + dw * dh * 3, np.uint8)
    lib().vo_warp_nv12_ex(ap, w, h, pp, int(mode), int(out_format), _p(out, ctypes.c_uint8), dw, dh, _p(work, ctypes.c_uint8))
    if out_format == 0:
        return out.reshape(dh, dw, 3)
    return out[: dw * dh].reshape(dh, dw), out[dw * dh:].reshape(ch, cw, 2)


def create_map_rs(params, rot_bottom, cols, rows, mode=0):
    """Rolling-shutter map planes (config 5): row y uses the rotation interpolated between params[8:17] and rot_bottom."""
    p, pp = _f32(params)
    rb, rbp = _f32(np.asarray(rot_bottom, np.float32).reshape(9))
    mx, my = np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32)
    lib().vo_create_map_rs(_p(mx, ctypes.c_float), _p(my, ctypes.c_float), cols, rows, pp, rbp, int(mode))
    return mx, my


def warp_nv12_rs(nv12, params, rot_bottom, dw, dh, mode=0, out_format=0):
    """Rolling-shutter warp chain.  out_format 0 -> (dh, dw, 3) BGR; 1 -> (y, uv) as cvt_bgr_nv12."""
    rows, w = nv12.shape
    h = rows * 2 // 3
    a, ap = _u8(nv12)
    p, pp = _f32(params)
    rb, rbp = _f32(np.asarray(rot_bottom, np.float32).reshape(9))
    cw, ch = (dw + 1) // 2, (dh + 1) // 2
    out = np.empty(dw * dh * 3 if out_format == 0 else dw * dh + 2 * cw * ch, np.uint8)
    work = np.empty(w * h * 3 + 16 + 2 * dw * dh * 4 + dw * dh * 3, np.uint8)
    lib().vo_warp_nv12_rs(ap, w, h, pp, rbp, int(mode), int(out_format), _p(out, ctypes.c_uint8), dw, dh, _p(work, ctypes.c_uint8))
    if out_format == 0:
        return out.reshape(dh, dw, 3)
    return out[: dw * dh].reshape(dh, dw), out[dw * dh:].reshape(ch, cw, 2)


def chroma_maps(mapx, mapy):
    """Map planes of the chroma planes in the plane-wise warp: cmap(cx, cy) = map(2 cx, 2 cy) * 0.5f (vo_chroma_maps)."""
    mx, mxp = _f32(mapx)
    my, myp = _f32(mapy)
    dh, dw = mx.shape
    cmx = np.empty(((dh + 1) // 2, (dw + 1) // 2), np.float32)
    cmy = np.empty_like(cmx)
    lib().vo_chroma_maps(mxp, myp, dw, dh, _p(cmx, ctypes.c_float), _p(cmy, ctypes.c_float))
    return cmx, cmy


def warp_planar_mapped(y, uv, mapx, mapy, depth=8, blend=0):
    """The plane-wise warp (SURVEY.md 8(f) row 2 as written: no colour round trip; DEFINED in vo_remap_plane / vo_warp_planar_mapped)
    with luma map planes from anywhere.  depth 8: y (h, w) uint8, uv (h/2, w) uint8 interleaved -> (y' (dh, dw), uv' (ceil(dh/2),
    2 * ceil(dw/2))) uint8.  depth 10: the same shapes in uint16 P010 words."""
    dt = np.uint8 if depth == 8 else np.uint16
    ya, ua = np.ascontiguousarray(y, dt), np.ascontiguousarray(uv, dt)
    h, w = ya.shape
    mx, mxp = _f32(mapx)
    my, myp = _f32(mapy)
    dh, dw = mx.shape
    cw, ch = (dw + 1) // 2, (dh + 1) // 2
    oy, ouv = np.empty((dh, dw), dt), np.empty((ch, 2 * cw), dt)
    work = np.empty(2 * cw * ch, np.float32)
    vp = ctypes.c_void_p
    lib().vo_warp_planar_mapped(ya.ctypes.data_as(vp), ctypes.c_size_t(ya.strides[0]), ua.ctypes.data_as(vp), ctypes.c_size_t(ua.strides[0]), w, h,
                                int(depth), mxp, myp, int(blend), oy.ctypes.data_as(vp), ouv.ctypes.data_as(vp), dw, dh, _p(work, ctypes.c_float))
    return oy, ouv


def _split_nv12(nv12):
    rows, w = nv12.shape
    h = rows * 2 // 3
    return nv12[:h], nv12[h:]


def warp_nv12_planar(nv12, params, dw, dh, mode=0, rot_bottom=None):
    """Plane-wise NV12 -> NV12 warp, map in the CPU oracle's (IEEE) arithmetic: -> (y', uv')."""
    mx, my = create_map_rs(params, rot_bottom, dw, dh, mode) if rot_bottom is not None else create_map_ex(params, dw, dh, mode)
    return warp_planar_mapped(*_split_nv12(np.asarray(nv12)), mx, my)


def warp_nv12_planar_ref_gfx950(nv12, params, dw, dh, rot_bottom=None):
    """The same with the map from the REFERENCE's own kernel on this GPU (the pipeline object's default arithmetic)."""
    mx, my = create_map_ref_gfx950(params, dw, dh) if rot_bottom is None else create_map_ref_gfx950_rs(params, rot_bottom, dw, dh)
    return warp_planar_mapped(*_split_nv12(np.asarray(nv12)), mx, my)


def warp_p010_planar(y, uv, params, dw, dh, mode=0, rot_bottom=None, blend=0):
    """Plane-wise P010 -> P010 warp (config 5's encoder hand-off without a colour round trip), IEEE map."""
    mx, my = create_map_rs(params, rot_bottom, dw, dh, mode) if rot_bottom is not None else create_map_ex(params, dw, dh, mode)
    return warp_planar_mapped(y, uv, mx, my, 10, blend)


def warp_p010_planar_ref_gfx950(y, uv, params, dw, dh, rot_bottom=None, blend=0):
    mx, my = create_map_ref_gfx950(params, dw, dh) if rot_bottom is None else create_map_ref_gfx950_rs(params, rot_bottom, dw, dh)
    return warp_planar_mapped(y, uv, mx, my, 10, blend)


def min_eig(gray):
    g, gp = _u8(gray)
    h, w = g.shape
    e = np.empty((h, w), np.float32)
    lib().vo_min_eig(gp, g.strides[0], w, h, _p(e, ctypes.c_float))
    return e


def good_features(gray, max_corners=200, quality=0.01, min_distance=30.0, return_eig=False):
    """find_corners, FrameSourceWarp.cpp:228-240."""
    g, gp = _u8(gray)
    h, w = g.shape
    cap = max_corners if max_corners > 0 else w * h
    xy = np.zeros((cap, 2), np.float32)
    eig = np.empty((h, w), np.float32) if return_eig else None
    n = lib().vo_good_features(gp, g.strides[0], w, h, max_corners, quality, min_distance,
                               _p(xy, ctypes.c_float), _p(eig, ctypes.c_float) if return_eig else None)
    return (xy[:n].copy(), eig) if return_eig else xy[:n].copy()


def pyr_down(img):
    s, sp = _u8(img)
    h, w = s.shape
    d = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().vo_pyr_down(sp, s.strides[0], w, h, _p(d, ctypes.c_uint8), d.strides[0])
    return d


def scharr(img):
    s, sp = _u8(img)
    h, w = s.shape
    d = np.empty((h, w, 2), np.int16)
    lib().vo_scharr(sp, s.strides[0], w, h, _p(d, ctypes.c_int16))
    return d


def set_lk_accumulation(float_raster_order):
    """False (normative): exact int64 LK sums.  True: fp32 raster-order accumulation, for measuring the gap."""
    lib().vo_set_lk_accumulation(1 if float_raster_order else 0)


def set_lk_final_check(on):
    """True (normative): OpenCV's re-test of the final position behind the iteration loop (level 0).  False: without it."""
    lib().vo_set_lk_final_check(1 if on else 0)


def warp_p010(y, uv, params, dw, dh, rot_bottom=None, mode=0, blend=0):
    """Config 5's 10-bit pixel path (DEFINED in vo_warp_p010).  y: (h, w) uint16 P010 luma, uv: (h/2, w) uint16
    interleaved chroma; -> (dh, dw, 3) uint16 BGR, values 0..1023."""
    ya = np.ascontiguousarray(y, np.uint16)
    ua = np.ascontiguousarray(uv, np.uint16)
    h, w = ya.shape
    p, pp = _f32(params)
    rb = None if rot_bottom is None else np.ascontiguousarray(np.asarray(rot_bottom, np.float32).reshape(9))
    out = np.empty((dh, dw, 3), np.uint16)
    work = np.empty(w * h * 6 + 16 + 2 * dw * dh * 4, np.uint8)
    lib().vo_warp_p010(ya.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ya.strides[0]), ua.ctypes.data_as(ctypes.c_void_p),
                       ctypes.c_size_t(ua.strides[0]), w, h, pp, None if rb is None else _p(rb, ctypes.c_float), int(mode), int(blend),
                       out.ctypes.data_as(ctypes.c_void_p), dw, dh, _p(work, ctypes.c_uint8))
    return out


def cvt_bgr10_p010(bgr):
    """(h, w, 3) uint16 BGR, values 0..1023 -> ((h, w) uint16 P010 luma, (ceil(h/2), 2 * ceil(w/2)) uint16 interleaved chroma)."""
    b = np.ascontiguousarray(bgr, np.uint16)
    h, w = b.shape[:2]
    y = np.empty((h, w), np.uint16)
    uv = np.empty(((h + 1) // 2, 2 * ((w + 1) // 2)), np.uint16)
    lib().vo_cvt_bgr10_p010(b.ctypes.data_as(ctypes.c_void_p), w, h, y.ctypes.data_as(ctypes.c_void_p), uv.ctypes.data_as(ctypes.c_void_p))
    return y, uv


def cvt_p010_bgr10(y, uv):
    ya = np.ascontiguousarray(y, np.uint16)
    ua = np.ascontiguousarray(uv, np.uint16)
    h, w = ya.shape
    out = np.empty((h, w, 3), np.uint16)
    lib().vo_cvt_p010_bgr10(ya.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ya.strides[0]), ua.ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_size_t(ua.strides[0]), w, h, out.ctypes.data_as(ctypes.c_void_p))
    return out


def pyr_lk_iterations(prev, nxt, pts):
    """pyr_lk plus the Gauss-Newton iterations every feature ran on each pyramid level: -> (next, status, (n, 4) ints)."""
    a, ap = _u8(prev)
    b, bp = _u8(nxt)
    h, w = a.shape
    p, pp = _f32(np.asarray(pts, np.float32).reshape(-1, 2))
    n = p.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    it = np.zeros((n, 4), np.int32)
    lib().vo_pyr_lk_iterations(ap, a.strides[0], bp, b.strides[0], w, h, pp, n, _p(out, ctypes.c_float), _p(st, ctypes.c_uint8),
                               it.ctypes.data_as(ctypes.c_void_p))
    return out, st, it


def pyr_lk(prev, nxt, pts):
    """find_point_pairs_with_optical_flow's calcOpticalFlowPyrLK, FrameSourceWarp.cpp:252."""
    a, ap = _u8(prev)
    b, bp = _u8(nxt)
    h, w = a.shape
    p, pp = _f32(np.asarray(pts, np.float32).reshape(-1, 2))
    n = p.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    lib().vo_pyr_lk(ap, a.strides[0], bp, b.strides[0], w, h, pp, n, _p(out, ctypes.c_float), _p(st, ctypes.c_uint8))
    return out, st


from .geometry import *  # noqa: E402,F401,F403
