"""ctypes binding of libvstab.so (the C ABI in include/vstab.h) for tests, bench.py and smoke().

This is a thin test/bench driver, not the product: the product is the HIP/C++ library.  There
is NO fallback: if lib/libvstab.so is missing or fails to load, importing this module raises.
Device memory and streams come from PyTorch-ROCm (plumbing only): functions take CUDA uint8 /
float32 tensors and enqueue on torch's current stream.

Import with ``importlib.import_module("video-annotator_amd")`` (the directory name is not a
Python identifier).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Always the in-tree product library.  (Development tools that want another build load this module by hand and plant a
# path in its namespace first -- tools/devlib.py; no environment variable can swap the library under the tests.)
LIB_PATH = globals().get("_VSTAB_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libvstab.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
        "there is no CPU fallback for the HIP path")

# PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the one already mapped when libvstab.so
# resolves the same SONAME, or the process ends up with two HIP runtimes and no visible device.
import torch  # noqa: E402,F401  (plumbing: device memory + streams)

_L = ctypes.CDLL(LIB_PATH)

OK, EOF, ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_SOURCE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

(GOPRO_H4B_WIDE43_PUBLISHED, GOPRO_H4B_WIDE43_MEASURED, GOPRO_H4B_WIDE43_MEASURED_STABILISATION,
 GOPRO_H4B_WIDE169_PUBLISHED, GOPRO_H4B_WIDE169_MEASURED, GOPRO_H4B_WIDE169_MEASURED_STABILISATION) = range(6)

_c = ctypes
_vp, _sz, _i, _d = _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_double
_dp, _fp, _ip = _c.POINTER(_c.c_double), _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)

_u8p, _i64, _u64 = _c.POINTER(_c.c_ubyte), _c.c_int64, _c.c_uint64


class Frame(_c.Structure):  # vstab_frame
    _fields_ = [("y", _vp), ("uv", _vp), ("pitch_y", _sz), ("pitch_uv", _sz), ("width", _i), ("height", _i),
                ("mem", _i), ("pts", _i64), ("delta_rotation", _dp), ("bit_depth", _i), ("hold", _i),
                ("readout_rotation", _dp), ("dmabuf_fd", _i), ("dmabuf_size", _sz), ("dmabuf_modifier", _u64)]


PULL_FN = _c.CFUNCTYPE(_i, _vp, _c.POINTER(Frame))


class Source(_c.Structure):  # vstab_source
    _fields_ = [("pull", PULL_FN), ("peek", PULL_FN), ("user", _vp)]


class Config(_c.Structure):  # vstab_config
    _fields_ = [("abi_version", _i), ("preset", _i), ("scale", _d), ("crop_borders", _i), ("zoom", _d), ("smooth_radius", _i),
                ("interpolation", _i), ("smoother", _i), ("tracking", _i), ("seed", _u64), ("stream", _vp),
                ("lens_mode", _i), ("in_projection", _i), ("out_projection", _i), ("in_dfov", _d), ("out_dfov", _d),
                ("out_width", _i), ("out_height", _i), ("out_cx", _d), ("out_cy", _d), ("debug", _i), ("pixel_depth", _i),
                ("blend", _i), ("read_ahead", _i), ("map_precision", _i)]


class FrameLog(_c.Structure):  # vstab_frame_log
    _fields_ = [("key_frame", _i), ("n_corners", _i), ("n_tracked", _i), ("n_inliers", _i), ("fallback", _i),
                ("R_frame", _d * 9), ("R_accum", _d * 9)]


class Profile(_c.Structure):  # vstab_profile
    _fields_ = [("frames_consumed", _c.c_long), ("frames_emitted", _c.c_long), ("key_frames", _c.c_long),
                ("gpu_ingest_ms", _d), ("gpu_pyramid_ms", _d), ("gpu_corners_ms", _d), ("gpu_lk_ms", _d), ("gpu_warp_ms", _d),
                ("host_corners_ms", _d), ("host_track_wait_ms", _d), ("host_estimate_ms", _d), ("host_smooth_ms", _d),
                ("warp_launches", _c.c_long), ("warp_timed", _c.c_long),
                ("dmabuf_imports", _c.c_long), ("dmabuf_evictions", _c.c_long), ("dmabuf_cached", _c.c_long),
                ("corner_selections_by_caller", _c.c_long), ("corner_selections_by_helper", _c.c_long), ("epochs_in_turn", _c.c_long)]


SMOOTHER_SG, SMOOTHER_KALMAN, SMOOTHER_NONE, SMOOTHER_FIXED = 0, 1, 2, 3
PROJ_RECT, PROJ_FISH = 0, 1
MAP_CREATEMAP_CL, MAP_FISH_TO_RECT, MAP_FISH_TO_FISH, MAP_RECT_TO_RECT, MAP_RECT_TO_FISH, MAP_CREATEMAP_CL_OPENCL = range(6)
OUT_BGR8, OUT_NV12, OUT_NV12_PLANAR = 0, 1, 2
MAP_PRECISION_IEEE, MAP_PRECISION_OPENCL = 0, 1
_pp = _c.POINTER(_vp)

# name -> (restype, argtypes); mirrors include/vstab.h one to one
SIGNATURES = {
    "vstab_last_error": (_c.c_char_p, []),
    "vstab_version": (_c.c_char_p, []),
    "vstab_device_count": (_i, []),
    "vstab_struct_size": (_i, [_i]),
    "vstab_abi_version": (_i, []),
    "vstab_get_preset_camera": (_i, [_i, _i, _i, _dp]),
    "vstab_get_output_camera": (_i, [_dp, _i, _i, _d, _i, _d, _dp, _ip, _ip]),
    "vstab_fisheye_undistort_points": (_i, [_dp, _i, _dp, _dp, _dp, _dp]),
    "vstab_map_params": (None, [_dp, _dp, _dp, _fp]),
    "vstab_pack_nv12": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _vp]),
    "vstab_pack_p010": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _vp]),
    "vstab_cvt_nv12_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _sz, _vp]),
    "vstab_create_map": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _vp]),
    "vstab_remap_bilinear": (_i, [_vp, _sz, _i, _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_nv12_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_nv12_nearest": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_nv12_nearest_ex": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _i, _vp, _sz, _i, _i, _vp]),
    "vstab_create_map_ex": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _i, _vp]),
    "vstab_warp_nv12_ex": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _i, _i, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_nv12_rs": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_quantised_map_bytes": (_sz, [_i, _i]),
    "vstab_quantised_map": (_i, [_vp, _i, _i, _fp, _i, _vp]),
    "vstab_warp_nv12_mapped": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _i, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_draw_markers": (_i, [_vp, _sz, _i, _i, _i, _vp, _i, _i, _c.c_uint, _vp]),
    "vstab_pyr_down": (_i, [_vp, _sz, _i, _i, _vp, _sz, _vp]),
    "vstab_pyr_down_x2": (_i, [_vp, _sz, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "vstab_min_eig": (_i, [_vp, _sz, _i, _i, _vp, _vp]),
    "vstab_good_features": (_i, [_vp, _sz, _i, _i, _i, _d, _d, _fp, _ip, _vp]),
    "vstab_pull_frame_bgr16": (_i, [_vp, _vp, _sz]),
    "vstab_warp_p010": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _i, _i, _vp]),
    "vstab_good_features_ex": (_i, [_vp, _sz, _i, _i, _i, _d, _d, _i, _fp, _ip, _ip, _vp]),
    "vstab_pyr_lk": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _i, _fp, _u8p, _vp]),
    "vstab_estimate_rotation": (_i, [_fp, _fp, _i, _dp, _dp, _u64, _dp, _ip]),
    "vstab_time_next_launch": (None, [_vp, _vp]),
    "vstab_sg_weights": (_i, [_i, _dp]),
    "vstab_gyro_integrate": (_i, [_vp, _i, _d, _d, _d, _d, _dp, _dp]),
    "vstab_gpmf_parse_gyro": (_i, [_vp, _sz, _d, _d, _vp, _i, _ip]),
    "vstab_preload_kernels": (_i, []),
    "vstab_rotation_filter_create": (_i, [_i, _pp]),
    "vstab_rotation_filter_add": (_i, [_vp, _dp]),
    "vstab_rotation_filter_filter": (_i, [_vp, _dp]),
    "vstab_rotation_filter_destroy": (None, [_vp]),
    "vstab_config_default": (None, [_c.POINTER(Config)]),
    "vstab_create": (_i, [_c.POINTER(Config), _c.POINTER(Source), _pp]),
    "vstab_get_output_info": (_i, [_vp, _ip, _ip, _dp, _dp]),
    "vstab_pull_frame": (_i, [_vp, _vp, _sz]),
    "vstab_warp_p010_planes": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_p010_planar": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _fp, _i, _i, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_cvt_bgr16_p010": (_i, [_vp, _sz, _i, _i, _vp, _sz, _vp, _sz, _vp]),
    "vstab_pull_frame_p010": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vstab_pull_frames": (_i, [_vp, _i, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t), _i, _i, _c.POINTER(_c.c_int)]),
    "vstab_pull_frame_nv12": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vstab_pull_frame_nv12_planar": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vstab_pull_frame_p010_planar": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vstab_pull_frame_host": (_i, [_vp, _vp, _sz]),
    "vstab_lens_camera": (_i, [_i, _d, _i, _i, _d, _d, _dp]),
    "vstab_peek_frame": (_i, [_vp, _vp, _sz]),
    "vstab_destroy": (None, [_vp]),
    "vstab_frame_log_count": (_i, [_vp]),
    "vstab_get_frame_log": (_i, [_vp, _i, _c.POINTER(FrameLog)]),
    "vstab_get_warp_rotation": (_i, [_vp, _i, _dp]),
    "vstab_enable_profiling": (_i, [_vp, _i]),
    "vstab_get_profile": (_i, [_vp, _c.POINTER(Profile)]),
    "vstab_ring_source_create": (_i, [_pp, _i, _i, _i, _sz, _c.c_long, _pp, _c.POINTER(Source)]),
    "vstab_ring_source_create_ex": (_i, [_c.POINTER(_vp), _i, _i, _i, _sz, _c.c_long, _i, _dp, _c.POINTER(_vp), _c.POINTER(Source)]),
    "vstab_ring_source_set_hold": (None, [_vp, _i]),
    "vstab_ring_source_destroy": (None, [_vp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(_L, _name)  # AttributeError here = header/library mismatch: fail loudly
    _f.restype, _f.argtypes = _res, _args

lib = _L
ABI_VERSION = 0x56534205  # include/vstab.h: VSTAB_ABI_VERSION ("VSB" + layout version 5)
if _L.vstab_abi_version() != ABI_VERSION:
    raise ImportError(f"video-annotator_amd: this binding mirrors ABI version {ABI_VERSION}, libvstab.so is version {_L.vstab_abi_version()}")
for _k, _t in enumerate((Frame, Source, Config, FrameLog, Profile)):  # the ctypes mirrors must match the compiled structs
    if _L.vstab_struct_size(_k) != _c.sizeof(_t):
        raise ImportError(f"video-annotator_amd: ctypes mirror of {_t.__name__} is {_c.sizeof(_t)} bytes, libvstab.so has {_L.vstab_struct_size(_k)}")


class VstabError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__(f"{where}: status {status}: {_L.vstab_last_error().decode()}")


def _check(st, where):
    if st != OK:
        raise VstabError(st, where)


def _stream():
    import torch
    return _vp(torch.cuda.current_stream().cuda_stream)


def _dptr(a):
    return a.ctypes.data_as(_dp)


def _fptr(a):
    return a.ctypes.data_as(_fp)


def version():
    return _L.vstab_version().decode()


def device_count():
    return _L.vstab_device_count()


# ---------------------------------------------------------------------------------------------
# cameras (host)
# ---------------------------------------------------------------------------------------------
def get_preset_camera(preset, width, height):
    K = np.zeros(9)
    _check(_L.vstab_get_preset_camera(preset, width, height, _dptr(K)), "vstab_get_preset_camera")
    return K.reshape(3, 3)


def get_output_camera(K_in, width, height, scale=1.0, crop_borders=False, zoom=1.0):
    Ki = np.ascontiguousarray(K_in, np.float64).reshape(9)
    Ko = np.zeros(9)
    ow, oh = _c.c_int(), _c.c_int()
    _check(_L.vstab_get_output_camera(_dptr(Ki), width, height, scale, int(crop_borders), zoom, _dptr(Ko),
                                      _c.byref(ow), _c.byref(oh)), "vstab_get_output_camera")
    return Ko.reshape(3, 3), (ow.value, oh.value)


def lens_camera(projection, dfov_deg, width, height, cx=-1.0, cy=-1.0):
    K = np.zeros(9)
    _check(_L.vstab_lens_camera(int(projection), float(dfov_deg), width, height, cx, cy, _dptr(K)), "vstab_lens_camera")
    return K.reshape(3, 3)


def fisheye_undistort_points(pts, K, R=None, P=None):
    p = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    Kc = np.ascontiguousarray(K, np.float64).reshape(9)
    Rc = None if R is None else np.ascontiguousarray(R, np.float64).reshape(9)
    Pc = None if P is None else np.ascontiguousarray(P, np.float64).reshape(9)
    out = np.zeros_like(p)
    _check(_L.vstab_fisheye_undistort_points(_dptr(p), p.shape[0], _dptr(Kc), None if Rc is None else _dptr(Rc),
                                             None if Pc is None else _dptr(Pc), _dptr(out)),
           "vstab_fisheye_undistort_points")
    return out


def map_params(K_in, K_out, R):
    a = np.ascontiguousarray(K_in, np.float64).reshape(9)
    b = np.ascontiguousarray(K_out, np.float64).reshape(9)
    c = np.ascontiguousarray(R, np.float64).reshape(9)
    p = np.zeros(17, np.float32)
    _L.vstab_map_params(_dptr(a), _dptr(b), _dptr(c), _fptr(p))
    return p


# ---------------------------------------------------------------------------------------------
# stateless device operators (torch CUDA tensors in / out)
# ---------------------------------------------------------------------------------------------
def _planes(nv12):
    """(h*3/2, w) packed NV12 tensor (any row stride) -> y ptr, uv ptr, pitch, w, h."""
    rows, w = nv12.shape
    assert nv12.stride(1) == 1
    h = rows * 2 // 3
    pitch = nv12.stride(0)
    return nv12.data_ptr(), nv12.data_ptr() + h * pitch, pitch, w, h


def pack_nv12(y, uv):
    """y: (h, >=w) view with row stride = pitch; uv: (h/2, >=w) view."""
    import torch
    h, w = y.shape
    dst = torch.empty((h * 3 // 2, w), dtype=torch.uint8, device=y.device)
    _check(_L.vstab_pack_nv12(y.data_ptr(), y.stride(0), uv.data_ptr(), uv.stride(0), w, h, dst.data_ptr(),
                              _stream()), "vstab_pack_nv12")
    return dst


def pack_p010(y16, uv16):
    """y16: (h, >=w) uint16/int16 view of the luma plane, uv16: (h/2, >=w) view of the interleaved chroma plane
    (P010: significant bits at the top) -> packed 8-bit NV12 (h*3/2, w)."""
    import torch
    h, w = y16.shape
    dst = torch.empty((h * 3 // 2, w), dtype=torch.uint8, device=y16.device)
    _check(_L.vstab_pack_p010(y16.data_ptr(), y16.stride(0) * 2, uv16.data_ptr(), uv16.stride(0) * 2, w, h, dst.data_ptr(),
                              _stream()), "vstab_pack_p010")
    return dst


def cvt_nv12_bgr(nv12, out=None):
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    if out is None:
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=nv12.device)
    _check(_L.vstab_cvt_nv12_bgr(yp, pitch, uvp, pitch, w, h, out.data_ptr(), out.stride(0), _stream()),
           "vstab_cvt_nv12_bgr")
    return out


def create_map(params, cols, rows, device="cuda", mode=MAP_CREATEMAP_CL):
    import torch
    p = np.ascontiguousarray(params, np.float32)
    mx = torch.empty((rows, cols), dtype=torch.float32, device=device)
    my = torch.empty((rows, cols), dtype=torch.float32, device=device)
    if mode == MAP_CREATEMAP_CL:
        _check(_L.vstab_create_map(mx.data_ptr(), mx.stride(0) * 4, my.data_ptr(), my.stride(0) * 4, cols, rows,
                                   _fptr(p), _stream()), "vstab_create_map")
    else:
        _check(_L.vstab_create_map_ex(mx.data_ptr(), mx.stride(0) * 4, my.data_ptr(), my.stride(0) * 4, cols, rows,
                                      _fptr(p), int(mode), _stream()), "vstab_create_map_ex")
    return mx, my


def remap_bilinear(src, mapx, mapy):
    import torch
    cn = 1 if src.dim() == 2 else src.shape[2]
    sh, sw = src.shape[0], src.shape[1]
    dh, dw = mapx.shape
    shape = (dh, dw) if src.dim() == 2 else (dh, dw, cn)
    out = torch.empty(shape, dtype=torch.uint8, device=src.device)
    _check(_L.vstab_remap_bilinear(src.data_ptr(), src.stride(0), sw, sh, cn, mapx.data_ptr(), mapx.stride(0) * 4,
                                   mapy.data_ptr(), mapy.stride(0) * 4, out.data_ptr(), out.stride(0), dw, dh,
                                   _stream()), "vstab_remap_bilinear")
    return out


def warp_nv12_bgr(nv12, params, dw, dh, out=None):
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    p = np.ascontiguousarray(params, np.float32)
    if out is None:
        out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=nv12.device)
    _check(_L.vstab_warp_nv12_bgr(yp, pitch, uvp, pitch, w, h, _fptr(p), out.data_ptr(), out.stride(0), dw, dh,
                                  _stream()), "vstab_warp_nv12_bgr")
    return out


def warp_nv12_nearest(nv12, params, dw, dh, out=None, mode=None):
    """vstab_warp_nv12_nearest(_ex): the fused warp with cv::remap's INTER_NEAREST."""
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    p = np.ascontiguousarray(params, np.float32)
    if out is None:
        out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=nv12.device)
    if mode is None:
        _check(_L.vstab_warp_nv12_nearest(yp, pitch, uvp, pitch, w, h, _fptr(p), out.data_ptr(), out.stride(0), dw, dh, _stream()), "vstab_warp_nv12_nearest")
    else:
        _check(_L.vstab_warp_nv12_nearest_ex(yp, pitch, uvp, pitch, w, h, _fptr(p), int(mode), out.data_ptr(), out.stride(0), dw, dh, _stream()),
               "vstab_warp_nv12_nearest_ex")
    return out


def nv12_out_planes(dw, dh, device="cuda"):
    """Output planes for OUT_NV12: luma (dh, dw) and interleaved chroma (ceil(dh/2), 2*ceil(dw/2))."""
    import torch
    return (torch.empty((dh, dw), dtype=torch.uint8, device=device),
            torch.empty(((dh + 1) // 2, 2 * ((dw + 1) // 2)), dtype=torch.uint8, device=device))


def warp_nv12(nv12, params, dw, dh, mode=MAP_CREATEMAP_CL, out_format=OUT_BGR8, out=None):
    """vstab_warp_nv12_ex.  OUT_BGR8 -> (dh, dw, 3) tensor; OUT_NV12 -> (luma, chroma) tensors."""
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    p = np.ascontiguousarray(params, np.float32)
    if out_format == OUT_BGR8:
        if out is None:
            out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=nv12.device)
        _check(_L.vstab_warp_nv12_ex(yp, pitch, uvp, pitch, w, h, _fptr(p), int(mode), OUT_BGR8, out.data_ptr(),
                                     out.stride(0), None, 0, dw, dh, _stream()), "vstab_warp_nv12_ex")
        return out
    if out is None:
        out = nv12_out_planes(dw, dh, nv12.device)
    yo, co = out
    _check(_L.vstab_warp_nv12_ex(yp, pitch, uvp, pitch, w, h, _fptr(p), int(mode), int(out_format), yo.data_ptr(), yo.stride(0),
                                 co.data_ptr(), co.stride(0), dw, dh, _stream()), "vstab_warp_nv12_ex")
    return yo, co


def warp_nv12_rs(nv12, params, rot_bottom, dw, dh, mode=MAP_CREATEMAP_CL, out_format=OUT_BGR8, out=None):
    """vstab_warp_nv12_rs: the warp with a rotation per output row (first row params[8:17], last row rot_bottom)."""
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    p = np.ascontiguousarray(params, np.float32)
    rb = np.ascontiguousarray(rot_bottom, np.float32).reshape(9)
    if out_format == OUT_BGR8:
        if out is None:
            out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=nv12.device)
        _check(_L.vstab_warp_nv12_rs(yp, pitch, uvp, pitch, w, h, _fptr(p), _fptr(rb), int(mode), OUT_BGR8, out.data_ptr(), out.stride(0), None, 0,
                                     dw, dh, _stream()), "vstab_warp_nv12_rs")
        return out
    if out is None:
        out = nv12_out_planes(dw, dh, nv12.device)
    yo, co = out
    _check(_L.vstab_warp_nv12_rs(yp, pitch, uvp, pitch, w, h, _fptr(p), _fptr(rb), int(mode), int(out_format), yo.data_ptr(), yo.stride(0),
                                 co.data_ptr(), co.stride(0), dw, dh, _stream()), "vstab_warp_nv12_rs")
    return yo, co


BLEND_EXACT, BLEND_FP16 = 0, 1


def warp_p010(y, uv, params, dw, dh, rot_bottom=None, mode=MAP_CREATEMAP_CL, blend=BLEND_EXACT, out=None):
    """vstab_warp_p010 (config 5): y (h, w) / uv (h/2, w) int16-or-uint16 CUDA tensors holding P010 samples ->
    (dh, dw, 3) int16 tensor of BGR values 0..1023."""
    import torch
    h, w = y.shape
    p = np.ascontiguousarray(params, np.float32)
    rb = None if rot_bottom is None else np.ascontiguousarray(rot_bottom, np.float32).reshape(9)
    if out is None:
        out = torch.empty((dh, dw, 3), dtype=torch.int16, device=y.device)
    _check(_L.vstab_warp_p010(y.data_ptr(), y.stride(0) * 2, uv.data_ptr(), uv.stride(0) * 2, w, h, _fptr(p), None if rb is None else _fptr(rb),
                              int(mode), int(blend), out.data_ptr(), out.stride(0) * 2, dw, dh, _stream()), "vstab_warp_p010")
    return out


def warp_p010_planes(y, uv, params, dw, dh, rot_bottom=None, mode=MAP_CREATEMAP_CL, blend=BLEND_EXACT, out_y=None, out_uv=None):
    """vstab_warp_p010_planes: the 10-bit warp with P010 planes out, one kernel.  Raises VstabError (ERR_UNSUPPORTED) for planes the
    tiled kernel cannot take."""
    import torch
    h, w = y.shape
    p = np.ascontiguousarray(params, np.float32)
    rb = None if rot_bottom is None else np.ascontiguousarray(rot_bottom, np.float32).reshape(9)
    if out_y is None:
        out_y = torch.empty((dh, dw), dtype=torch.int16, device=y.device)
    if out_uv is None:
        out_uv = torch.empty(((dh + 1) // 2, 2 * ((dw + 1) // 2)), dtype=torch.int16, device=y.device)
    _check(_L.vstab_warp_p010_planes(y.data_ptr(), y.stride(0) * 2, uv.data_ptr(), uv.stride(0) * 2, w, h, _fptr(p), None if rb is None else _fptr(rb),
                                     int(mode), int(blend), out_y.data_ptr(), out_y.stride(0) * 2, out_uv.data_ptr(), out_uv.stride(0) * 2, dw, dh, _stream()),
           "vstab_warp_p010_planes")
    return out_y, out_uv


def warp_p010_planar(y, uv, params, dw, dh, rot_bottom=None, mode=MAP_CREATEMAP_CL, blend=BLEND_EXACT, out_y=None, out_uv=None):
    """vstab_warp_p010_planar: the plane-wise 10-bit warp, P010 planes in and out, no colour round trip."""
    import torch
    h, w = y.shape
    p = np.ascontiguousarray(params, np.float32)
    rb = None if rot_bottom is None else np.ascontiguousarray(rot_bottom, np.float32).reshape(9)
    if out_y is None:
        out_y = torch.empty((dh, dw), dtype=torch.int16, device=y.device)
    if out_uv is None:
        out_uv = torch.empty(((dh + 1) // 2, 2 * ((dw + 1) // 2)), dtype=torch.int16, device=y.device)
    _check(_L.vstab_warp_p010_planar(y.data_ptr(), y.stride(0) * 2, uv.data_ptr(), uv.stride(0) * 2, w, h, _fptr(p), None if rb is None else _fptr(rb),
                                     int(mode), int(blend), out_y.data_ptr(), out_y.stride(0) * 2, out_uv.data_ptr(), out_uv.stride(0) * 2, dw, dh, _stream()),
           "vstab_warp_p010_planar")
    return out_y, out_uv


def cvt_bgr16_p010(bgr16, out_y=None, out_uv=None):
    """vstab_cvt_bgr16_p010: (h, w, 3) int16 CUDA tensor of BGR values 0..1023 -> P010 planes (h, w) and (ceil(h/2), 2 * ceil(w/2)), int16 bit patterns."""
    import torch
    h, w = bgr16.shape[:2]
    if out_y is None:
        out_y = torch.empty((h, w), dtype=torch.int16, device=bgr16.device)
    if out_uv is None:
        out_uv = torch.empty(((h + 1) // 2, 2 * ((w + 1) // 2)), dtype=torch.int16, device=bgr16.device)
    _check(_L.vstab_cvt_bgr16_p010(bgr16.data_ptr(), bgr16.stride(0) * 2, w, h, out_y.data_ptr(), out_y.stride(0) * 2, out_uv.data_ptr(), out_uv.stride(0) * 2,
                                   _stream()), "vstab_cvt_bgr16_p010")
    return out_y, out_uv


def time_next_launch(start_event, stop_event):
    """The next stateless warp call stamps its kernel's own start / end into the two torch.cuda.Event(enable_timing=True):
    kernel-only time, as rocprofv3's kernel trace reports it."""
    for e in (start_event, stop_event):
        if not e.cuda_event:   # torch creates the hipEvent_t lazily, at the first record
            e.record()
    _L.vstab_time_next_launch(_c.c_void_p(start_event.cuda_event), _c.c_void_p(stop_event.cuda_event))


def quantised_map(params, dw, dh, mode=MAP_CREATEMAP_CL, device="cuda"):
    """The per-pixel quantised map for a run of frames with the same warp parameters -> opaque device tensor."""
    import torch
    p = np.ascontiguousarray(params, np.float32)
    q = torch.empty(_L.vstab_quantised_map_bytes(dw, dh), dtype=torch.uint8, device=device)
    _check(_L.vstab_quantised_map(q.data_ptr(), dw, dh, _fptr(p), int(mode), _stream()), "vstab_quantised_map")
    return q


def warp_nv12_mapped(nv12, qmap, dw, dh, out_format=OUT_BGR8, out=None):
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    if out_format == OUT_BGR8:
        if out is None:
            out = torch.empty((dh, dw, 3), dtype=torch.uint8, device=nv12.device)
        _check(_L.vstab_warp_nv12_mapped(yp, pitch, uvp, pitch, w, h, qmap.data_ptr(), OUT_BGR8, out.data_ptr(), out.stride(0), None, 0,
                                         dw, dh, _stream()), "vstab_warp_nv12_mapped")
        return out
    if out is None:
        out = nv12_out_planes(dw, dh, nv12.device)
    yo, co = out
    _check(_L.vstab_warp_nv12_mapped(yp, pitch, uvp, pitch, w, h, qmap.data_ptr(), int(out_format), yo.data_ptr(), yo.stride(0),
                                     co.data_ptr(), co.stride(0), dw, dh, _stream()), "vstab_warp_nv12_mapped")
    return yo, co


# ---------------------------------------------------------------------------------------------
# tracking front-end / motion model
# ---------------------------------------------------------------------------------------------
def pyr_down(img):
    import torch
    h, w = img.shape
    out = torch.empty(((h + 1) // 2, (w + 1) // 2), dtype=torch.uint8, device=img.device)
    _check(_L.vstab_pyr_down(img.data_ptr(), img.stride(0), w, h, out.data_ptr(), out.stride(0), _stream()), "vstab_pyr_down")
    return out


def pyr_down_x2(img):
    """vstab_pyr_down_x2: two pyramid levels in one launch -> (mid, dst)."""
    import torch
    h, w = img.shape
    mw, mh = (w + 1) // 2, (h + 1) // 2
    mid = torch.empty((mh, mw), dtype=torch.uint8, device=img.device)
    dst = torch.empty(((mh + 1) // 2, (mw + 1) // 2), dtype=torch.uint8, device=img.device)
    _check(_L.vstab_pyr_down_x2(img.data_ptr(), img.stride(0), w, h, mid.data_ptr(), mid.stride(0), dst.data_ptr(), dst.stride(0), _stream()), "vstab_pyr_down_x2")
    return mid, dst


def min_eig(gray):
    import torch
    h, w = gray.shape
    out = torch.empty((h, w), dtype=torch.float32, device=gray.device)
    _check(_L.vstab_min_eig(gray.data_ptr(), gray.stride(0), w, h, out.data_ptr(), _stream()), "vstab_min_eig")
    return out


DETECTOR_AUTO, DETECTOR_TWO_PASS, DETECTOR_FUSED = 0, 1, 2


def good_features(gray, max_corners=200, quality=0.01, min_distance=30.0, detector=None, info=None):
    """detector: None -> vstab_good_features; DETECTOR_AUTO / DETECTOR_TWO_PASS -> vstab_good_features_ex, and
    info["detector_used"] (if a dict is given) says which kernel path produced the corners."""
    h, w = gray.shape
    xy = np.zeros((max_corners, 2), np.float32)
    n = _c.c_int()
    if detector is None:
        _check(_L.vstab_good_features(gray.data_ptr(), gray.stride(0), w, h, max_corners, quality, min_distance, _fptr(xy),
                                      _c.byref(n), _stream()), "vstab_good_features")
    else:
        used = _c.c_int()
        _check(_L.vstab_good_features_ex(gray.data_ptr(), gray.stride(0), w, h, max_corners, quality, min_distance, detector, _fptr(xy),
                                         _c.byref(n), _c.byref(used), _stream()), "vstab_good_features_ex")
        if info is not None:
            info["detector_used"] = used.value
    return xy[:n.value].copy()


def pyr_lk(prev, nxt, pts):
    h, w = prev.shape
    p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = p.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    _check(_L.vstab_pyr_lk(prev.data_ptr(), prev.stride(0), nxt.data_ptr(), nxt.stride(0), w, h, _fptr(p), n, _fptr(out),
                           st.ctypes.data_as(_u8p), _stream()), "vstab_pyr_lk")
    return out, st


def estimate_rotation(prev_xy, cur_xy, K_in, K_out, seed=1):
    a = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(cur_xy, np.float32).reshape(-1, 2)
    Ki = np.ascontiguousarray(K_in, np.float64).reshape(9)
    Ko = np.ascontiguousarray(K_out, np.float64).reshape(9)
    R = np.zeros(9)
    inl = _c.c_int()
    _check(_L.vstab_estimate_rotation(_fptr(a), _fptr(b), a.shape[0], _dptr(Ki), _dptr(Ko), seed, _dptr(R), _c.byref(inl)),
           "vstab_estimate_rotation")
    return R.reshape(3, 3), inl.value


def gyro_integrate(samples, rate_scale, t_prev_first_row, t_first_row, t_last_row):
    """vstab_gyro_integrate.  samples: (n, 5) doubles {start_ts, end_ts, roll, pitch, yaw} (the reference's GyroFrame,
    gpmf.cpp:5-11) -> (R_delta, R_readout) 3x3."""
    a = np.ascontiguousarray(samples, np.float64).reshape(-1, 5)
    Rd, Rr = np.zeros(9), np.zeros(9)
    _check(_L.vstab_gyro_integrate(a.ctypes.data_as(_vp), a.shape[0], float(rate_scale), float(t_prev_first_row), float(t_first_row),
                                   float(t_last_row), Rd.ctypes.data_as(_dp), Rr.ctypes.data_as(_dp)), "vstab_gyro_integrate")
    return Rd.reshape(3, 3), Rr.reshape(3, 3)


def preload_kernels():
    """vstab_preload_kernels: load the library's GPU code objects now (vstab_create does it itself)."""
    _check(_L.vstab_preload_kernels(), "vstab_preload_kernels")


def gpmf_parse_gyro(payload, pkt_ts, pkt_dur, cap=None):
    """vstab_gpmf_parse_gyro: bytes of one GPMF packet -> (n, 5) array of {start_ts, end_ts, roll, pitch, yaw} samples."""
    buf = bytes(payload)
    n = _c.c_int(0)
    cap = (len(buf) // 2 + 1) if cap is None else int(cap)
    out = np.zeros((max(cap, 1), 5), np.float64)
    _check(_L.vstab_gpmf_parse_gyro(buf, len(buf), float(pkt_ts), float(pkt_dur), out.ctypes.data, cap, _c.byref(n)), "vstab_gpmf_parse_gyro")
    return out[: min(n.value, cap)].copy(), n.value


def sg_weights(m):
    w = np.zeros(2 * m + 1)
    _check(_L.vstab_sg_weights(m, _dptr(w)), "vstab_sg_weights")
    return w


class RotationFilter:
    def __init__(self, m):
        self._h = _vp()
        _check(_L.vstab_rotation_filter_create(m, _c.byref(self._h)), "vstab_rotation_filter_create")

    def add(self, R):
        r = np.ascontiguousarray(R, np.float64).reshape(9)
        _check(_L.vstab_rotation_filter_add(self._h, _dptr(r)), "vstab_rotation_filter_add")

    def filter(self):
        out = np.zeros(9)
        _check(_L.vstab_rotation_filter_filter(self._h, _dptr(out)), "vstab_rotation_filter_filter")
        return out.reshape(3, 3)

    def __del__(self):
        if getattr(self, "_h", None) and _L is not None:
            _L.vstab_rotation_filter_destroy(self._h)
            self._h = None


# ---------------------------------------------------------------------------------------------
# the pipeline object (FrameSourceWarp replacement)
# ---------------------------------------------------------------------------------------------
def default_config(**kw):
    cfg = Config()
    _L.vstab_config_default(_c.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


class Stabilizer:
    """vstab_handle wrapper.  `frames`: list of packed NV12 CUDA tensors (cycled by the C ring
    source for `total` pulls) or a Python iterable of such tensors (python callback source)."""

    def __init__(self, frames, total=None, use_torch_stream=True, hold=12, bit_depth=8, readouts=None, ring_hold=None, **cfg_kw):
        """hold (iterable sources): vstab_frame.hold -- how many further pulls each tensor is kept alive and unchanged
        for; from smooth_radius + 14 on the library uses the tensors in place instead of copying them.
        bit_depth / readouts (list sources): P010 frames as int16 tensors of shape (h * 3 / 2, w); one 3x3 read-out
        rotation per ring frame (vstab_frame.readout_rotation).  ring_hold (list sources): the hold the ring source
        promises (default: forever, frames used in place; 0: every frame is copied into the library's ring)."""
        import torch
        self._keep = []
        self._src = Source()
        self._ring = None
        if isinstance(frames, (list, tuple)):
            f0 = frames[0]
            rows, w = f0.shape
            h = rows * 2 // 3
            ptrs = (_vp * len(frames))(*[f.data_ptr() for f in frames])
            self._keep += [frames, ptrs]
            self._ring = _vp()
            ro = None if readouts is None else np.ascontiguousarray(np.stack([np.asarray(r, np.float64).reshape(9) for r in readouts]))
            assert ro is None or ro.shape == (len(frames), 9)
            _check(_L.vstab_ring_source_create_ex(ptrs, len(frames), w, h, f0.stride(0) * f0.element_size(), len(frames) if total is None else total,
                                                  int(bit_depth), None if ro is None else _dptr(ro), _c.byref(self._ring), _c.byref(self._src)),
                   "vstab_ring_source_create_ex")
            if ring_hold is not None:
                _L.vstab_ring_source_set_hold(self._ring, int(ring_hold))
        else:
            it = iter(frames)
            state = {"next": None, "done": False}

            def fill(out, advance):
                if state["next"] is None and not state["done"]:
                    try:
                        state["next"] = next(it)
                    except StopIteration:
                        state["done"] = True
                if state["next"] is None:
                    return EOF
                f = state["next"]
                yp, uvp, pitch, w, h = _planes(f)
                o = out.contents
                o.y, o.uv, o.pitch_y, o.pitch_uv, o.width, o.height, o.mem, o.pts = yp, uvp, pitch, pitch, w, h, 0, 0
                o.hold = hold  # self._keep holds the last hold + 4 tensors alive
                if advance:
                    self._keep.append(f)
                    if len(self._keep) > hold + 4:
                        self._keep.pop(0)
                    state["next"] = None
                return 0
            self._pull = PULL_FN(lambda user, out: fill(out, True))
            self._peek = PULL_FN(lambda user, out: fill(out, False))
            self._src.pull, self._src.peek, self._src.user = self._pull, self._peek, None
        cfg = default_config(**cfg_kw)
        if use_torch_stream:
            cfg.stream = torch.cuda.current_stream().cuda_stream
        self._h = _vp()
        _check(_L.vstab_create(_c.byref(cfg), _c.byref(self._src), _c.byref(self._h)), "vstab_create")
        ow, oh = _c.c_int(), _c.c_int()
        Ki, Ko = np.zeros(9), np.zeros(9)
        _check(_L.vstab_get_output_info(self._h, _c.byref(ow), _c.byref(oh), _dptr(Ki), _dptr(Ko)), "vstab_get_output_info")
        self.out_size = (ow.value, oh.value)
        self.K_in, self.K_out = Ki.reshape(3, 3), Ko.reshape(3, 3)

    def pull_into(self, out, timing=None):
        """Returns True, or False at end of stream (the reference throws EOF)."""
        st = _L.vstab_pull_frame(self._h, out.data_ptr(), out.stride(0))
        if st == EOF:
            return False
        _check(st, "vstab_pull_frame")
        return True

    def pull_frames_into(self, outs, first, n):
        """vstab_pull_frames: n frames in one call, frame i into outs[(first + i) % len(outs)].  Returns the number emitted
        (fewer than n at end of stream)."""
        ring = getattr(self, "_out_ring", None)
        if ring is None or ring[0] is not outs:
            ptrs = (_c.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            pitches = (_c.c_size_t * len(outs))(*[o.stride(0) for o in outs])
            ring = self._out_ring = (outs, ptrs, pitches)
        done = _c.c_int(0)
        st = _L.vstab_pull_frames(self._h, int(n), ring[1], ring[2], len(outs), int(first) % len(outs), _c.byref(done))
        if st != EOF:
            _check(st, "vstab_pull_frames")
        return done.value

    def pull(self):
        import torch
        out = torch.empty((self.out_size[1], self.out_size[0], 3), dtype=torch.uint8, device="cuda")
        return out if self.pull_into(out) else None

    def pull_bgr16_into(self, out):
        """pixel_depth = 10 handles: out is a (h, w, 3) int16 CUDA tensor; values 0..1023."""
        st = _L.vstab_pull_frame_bgr16(self._h, out.data_ptr(), out.stride(0) * 2)
        if st == EOF:
            return False
        _check(st, "vstab_pull_frame_bgr16")
        return True

    def pull_p010_into(self, out_y, out_uv):
        """pixel_depth = 10 handles: P010 planes (int16 CUDA tensors, (h, w) and (ceil(h/2), 2 * ceil(w/2)))."""
        st = _L.vstab_pull_frame_p010(self._h, out_y.data_ptr(), out_y.stride(0) * 2, out_uv.data_ptr(), out_uv.stride(0) * 2)
        if st == EOF:
            return False
        _check(st, "vstab_pull_frame_p010")
        return True

    def pull_host(self):
        """-> (h, w, 3) numpy array in host memory, or None at end of stream."""
        out = np.empty((self.out_size[1], self.out_size[0], 3), np.uint8)
        st = _L.vstab_pull_frame_host(self._h, out.ctypes.data, out.strides[0])
        if st == EOF:
            return None
        _check(st, "vstab_pull_frame_host")
        return out

    def pull_nv12_into(self, y, uv, planar=False):
        """planar=True: vstab_pull_frame_nv12_planar, the plane-wise warp (no colour round trip)."""
        fn, name = (_L.vstab_pull_frame_nv12_planar, "vstab_pull_frame_nv12_planar") if planar else (_L.vstab_pull_frame_nv12, "vstab_pull_frame_nv12")
        st = fn(self._h, y.data_ptr(), y.stride(0), uv.data_ptr(), uv.stride(0))
        if st == EOF:
            return False
        _check(st, name)
        return True

    def pull_nv12(self, planar=False):
        """-> (luma, chroma) tensors, or None at end of stream."""
        y, uv = nv12_out_planes(self.out_size[0], self.out_size[1])
        return (y, uv) if self.pull_nv12_into(y, uv, planar) else None

    def pull_p010_planar_into(self, out_y, out_uv):
        """pixel_depth = 10 handles: the frame warped plane by plane (vstab_pull_frame_p010_planar), P010 planes as pull_p010_into."""
        st = _L.vstab_pull_frame_p010_planar(self._h, out_y.data_ptr(), out_y.stride(0) * 2, out_uv.data_ptr(), out_uv.stride(0) * 2)
        if st == EOF:
            return False
        _check(st, "vstab_pull_frame_p010_planar")
        return True

    def frame_log(self):
        out = []
        for i in range(_L.vstab_frame_log_count(self._h)):
            lg = FrameLog()
            _check(_L.vstab_get_frame_log(self._h, i, _c.byref(lg)), "vstab_get_frame_log")
            out.append(dict(key=bool(lg.key_frame), n_corners=lg.n_corners, n_tracked=lg.n_tracked, inliers=lg.n_inliers,
                            fallback=bool(lg.fallback), R=np.array(lg.R_frame).reshape(3, 3),
                            R_accum=np.array(lg.R_accum).reshape(3, 3)))
        return out

    def enable_profiling(self, level=2):
        _check(_L.vstab_enable_profiling(self._h, int(level)), "vstab_enable_profiling")

    def profile(self):
        p = Profile()
        _check(_L.vstab_get_profile(self._h, _c.byref(p)), "vstab_get_profile")
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def warp_rotation(self, i):
        R = np.zeros(9)
        _check(_L.vstab_get_warp_rotation(self._h, i, _dptr(R)), "vstab_get_warp_rotation")
        return R.reshape(3, 3)

    def close(self):
        if _L is None:
            return
        if getattr(self, "_h", None):
            _L.vstab_destroy(self._h)
            self._h = None
        if getattr(self, "_ring", None):
            _L.vstab_ring_source_destroy(self._ring)
            self._ring = None

    def __del__(self):
        self.close()
