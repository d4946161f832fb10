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
LIB_PATH = os.path.join(_HERE, "lib", "libvstab.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
        "there is no CPU fallback for the HIP path")

# PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the one already mapped when libvstab.so
# resolves the same SONAME, or the process ends up with two HIP runtimes and no visible device.
import torch  # noqa: E402,F401  (plumbing: device memory + streams)

_L = ctypes.CDLL(LIB_PATH)

OK, EOF, ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_SOURCE = 0, -1, -2, -3, -4, -5

(GOPRO_H4B_WIDE43_PUBLISHED, GOPRO_H4B_WIDE43_MEASURED, GOPRO_H4B_WIDE43_MEASURED_STABILISATION,
 GOPRO_H4B_WIDE169_PUBLISHED, GOPRO_H4B_WIDE169_MEASURED, GOPRO_H4B_WIDE169_MEASURED_STABILISATION) = range(6)

_c = ctypes
_vp, _sz, _i, _d = _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_double
_dp, _fp, _ip = _c.POINTER(_c.c_double), _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)

# name -> (restype, argtypes); mirrors include/vstab.h one to one
SIGNATURES = {
    "vstab_last_error": (_c.c_char_p, []),
    "vstab_version": (_c.c_char_p, []),
    "vstab_device_count": (_i, []),
    "vstab_get_preset_camera": (_i, [_i, _i, _i, _dp]),
    "vstab_get_output_camera": (_i, [_dp, _i, _i, _d, _i, _d, _dp, _ip, _ip]),
    "vstab_fisheye_undistort_points": (_i, [_dp, _i, _dp, _dp, _dp, _dp]),
    "vstab_map_params": (None, [_dp, _dp, _dp, _fp]),
    "vstab_pack_nv12": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _vp]),
    "vstab_cvt_nv12_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _vp, _sz, _vp]),
    "vstab_create_map": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _vp]),
    "vstab_remap_bilinear": (_i, [_vp, _sz, _i, _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _i, _i, _vp]),
    "vstab_warp_nv12_bgr": (_i, [_vp, _sz, _vp, _sz, _i, _i, _fp, _vp, _sz, _i, _i, _vp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(_L, _name)  # AttributeError here = header/library mismatch: fail loudly
    _f.restype, _f.argtypes = _res, _args

lib = _L


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


def cvt_nv12_bgr(nv12, out=None):
    import torch
    yp, uvp, pitch, w, h = _planes(nv12)
    if out is None:
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=nv12.device)
    _check(_L.vstab_cvt_nv12_bgr(yp, pitch, uvp, pitch, w, h, out.data_ptr(), out.stride(0), _stream()),
           "vstab_cvt_nv12_bgr")
    return out


def create_map(params, cols, rows, device="cuda"):
    import torch
    p = np.ascontiguousarray(params, np.float32)
    mx = torch.empty((rows, cols), dtype=torch.float32, device=device)
    my = torch.empty((rows, cols), dtype=torch.float32, device=device)
    _check(_L.vstab_create_map(mx.data_ptr(), mx.stride(0) * 4, my.data_ptr(), my.stride(0) * 4, cols, rows,
                               _fptr(p), _stream()), "vstab_create_map")
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
