"""What a frame emitted by the pipeline object must be, by map precision (vstab_config.map_precision).

OPENCL (the library default): cvtColor (oracle) -> createMap as the REFERENCE's own kernel computes it on this GPU
(oracle/_ref/createMap.gfx950.co, /root/reference/opencv/createMap.cl built unmodified for gfx950) -> cv::remap (oracle).
IEEE: the CPU oracle's chain (every map operation IEEE-rounded).  Test infrastructure only."""
import numpy as np

import oracle

IEEE, OPENCL = 0, 1
DEFAULT = OPENCL   # vstab_config_default: map_precision = VSTAB_MAP_PRECISION_OPENCL


def warp(frame, params, dw, dh, prec=DEFAULT, rot_bottom=None, nearest=False):
    """8-bit BGR frame the warp must emit for `frame` (packed NV12) under `params` (and a last-row rotation)."""
    if prec == OPENCL:
        return oracle.warp_nv12_ref_gfx950(frame, params, dw, dh, rot_bottom, nearest)
    if nearest:
        return oracle.remap_nearest(oracle.cvt_nv12_bgr(frame), *oracle.create_map(params, dw, dh))
    if rot_bottom is not None:
        return oracle.warp_nv12_rs(frame, params, rot_bottom, dw, dh)
    return oracle.warp_nv12(frame, params, dw, dh)


def warp_p010(y, uv, params, dw, dh, rot_bottom=None, blend=0, prec=DEFAULT):
    """16-bit BGR (0..1023) frame of the 10-bit path."""
    if prec == OPENCL:
        return oracle.warp_p010_ref_gfx950(y, uv, params, dw, dh, rot_bottom, blend)
    return oracle.warp_p010(y, uv, params, dw, dh, rot_bottom, 0, blend)


def warp_planar(frame, params, dw, dh, prec=DEFAULT, rot_bottom=None, mode=0):
    """(y', uv') the plane-wise NV12 -> NV12 warp must emit for `frame` (packed NV12).  `mode`: IEEE projection pairs 0..4."""
    if prec == OPENCL:
        return oracle.warp_nv12_planar_ref_gfx950(frame, params, dw, dh, rot_bottom)
    return oracle.warp_nv12_planar(frame, params, dw, dh, mode, rot_bottom)


def warp_p010_planar(y, uv, params, dw, dh, rot_bottom=None, blend=0, prec=DEFAULT, mode=0):
    """(y', uv') P010 words of the plane-wise 10-bit warp."""
    if prec == OPENCL:
        return oracle.warp_p010_planar_ref_gfx950(y, uv, params, dw, dh, rot_bottom, blend)
    return oracle.warp_p010_planar(y, uv, params, dw, dh, mode, rot_bottom, blend)
