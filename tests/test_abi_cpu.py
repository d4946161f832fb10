"""CPU tests of the drop-in boundary: libvstab.so loads, exports every symbol include/vstab.h
declares, and its host-side functions (cameras) agree with the oracle.  No device compute."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vstab.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"VSTAB_API[^;(]*?\b(vstab_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(vs):
    names = declared_symbols()
    assert len(names) >= 12
    L = ctypes.CDLL(vs.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the binding covers the same set, so header / library / binding cannot drift apart
    assert sorted(vs.SIGNATURES) == names


def test_library_embeds_gfx950_code_object(vs):
    assert "gfx950" in vs.version()
    blob = open(vs.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_warp_nv12_bgr" in blob


def test_no_oracle_in_product():
    """The product must not import, link, call or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "video-annotator_amd")
    banned = ("import oracle", "from oracle", "oracle/", "oracle.", "vo_", "libvstab_oracle", "createmap_ref")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                hits = [b for b in banned if b in text]
                assert not hits, (dirpath, f, hits)
    blob = open(os.path.join(pkg, "lib", "libvstab.so"), "rb").read()
    assert b"vo_create_map" not in blob and b"libvstab_oracle" not in blob


def test_device_count_without_gpu_is_an_error_code_not_a_crash(vs):
    n = vs.device_count()
    assert isinstance(n, int)


def test_bad_arguments_are_reported(vs):
    K = np.zeros(9)
    assert vs.lib.vstab_get_preset_camera(99, 1920, 1080, K.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == vs.ERR_INVALID
    assert b"preset" in vs.lib.vstab_last_error()
    p = np.zeros(17, np.float32).ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    # null device pointers / odd sizes are rejected before any launch
    assert vs.lib.vstab_warp_nv12_bgr(None, 0, None, 0, 64, 36, p, None, 0, 10, 10, None) == vs.ERR_INVALID
    assert vs.lib.vstab_pack_nv12(1, 16, 1, 16, 15, 8, 1, None) == vs.ERR_INVALID
    assert b"Mismatched image dimensions" in vs.lib.vstab_last_error()   # FrameSourceFfmpegOpenCl.cpp:54
    assert vs.lib.vstab_create_map(1, 4, 1, 4, 40000, 1, p, None) == vs.ERR_INVALID
    # round-2 entry points: the 10-bit operator, the detector choice, the P010 ring source, the 10-bit handle
    assert vs.lib.vstab_warp_p010(16, 128, 16, 128, 64, 36, p, None, 0, 5, 16, 1024, 10, 10, None) == vs.ERR_INVALID and b"blend" in vs.lib.vstab_last_error()
    assert vs.lib.vstab_warp_p010(16, 128, 16, 128, 64, 36, p, None, 9, 0, 16, 1024, 10, 10, None) == vs.ERR_INVALID and b"map mode" in vs.lib.vstab_last_error()
    assert vs.lib.vstab_warp_p010(16, 126, 16, 128, 64, 36, p, None, 0, 0, 16, 1024, 10, 10, None) == vs.ERR_INVALID     # luma pitch < 2 * width
    assert vs.lib.vstab_warp_p010(16, 128, 18, 128, 64, 36, p, None, 0, 0, 16, 1024, 10, 10, None) == vs.ERR_INVALID     # chroma pairs not 4-byte aligned
    xy, n = np.zeros(8, np.float32), ctypes.c_int()
    assert vs.lib.vstab_good_features_ex(16, 64, 64, 36, 4, 0.01, 3.0, 7, xy.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(n), None,
                                         None) == vs.ERR_INVALID
    ptrs, ring, src = (ctypes.c_void_p * 1)(16), ctypes.c_void_p(), vs.Source()
    assert vs.lib.vstab_ring_source_create_ex(ptrs, 1, 64, 36, 128, 10, 9, None, ctypes.byref(ring), ctypes.byref(src)) == vs.ERR_INVALID   # bit depth 9
    assert vs.lib.vstab_ring_source_create_ex(ptrs, 1, 64, 36, 64, 10, 10, None, ctypes.byref(ring), ctypes.byref(src)) == vs.ERR_INVALID   # pitch < 2 * width
    cfg = vs.default_config(pixel_depth=12)
    assert (cfg.pixel_depth, vs.default_config().pixel_depth, vs.default_config().blend) == (12, 8, vs.BLEND_EXACT)


def test_cameras_match_oracle_and_golden(vs):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    for row in kat["cameras"]:
        preset, w, h, sc, crop = int(row[0]), int(row[1]), int(row[2]), float(row[3]), bool(row[4])
        K = vs.get_preset_camera(preset, w, h)
        Ko, sz = vs.get_output_camera(K, w, h, sc, crop, 1.0)
        assert np.allclose(K.reshape(-1), row[5:14], rtol=0, atol=1e-11)
        assert np.allclose(Ko.reshape(-1), row[14:23], rtol=0, atol=1e-11)
        assert list(sz) == [int(row[23]), int(row[24])]


def test_zoom_and_scale_semantics(vs):
    K = vs.get_preset_camera(vs.GOPRO_H4B_WIDE169_MEASURED, 1920, 1080)
    Ko1, s1 = vs.get_output_camera(K, 1920, 1080, 1.0, False, 1.0)
    Ko2, s2 = vs.get_output_camera(K, 1920, 1080, 1.0, False, 2.0)
    Ko_o, s_o = oracle.get_output_camera(K, 1920, 1080, 1.0, False, 2.0)
    assert s2 == s_o and np.allclose(Ko2, Ko_o, atol=1e-11)
    assert s2[0] == s1[0] // 2 or s2[0] == (s1[0] - 1) // 2   # :156-163 zoom shrinks the canvas, focal stays
    assert Ko2[0, 0] == Ko1[0, 0]


def test_undistort_points_and_map_params_match_oracle(vs):
    K = oracle.get_preset_camera(4, 3840, 2160)
    Ko, _ = oracle.get_output_camera(K, 3840, 2160)
    rng = np.random.default_rng(1)
    pts = rng.uniform([0, 0], [3839, 2159], (200, 2))
    assert np.allclose(vs.fisheye_undistort_points(pts, K), oracle.fisheye_undistort_points(pts, K), atol=1e-12)
    R = oracle.rodrigues([0.01, 0.02, -0.03])
    assert np.allclose(vs.fisheye_undistort_points(pts, K, R, Ko), oracle.fisheye_undistort_points(pts, K, R, Ko), atol=1e-9)
    assert np.array_equal(vs.map_params(K, Ko, R), oracle.map_params(K, Ko, R))


def test_ctypes_mirrors_have_the_compiled_struct_sizes(vs):
    for k, t in enumerate((vs.Frame, vs.Source, vs.Config, vs.FrameLog, vs.Profile)):
        assert vs.lib.vstab_struct_size(k) == ctypes.sizeof(t), t.__name__
    assert vs.lib.vstab_struct_size(99) == -1


def test_abi_version_gates_vstab_create(vs):
    """vstab_config carries the struct-layout version: vstab_config_default writes it, vstab_create refuses a config that was
    zero-filled or assembled against another header (before anything touches the device), and the header, the library and
    the binding agree on the number."""
    text = open(os.path.join(ROOT, "include", "vstab.h")).read()
    ver = int(re.search(r"#define VSTAB_ABI_VERSION (0x[0-9a-fA-F]+|\d+)", text).group(1), 0)
    assert ver == vs.lib.vstab_abi_version() == vs.ABI_VERSION
    assert ver >> 8 == 0x565342 and ver & 255 >= 5   # "VSB" + the layout version: not a value any field of an older layout can hold
    cfg = vs.default_config()
    assert cfg.abi_version == vs.ABI_VERSION and cfg.map_precision == vs.MAP_PRECISION_OPENCL
    calls = []
    cb = vs.PULL_FN(lambda user, out: calls.append(1) or vs.EOF)
    src = vs.Source(cb, cb, None)
    h = ctypes.c_void_p()
    for bad in (vs.Config(), vs.default_config(abi_version=vs.ABI_VERSION - 1), vs.default_config(abi_version=4),   # zero-filled; other headers' layouts
                vs.default_config(read_ahead=17), vs.default_config(read_ahead=-1)):
        assert vs.lib.vstab_create(ctypes.byref(bad), ctypes.byref(src), ctypes.byref(h)) == vs.ERR_INVALID
        assert (b"vstab_config_default" in vs.lib.vstab_last_error() or b"read_ahead" in vs.lib.vstab_last_error()) and not h.value
    assert not calls   # refused before upstream was touched
