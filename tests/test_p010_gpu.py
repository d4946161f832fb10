"""GPU parity tests of vstab_warp_p010 (BASELINE.json config 5: P010 in, 10-bit BGR out, exact or fp16 blend, optional
rotation per output row) against its definition in the oracle (vo_warp_p010).  Bar: every 16-bit sample equal."""
import numpy as np
import pytest

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
