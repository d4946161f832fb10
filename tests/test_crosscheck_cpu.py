"""Cross-checks of the oracle's restatements of third-party arithmetic (OpenCV 4.5, gram_savitzky_golay --
neither is in the container) against INDEPENDENT implementations that are: scipy / numpy fp64 models.
They pin the parts of the oracle that the reference tree itself cannot pin."""
import numpy as np
import pytest
import scipy.linalg
import scipy.ndimage as ndi
import scipy.signal
from scipy.spatial.transform import Rotation

import oracle
import synth


def test_sg_weights_are_the_savitzky_golay_smoothing_coefficients():
    """gram_sg::SavitzkyGolayFilterConfig(m, t=0, n=2, s=0) (FrameSourceWarp.cpp:212) = the classic centre-point
    quadratic smoothing filter; scipy computes it by least squares, the oracle by the Gram-polynomial recursion."""
    for m in (1, 2, 5, 30, 60):
        ref = scipy.signal.savgol_coeffs(2 * m + 1, 2)
        got = oracle.sg_weights(m)
        assert np.allclose(got, ref, rtol=0, atol=1e-13), m
        assert abs(got.sum() - 1) < 1e-13


def test_rotation_filter_projection_is_the_polar_factor():
    """RotationFilter::filter() returns U*V^T of the weighted matrix sum = the orthogonal polar factor."""
    rng = np.random.default_rng(0)
    f = oracle.RotationFilter(3)
    for _ in range(9):
        f.add(oracle.rodrigues(rng.normal(0, 0.2, 3)))
    got = f.filter()
    w = oracle.sg_weights(3)
    M = sum(wi * Ri for wi, Ri in zip(w, f.buf))
    U, _ = scipy.linalg.polar(M)
    assert np.allclose(got, U, atol=1e-12)


def test_rodrigues_matches_scipy():
    rng = np.random.default_rng(1)
    for _ in range(50):
        rv = rng.normal(0, 1.0, 3)
        assert np.allclose(oracle.rodrigues(rv), Rotation.from_rotvec(rv).as_matrix(), atol=1e-13)
    assert np.array_equal(oracle.rodrigues((0, 0, 0)), np.eye(3))


def test_pyr_down_is_the_5tap_binomial_with_reflect101():
    """pyrDown: [1 4 6 4 1]/16 separable, BORDER_REFLECT_101 (= scipy 'mirror'), (v + 128) >> 8, every second pixel."""
    for (h, w, seed) in [(36, 64, 0), (37, 63, 1), (5, 7, 2)]:
        img = synth.luma(seed, w, h).astype(np.int64)
        k = np.array([1, 4, 6, 4, 1])
        acc = ndi.correlate1d(ndi.correlate1d(img, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
        exp = ((acc + 128) >> 8)[::2, ::2].astype(np.uint8)
        assert np.array_equal(oracle.pyr_down(img.astype(np.uint8)), exp), (h, w)


def test_scharr_derivatives_match_correlation():
    """calcSharrDeriv: dx = [3 10 3]^T x [-1 0 1], dy transposed, BORDER_REFLECT_101, int16 interleaved."""
    img = synth.luma(3, 50, 31).astype(np.int64)
    sm, df = np.array([3, 10, 3]), np.array([-1, 0, 1])
    dx = ndi.correlate1d(ndi.correlate1d(img, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndi.correlate1d(ndi.correlate1d(img, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    got = oracle.scharr(img.astype(np.uint8))
    assert np.array_equal(got[..., 0], dx) and np.array_equal(got[..., 1], dy)


def test_min_eig_matches_fp64_model():
    """cornerMinEigenVal(3, 3): Sobel / (4*3*255), 3x3 box sums of the products, smaller eigenvalue."""
    img = synth.luma(4, 96, 54).astype(np.float64)
    sob_s, sob_d = np.array([1, 2, 1.0]), np.array([-1, 0, 1.0])
    sc = 1.0 / (4 * 3 * 255)
    dx = ndi.correlate1d(ndi.correlate1d(img, sob_s, axis=0, mode="mirror"), sob_d, axis=1, mode="mirror") * sc
    dy = ndi.correlate1d(ndi.correlate1d(img, sob_d, axis=0, mode="mirror"), sob_s, axis=1, mode="mirror") * sc
    box = lambda a: ndi.correlate(a, np.ones((3, 3)), mode="mirror")
    a, b, c = box(dx * dx), box(dx * dy), box(dy * dy)
    exp = 0.5 * (a + c) - np.sqrt(0.25 * (a - c) ** 2 + b * b)
    got = oracle.min_eig(img.astype(np.uint8))
    assert np.allclose(got, exp, rtol=2e-4, atol=2e-7)


def test_remap_matches_linear_interpolation_within_quantisation():
    """cv::remap INTER_LINEAR = bilinear interpolation with the fraction rounded to 1/32 px and the result
    rounded half up: within 1 level + (local gradient)/64 of exact bilinear (scipy map_coordinates)."""
    h, w = 60, 80
    yy, xx = np.mgrid[0:h, 0:w]
    img = (120 + 80 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.uint8)   # smooth: gradients <= ~10 levels / px
    rng = np.random.default_rng(2)
    mx = rng.uniform(1, w - 2, (40, 50)).astype(np.float32)
    my = rng.uniform(1, h - 2, (40, 50)).astype(np.float32)
    got = oracle.remap_bilinear(img, mx, my).astype(np.float64)
    exp = ndi.map_coordinates(img.astype(np.float64), [my.astype(np.float64), mx.astype(np.float64)], order=1, mode="constant")
    assert np.abs(got - exp).max() <= 0.5 + 2 * 10 / 64 + 1e-9
    # on exact 1/32-px positions the only difference is the final rounding
    mxq, myq = np.round(mx * 32) / 32, np.round(my * 32) / 32
    got = oracle.remap_bilinear(img, mxq.astype(np.float32), myq.astype(np.float32)).astype(np.float64)
    exp = ndi.map_coordinates(img.astype(np.float64), [myq, mxq], order=1, mode="constant")
    assert np.array_equal(got, np.floor(exp + 0.5))


def test_nv12_to_bgr_is_bt601_limited_range_within_one_level():
    f = synth.nv12(6, 64, 36, full_range=True)
    h, w = 36, 64
    Y = f[:h].astype(np.float64)
    U = np.repeat(np.repeat(f[h:, 0::2], 2, axis=0), 2, axis=1).astype(np.float64) - 128
    V = np.repeat(np.repeat(f[h:, 1::2], 2, axis=0), 2, axis=1).astype(np.float64) - 128
    yy = 1.164 * (np.maximum(Y, 16) - 16)      # OpenCV clamps luma below 16
    exp = np.stack([yy + 2.018 * U, yy - 0.391 * U - 0.813 * V, yy + 1.596 * V], -1)
    got = oracle.cvt_nv12_bgr(f).astype(np.float64)
    assert np.abs(got - np.clip(exp, 0, 255)).max() <= 1.0


def test_lk_tracks_a_known_subpixel_translation():
    """calcOpticalFlowPyrLK on a frame pair with a known global shift recovers it to a few hundredths of a pixel."""
    w, h = 320, 240
    a = synth.luma(8, w, h)
    dx, dy = 2.3, -1.6
    b = synth.shifted(a, dx, dy)
    pts = oracle.good_features(a)
    pts = pts[(pts[:, 0] > 30) & (pts[:, 0] < w - 30) & (pts[:, 1] > 30) & (pts[:, 1] < h - 30)]
    nxt, st = oracle.pyr_lk(a, b, pts)
    assert st.mean() > 0.9
    d = (nxt - pts)[st > 0]
    assert np.abs(np.median(d[:, 0]) - dx) < 0.05 and np.abs(np.median(d[:, 1]) - dy) < 0.05


def test_opencv_itself_when_present():
    """if a machine with OpenCV ever appears, check the restatements against the real thing.  Neither the
    build container nor the GPU box has cv2 (no wheel, no network), so this normally skips; it is never required."""
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(0)
    w, h = 320, 180
    frame = synth.nv12(5, w, h)
    # a2: cvtColor(COLOR_YUV2BGR_NV12), CPU path -- bit-exact expected (20-bit fixed point)
    assert np.array_equal(oracle.cvt_nv12_bgr(frame), cv2.cvtColor(frame, cv2.COLOR_YUV2BGR_NV12))
    # a10: remap INTER_LINEAR, BORDER_CONSTANT 0 -- bit-exact expected (1/32-px quantisation, 15-bit weights)
    bgr = oracle.cvt_nv12_bgr(frame)
    mx = (np.tile(np.arange(w, dtype=np.float32), (h, 1)) * 0.93 + rng.uniform(-3, 3, (h, w))).astype(np.float32)
    my = (np.tile(np.arange(h, dtype=np.float32)[:, None], (1, w)) * 1.04 + rng.uniform(-3, 3, (h, w))).astype(np.float32)
    assert np.array_equal(oracle.remap_bilinear(bgr, mx, my), cv2.remap(bgr, mx, my, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0))
    # a3: goodFeaturesToTrack(200, 0.01, 30) -- same corner list expected (float summation order may move ties)
    gray = np.ascontiguousarray(frame[:h])
    ours = oracle.good_features(gray)
    theirs = cv2.goodFeaturesToTrack(gray, 200, 0.01, 30).reshape(-1, 2)
    assert len(ours) == len(theirs) and np.abs(ours - theirs).max() <= 1.0
    # a4: calcOpticalFlowPyrLK defaults -- tracks within the accumulation-order noise (test_oracle_cpu: <= 0.05 px)
    nxt = np.roll(gray, (1, 2), axis=(0, 1))
    o_pts, o_st = oracle.pyr_lk(gray, nxt, ours)
    c_pts, c_st, _ = cv2.calcOpticalFlowPyrLK(gray, nxt, ours.reshape(-1, 1, 2), None)
    both = (o_st > 0) & (c_st.reshape(-1) > 0)
    assert (o_st > 0).sum() == (c_st > 0).sum() and np.abs(o_pts - c_pts.reshape(-1, 2))[both].max() < 0.05
    # a4, the rule behind the iteration loop (err requested): features tracked out of the image by their last step
    for seed in (1, 39, 81):
        prev, nxt2, pts = synth.edge_leaving_pair(seed)
        o_pts, o_st = oracle.pyr_lk(prev, nxt2, pts)
        c_pts, c_st, _err = cv2.calcOpticalFlowPyrLK(prev, nxt2, pts.reshape(-1, 1, 2), None)
        assert np.array_equal(o_st > 0, c_st.reshape(-1) > 0), seed
