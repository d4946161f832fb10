"""CPU tests of the host-side motion model behind the C ABI (no device work): rotation estimator
against ground truth, SG filter / weights against the oracle and the golden vectors."""
import os

import numpy as np
import pytest

import oracle
import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pairs(K, w, h, R, n, rng, outlier_frac=0.0, noise=0.0):
    prev = rng.uniform([40, 40], [w - 40, h - 40], (n, 2))
    rays = np.array([[(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1]] for x, y in prev])
    th = np.linalg.norm(rays, axis=1)
    d = np.stack([rays[:, 0] * np.sin(th) / th, rays[:, 1] * np.sin(th) / th, np.cos(th)], axis=1)
    cur = synth.fisheye_project(K, d @ R.T) + rng.normal(0, noise, (n, 2))
    k = int(outlier_frac * n)
    cur[:k] += rng.uniform(-60, 60, (k, 2))
    return prev.astype(np.float32), cur.astype(np.float32)


def test_rotation_estimate_recovers_ground_truth(vs):
    rng = np.random.default_rng(0)
    for (w, h) in [(1920, 1080), (3840, 2160)]:
        K = oracle.get_preset_camera(4, w, h)
        Ko, _ = oracle.get_output_camera(K, w, h)
        for rv in [(0.004, -0.003, 0.002), (0.02, 0.015, -0.03), (0, 0, 0)]:
            Rt = oracle.rodrigues(rv)
            p, c = _pairs(K, w, h, Rt, 180, rng)
            R, inl = vs.estimate_rotation(p, c, K, Ko, seed=3)
            assert inl >= 170
            assert oracle.rotation_angle(R @ Rt.T) < 2e-4, (rv, oracle.rotation_angle(R @ Rt.T))
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_rotation_estimate_with_outliers_and_noise(vs):
    rng = np.random.default_rng(1)
    K = oracle.get_preset_camera(4, 1920, 1080)
    Ko, _ = oracle.get_output_camera(K, 1920, 1080)
    Rt = oracle.rodrigues((0.01, -0.02, 0.015))
    p, c = _pairs(K, 1920, 1080, Rt, 200, rng, outlier_frac=0.3, noise=0.2)
    R, inl = vs.estimate_rotation(p, c, K, Ko, seed=7)
    assert 120 <= inl <= 200
    assert oracle.rotation_angle(R @ Rt.T) < 1.5e-3


def test_rotation_estimate_equals_independent_restatement(vs):
    """The product's guess_camera_rotation (FrameSourceWarp.cpp:316-375: undistortion, seeded random depths, 5-point
    RANSAC at 8 px / 0.99 / 100 iterations, refit, Rodrigues) against oracle/geometry.py's numpy restatement on the same
    PCG32 stream: same inlier count, R to 1e-9 -- from small shakes to 12 degrees between frames, up to 50 % outliers --
    and both recover the true rotation."""
    rng = np.random.default_rng(5)
    for (w, h) in [(1920, 1080), (3840, 2160)]:
        K = oracle.get_preset_camera(4, w, h)
        Ko, _ = oracle.get_output_camera(K, w, h)
        for deg in (0.2, 3.0, 5.0, 8.0, 12.0):
            for outliers in (0.0, 0.3, 0.5):
                ax = rng.normal(size=3)
                Rt = oracle.rodrigues(ax / np.linalg.norm(ax) * np.deg2rad(deg))
                p, c = _pairs(K, w, h, Rt, 200, rng, outlier_frac=outliers, noise=0.2)
                seed = int(rng.integers(1, 1000))
                Rp, ip = vs.estimate_rotation(p, c, K, Ko, seed=seed)
                Ro, io = oracle.estimate_rotation(p, c, K, Ko, oracle.Pcg32(seed))
                assert ip == io and np.abs(Rp - Ro).max() < 1e-9, (deg, outliers, ip, io)
                assert ip >= 0.9 * (1 - outliers) * 200
                assert np.degrees(oracle.rotation_angle(Rp @ Rt.T)) < 0.1, (deg, outliers)


def test_bad_estimate_reaches_the_inlier_gate(vs):
    """FrameSourceWarp.cpp:432-438 through a BAD estimate rather than too few points: 200 tracked pairs of which 85 %
    are wrong leave fewer than 40 inliers, in the product and in the oracle alike."""
    rng = np.random.default_rng(9)
    K = oracle.get_preset_camera(4, 1920, 1080)
    Ko, _ = oracle.get_output_camera(K, 1920, 1080)
    p, c = _pairs(K, 1920, 1080, oracle.rodrigues((0.05, -0.02, 0.03)), 200, rng, outlier_frac=0.85, noise=0.3)
    Rp, ip = vs.estimate_rotation(p, c, K, Ko, seed=21)
    Ro, io = oracle.estimate_rotation(p, c, K, Ko, oracle.Pcg32(21))
    assert ip == io and ip < 40 and np.abs(Rp - Ro).max() < 1e-9


def test_pcg32_stream_matches_the_product(vs):
    """The seeded stream itself: identical estimates for identical seeds, different draws for different seeds."""
    g = oracle.Pcg32(42, 54)
    assert [g.next() for _ in range(3)] == [0xa15c02b7, 0x7b47f409, 0xba1d3330]   # pcg32-demo, seed 42 / sequence 54


def test_rotation_estimate_degenerate_inputs(vs):
    K = oracle.get_preset_camera(4, 1920, 1080)
    Ko, _ = oracle.get_output_camera(K, 1920, 1080)
    R, inl = vs.estimate_rotation(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), K, Ko)
    assert inl == 0 and np.array_equal(R, np.eye(3))          # FrameSourceWarp.cpp:367-371 -> (I, 0)
    R, inl = vs.estimate_rotation(np.ones((3, 2), np.float32) * 50, np.ones((3, 2), np.float32) * 50, K, Ko)
    assert inl == 0 and np.array_equal(R, np.eye(3))


def test_rotation_estimate_is_seeded(vs):
    rng = np.random.default_rng(2)
    K = oracle.get_preset_camera(4, 1920, 1080)
    Ko, _ = oracle.get_output_camera(K, 1920, 1080)
    p, c = _pairs(K, 1920, 1080, oracle.rodrigues((0.01, 0.0, 0.0)), 100, rng, noise=0.3)
    a = vs.estimate_rotation(p, c, K, Ko, seed=5)
    b = vs.estimate_rotation(p, c, K, Ko, seed=5)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]         # the reference's un-seeded rand() is replaced


def test_sg_weights_and_filter_match_oracle_and_golden(vs):
    kat = np.load(os.path.join(GOLD, "oracle_kat.npz"))
    assert np.allclose(vs.sg_weights(30), kat["sg_w30"], atol=1e-15)
    for m in (1, 2, 7, 30):
        assert np.allclose(vs.sg_weights(m), oracle.sg_weights(m), atol=1e-15)
    f = vs.RotationFilter(int(kat["sg_m"][0]))
    for R, exp in zip(kat["sg_traj"], kat["sg_filtered"]):
        f.add(R)
        assert np.allclose(f.filter(), exp, atol=1e-11)


def test_sg_filter_startup_reflection_quirk(vs):
    """Ring starts zero-filled: after ONE add the weighted sum is w[+m]*R with
    w[+m] < 0, whose polar factor U.V^T is -R (a reflection; gram_sg applies no determinant fix)."""
    R = oracle.rodrigues((0.1, 0.2, -0.1))
    f, o = vs.RotationFilter(30), oracle.RotationFilter(30)
    f.add(R), o.add(R)
    assert np.allclose(o.filter(), -R, atol=1e-12)
    assert np.allclose(f.filter(), -R, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# gyro samples -> rotations (the path gpmf.cpp:5-11 stubs)
# ---------------------------------------------------------------------------------------------
def _gyro_samples(rate_fn, t0, t1, hz):
    """GyroFrame records {start_ts, end_ts, roll, pitch, yaw}: the rate at the middle of each sample interval."""
    edges = np.arange(int(round((t1 - t0) * hz)) + 1) / hz + t0
    mid = 0.5 * (edges[:-1] + edges[1:])
    w = np.array([rate_fn(t) for t in mid])  # (x, y, z) = (pitch, yaw, roll)
    return np.stack([edges[:-1], edges[1:], w[:, 2], w[:, 0], w[:, 1]], 1)


def test_gyro_constant_rate_is_one_exponential_map(vs):
    w = np.array([0.11, -0.23, 0.31])  # rad/s about x, y, z
    s = _gyro_samples(lambda t: w, 0.0, 0.2, 400.0)
    tp, tf, tl = 0.0503, 0.0503 + 1 / 30, 0.0503 + 1 / 30 + 0.0081
    Rd, Rr = vs.gyro_integrate(s, 1.0, tp, tf, tl)
    assert np.allclose(Rd, oracle.rodrigues(w * (tf - tp)), atol=1e-14)
    assert np.allclose(Rr, oracle.rodrigues(w * (tl - tf)), atol=1e-14)
    # a body-rate gyro: rate_scale -1 gives the inverse rotation (what guess_camera_rotation would measure)
    Rb, _ = vs.gyro_integrate(s, -1.0, tp, tf, tl)
    assert np.allclose(Rb, Rd.T, atol=1e-14)
    # restatement agrees; time not covered by samples contributes nothing
    Od, Or = oracle.gyro_integrate(s, 1.0, tp, tf, tl)
    assert np.allclose(Rd, Od, atol=1e-15) and np.allclose(Rr, Or, atol=1e-15)
    Rg, _ = vs.gyro_integrate(s, 1.0, -1.0, 0.05, 0.06)
    assert np.allclose(Rg, oracle.rodrigues(w * 0.05), atol=1e-14)
    assert np.allclose(vs.gyro_integrate(s[:0], 1.0, 0.0, 0.1, 0.2)[0], np.eye(3))


def test_gyro_sinusoidal_shake_against_the_closed_form_and_a_fine_reference(vs):
    # one axis: rotations commute, the integral of the rate is the angle -- exact up to the sampling of the rate
    amp, f = 0.04, 7.0
    rate = lambda t: np.array([0.0, amp * 2 * np.pi * f * np.cos(2 * np.pi * f * t), 0.0])
    s = _gyro_samples(rate, 0.0, 0.5, 3200.0)
    tp, tf = 0.1, 0.1 + 1 / 29.97
    Rd, _ = vs.gyro_integrate(s, 1.0, tp, tf, tf)
    ang = amp * (np.sin(2 * np.pi * f * tf) - np.sin(2 * np.pi * f * tp))
    assert oracle.rotation_angle(Rd @ oracle.rodrigues([0, ang, 0]).T) < 1e-6   # midpoint sampling of a 7 Hz sine at 3.2 kHz
    # three axes with different phases (non-commuting): against the same product on a 64 x finer sampling, and the restatement
    rate3 = lambda t: np.array([0.9 * np.sin(40 * t), 0.7 * np.cos(31 * t + 0.3), 0.5 * np.sin(23 * t + 1.1)])
    coarse, fine = _gyro_samples(rate3, 0.0, 0.3, 800.0), _gyro_samples(rate3, 0.0, 0.3, 51200.0)
    tp, tf, tl = 0.0712, 0.0712 + 1 / 30, 0.0712 + 1 / 30 + 0.012
    Rd, Rr = vs.gyro_integrate(coarse, -1.0, tp, tf, tl)
    Fd, Fr = oracle.gyro_integrate(fine, -1.0, tp, tf, tl)
    assert oracle.rotation_angle(Rd @ Fd.T) < 1e-5 and oracle.rotation_angle(Rr @ Fr.T) < 5e-6   # 800 Hz piecewise-constant rate: 4e-6 rad
    Od, Or = oracle.gyro_integrate(coarse, -1.0, tp, tf, tl)
    assert np.allclose(Rd, Od, atol=1e-14) and np.allclose(Rr, Or, atol=1e-14)
    assert abs(np.linalg.det(Rd) - 1) < 1e-13 and np.allclose(Rd @ Rd.T, np.eye(3), atol=1e-13)


def test_gyro_bad_arguments(vs):
    s = _gyro_samples(lambda t: np.zeros(3), 0.0, 0.1, 100.0)
    with pytest.raises(vs.VstabError):
        vs.gyro_integrate(s[::-1], 1.0, 0.0, 0.01, 0.02)       # unordered
    with pytest.raises(vs.VstabError):
        vs.gyro_integrate(s, 1.0, 0.05, 0.01, 0.02)            # t_prev after t_first
